# the context server's empty-command round trip under different shapes
for cfg in "MMC_CTX_LAT=4" "MMC_CTX_LAT=4 MMC_CTX_LOOKAHEAD=0" "MMC_CTX_LAT=2 MMC_CTX_LOOKAHEAD=0" "MMC_CTX_LAT=1 MMC_CTX_LOOKAHEAD=0" "MMC_CTX_WGS=1 MMC_CTX_WAVES=8"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python scripts/dev_call_surface.py 2>&1 | grep -i "ping\|five calls\|loop body" | tail -4
done
