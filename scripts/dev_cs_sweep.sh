#!/bin/bash
# sweep of the context server's shape (workgroups x waves) on the Loop() call surface
for cfg in "1 8" "1 2" "2 4" "4 4"; do
  set -- $cfg
  echo "== WGS=$1 WAVES=$2"
  MMC_CTX_WGS=$1 MMC_CTX_WAVES=$2 timeout -k 10 120 python scripts/dev_call_surface.py 2>&1 | grep -v "^potential" | tail -6
done
