#!/bin/bash
# Host threads x who decides x library variant on the GPU box (developer tool).
#   scripts/threads_sweep.sh "steps warmup" "lib:threads:mode ..."
#   lib = base | <name> (build/<name>.so);  mode = host (MMC_HOST_ACCEPT=1) | dev (the kernel decides,
#   launches one step ahead) | dev0 (the kernel decides, MMC_RUN_AHEAD=0)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/tsweep
set -- $1 "$2"
STEPS=$1; WARM=$2; shift 2
for spec in $1; do
  IFS=: read lib th mode <<< "$spec"
  if [ "$lib" = base ]; then unset MMC_HIP_LIB; else export MMC_HIP_LIB=$R/build/$lib.so; fi
  unset MMC_HOST_ACCEPT MMC_RUN_AHEAD MMC_DEVICE_ACCEPT
  case "$mode" in host) export MMC_HOST_ACCEPT=1;; dev) export MMC_DEVICE_ACCEPT=1;; dev0) export MMC_DEVICE_ACCEPT=1 MMC_RUN_AHEAD=0;; esac
  f=$R/gpurun_out/tsweep/${lib}_${th}_${mode}_s$STEPS
  timeout -k 10 300 python3 $R/bench.py --no-cpu --no-secondary --steps $STEPS --warmup $WARM --threads $th \
      > $f.json 2> $f.err || { echo "$spec FAILED"; tail -5 $f.err; exit 1; }
  python3 -c "import json;d=json.load(open('$f.json'));r=d['roofline'];print('%-22s steps %4d  value %.4e  cost/launch %.1f us  span %.1f us  acc %.4f'%('$spec',$STEPS,d['value'],r.get('avg_launch_us',0),r.get('launch_span_us') or 0,d['acceptance']))"
done
