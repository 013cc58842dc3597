#!/bin/bash
# A/B on the GPU box: the default bench line (no CPU / secondary legs) for each library variant given.
#   scripts/ab.sh "bench args" base v1 v2 ...     ("base" = the in-tree library, others build/<name>.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; shift
mkdir -p $R/gpurun_out/ab
for v in "$@"; do
  if [ "$v" = base ]; then unset MMC_HIP_LIB; else export MMC_HIP_LIB=$R/build/$v.so; fi
  timeout -k 10 300 python3 $R/bench.py --no-cpu --no-secondary $ARGS > $R/gpurun_out/ab/$v.json 2> $R/gpurun_out/ab/$v.err || { echo "$v FAILED"; tail -5 $R/gpurun_out/ab/$v.err; exit 1; }  # (a GPU fault: stop, nothing more runs)
  python3 -c "import json;d=json.load(open('$R/gpurun_out/ab/$v.json'));r=d.get('roofline',{});print('%-10s value %.4e  ms/step %.4f  kernel us %.1f  drift %.1e  acc %.4f'%('$v',d['value'],d['ms_per_step'],r.get('avg_launch_us',0),d['energy_drift_rel'],d['acceptance']))"
done
