# A/B timing of library builds on one box: scripts/ab.sh <rounds> libA.so libB.so ... (paths under
# the repo).  Interleaved rounds of the default bench line, kernel time and moves/s per build.
N=${1:-3}; shift
R=$GRAFT_REPO_ROOT
for i in $(seq $N); do
  for L in "$@"; do
    MMC_HIP_LIB=$R/$L python3 $R/bench.py --no-cpu --no-secondary $BENCH_ARGS > /tmp/ab.json 2> /tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python3 -c "
import json;d=json.load(open('/tmp/ab.json'));print('$L', round(d['roofline']['avg_launch_us'],1), '%.4g'%d['value'], '%.2e'%d['energy_drift_rel'])"
  done
done
