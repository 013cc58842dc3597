# for each library given: does test_batch_parts_agree pass, and the default bench line's kernel time
R=$GRAFT_REPO_ROOT
for L in "$@"; do
  MMC_HIP_LIB=$R/$L timeout -k 10 120 python -m pytest $R/tests/test_gpu_batch.py -q -x -k "parts_agree or batch_eval_chain" < /dev/null 2>&1 | tail -1
  MMC_HIP_LIB=$R/$L timeout -k 10 120 python3 $R/bench.py --no-cpu --no-secondary --steps 100 --warmup 10 > /tmp/ab.json 2> /tmp/ab.err < /dev/null || { tail -3 /tmp/ab.err; exit 1; }
  python3 -c "
import json;d=json.load(open('/tmp/ab.json'));print('$L', round(d['roofline']['avg_launch_us'],1), '%.4g'%d['value'], '%.2e'%d['energy_drift_rel'])"
done
