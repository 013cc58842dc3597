# tests + the default bench line without the CPU / secondary legs
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/quick
mkdir -p $OUT
timeout -k 10 600 python -m pytest $R/tests -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
run() { # label, args
  timeout -k 10 300 python3 $R/bench.py --no-cpu --no-secondary $2 > $OUT/x.json 2>$OUT/x.err || { cat $OUT/x.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$OUT/x.json')); print('$1','value %.3e'%d['value'],'ms/step %.4f'%d['ms_per_step'],'kernel us %.1f'%d['roofline']['avg_launch_us'],'frac %.3f'%d['roofline']['frac'],'drift %.1e'%d['energy_drift_rel'],'eval ns %.0f'%d['ns_per_full_energy_eval'])"
}
run "default" ""
run "R=16384" "--replicas 16384"
run "R=2048" "--replicas 2048"
