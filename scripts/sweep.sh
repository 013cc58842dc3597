# Launch-size / stream-layout / move-generation sweep behind profiles/README.md's second table.
#   gpurun -- 'bash scripts/sweep.sh'
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sweep
mkdir -p $OUT
run() { # label, bench.py arguments
  timeout -k 10 300 python3 $R/bench.py --no-cpu --no-secondary $2 > $OUT/x.json 2>$OUT/x.err || { cat $OUT/x.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$OUT/x.json')); print('$1','value %.3e'%d['value'],'ms/step %.4f'%d['ms_per_step'],'kernel us %.1f'%d['roofline']['avg_launch_us'],'frac %.3f'%d['roofline']['frac'])" | tee -a $OUT/sweep.txt
}
: > $OUT/sweep.txt
run "device moves, 2 x 8192" "--replicas 16384"
run "device moves, 2 x 16384" "--replicas 32768"
run "device moves, 2 streams, 2 x 32768 (default)" ""
run "device moves, 2 x 65536" "--replicas 131072 --steps 300 --warmup 30"
run "device moves, 1 stream, 2 x 32768" "--streams 1"
run "device moves, 2 host threads" "--threads 2"
run "host moves, 8 threads, 2 x 8192" "--device-moves 0 --replicas 16384 --threads 8"
run "host moves, 2 threads, 2 x 8192" "--device-moves 0 --replicas 16384 --threads 2"
run "1 chain" "--replicas 1 --groups 1 --threads 1 --zero-copy-moves 1 --device-moves 0 --steps 3000 --warmup 300"
run "32 chains" "--replicas 32 --groups 2 --threads 2 --zero-copy-moves 1 --device-moves 0 --steps 3000 --warmup 300"
