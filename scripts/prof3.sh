R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 1 1" "16384 16 8" "256 1 1"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3_$1 -- python3 $R/bench.py --replicas $1 --groups $2 --threads $3 --steps 100 --warmup 10 --no-cpu > $R/gpurun_out/prof_r3_$1.log 2>&1
  grep "^{" $R/gpurun_out/prof_r3_$1.log | cut -c1-120
  grep -h "k_move_eval_fast\|copyBuffer" $R/gpurun_out/prof_r3_$1/*/*kernel_stats.csv | cut -c1-200
done
