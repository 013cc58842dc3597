set -x
python -m pytest tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -15 || exit 1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -- python3 $R/bench.py --replicas 1024 --groups 4 --steps 300 --warmup 30 --no-cpu > $R/gpurun_out/prof_r1_bench.log 2>&1
tail -2 $R/gpurun_out/prof_r1_bench.log
find $R/gpurun_out/prof_r1 -name "*stats*" | head
