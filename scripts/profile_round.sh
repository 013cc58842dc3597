# Round profile: the default bench line, the rocprofv3 kernel trace of the same command, and the
# HBM traffic counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round_profile
mkdir -p $OUT
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
# the kernel's loads and stores without its arithmetic, at the bench's launch shape (30720 replicas per group)
[ -x $R/build/gather_bw ] && $R/build/gather_bw 30720 > $R/gpurun_out/gather_bw.json
tail -c 600 $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
# every launch alone on the GPU (one stream): durations are costs, and the counters below belong to them
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu --no-secondary --streams 1 > $OUT/trace_bench.log 2>&1 || exit 1
# the default (two streams): launches overlap, durations are spans
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 $R/bench.py --no-cpu --no-secondary > $OUT/trace2_bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu --no-secondary --streams 1 --steps 40 --warmup 8 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu --no-secondary --streams 1 --steps 40 --warmup 8 > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq1 -- python3 $R/bench.py --no-cpu --no-secondary --streams 1 --steps 40 --warmup 8 > $OUT/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --no-cpu --no-secondary --streams 1 --steps 40 --warmup 8 > $OUT/pmc_sq2.log 2>&1 || exit 1
# the kernels that actually run BASELINE configs[1] (one chain) and the reference's call surface:
# the persistent servers are ONE dispatch each, counters are per dispatch
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_server -- python3 $R/bench.py --no-cpu --no-secondary --replicas 1 --steps 20000 --warmup 300 > $OUT/trace_server.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_server1 -- python3 $R/bench.py --no-cpu --no-secondary --replicas 1 --steps 20000 --warmup 300 > $OUT/pmc_server1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_server2 -- python3 $R/bench.py --no-cpu --no-secondary --replicas 1 --steps 20000 --warmup 300 > $OUT/pmc_server2.log 2>&1 || exit 1
echo done
echo "now run: python3 scripts/summarize_profile.py roundN [kernel name] (in the container)"
