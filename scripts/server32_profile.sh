# PMC passes of the latency move server at 32 chains (BASELINE configs[2]'s share of one GPU): ONE
# dispatch of 10 000 steps; counters are per dispatch.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/server32
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu --no-secondary --replicas 32 --steps 10000 --warmup 300"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc1 -- $B > $OUT/pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2 -- $B > $OUT/pmc2.log 2>&1 || exit 1
echo done
