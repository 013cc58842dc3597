# M2 profile: kernel trace + PMC passes of scripts/m2_bench.py (batched R replicas + single systems)
TAG=${1:-m2}; RR=${2:-65536}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
python3 $R/scripts/m2_bench.py $RR | tee $OUT/m2_bench.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/scripts/m2_bench.py $RR"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq1 -- $B > $OUT/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || exit 1
python3 $R/scripts/m2_summary.py $OUT | tee $OUT/summary.txt
