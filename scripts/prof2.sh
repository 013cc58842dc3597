set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2 -- python3 $R/bench.py --replicas 4096 --groups 8 --threads 8 --steps 100 --warmup 10 --no-cpu > $R/gpurun_out/prof_r2_bench.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/pmc_r2a -- python3 $R/bench.py --replicas 4096 --groups 8 --threads 8 --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/pmc_r2a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_r2b -- python3 $R/bench.py --replicas 4096 --groups 8 --threads 8 --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/pmc_r2b.log 2>&1
ls -R $R/gpurun_out/pmc_r2a | head
