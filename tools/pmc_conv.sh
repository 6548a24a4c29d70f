#!/bin/bash
# rocprofv3 passes for the conv micro-benchmark and the bench step (run on the GPU box).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/conv_bench.py --iters 1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $R/tools/conv_bench.py --iters 1 > $OUT/pmc_sq2.log 2>&1
find $OUT -name "*.csv" | head -20
ls -la $OUT/*
