#!/bin/bash
# MFMA utilisation of the conv kernels: SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE, one PMC
# pass over the serialised bench step.  Run on the GPU box; writes gpurun_out/mfma/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/mfma
rm -rf $OUT; mkdir -p $OUT
export CILRS_OVERLAP=0
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer --no-loader --profile-steps 0 > $OUT/bench.log 2>&1
python3 $R/tools/pmc_mfma_summary.py $OUT > $OUT/summary.log 2>&1
cat $OUT/summary.log
