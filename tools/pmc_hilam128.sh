#!/bin/bash
# SQ counters of the hidden-256 (bf16) kernels: one eager Hi-LAM-256 step under rocprofv3 --pmc.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_hilam128
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NLAM_MFMA=bf16x3
BARGS="--model hi_lam --hidden-dim 128 --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing --no-fp32-compare"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $BARGS > /dev/null 2> $OUT/pmc_sq.err; echo sq_exit=$?
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $BARGS > /dev/null 2> $OUT/pmc_sq2.err; echo sq2_exit=$?
python3 $R/tools/pmc_summary.py $OUT/pmc_sq $OUT/pmc_sq2 > $OUT/pmc_sq_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
ls $OUT
