#!/bin/bash
# PMC passes over the layer micro-benchmark for kernel selections (NLAM_K16 masks).
# usage (on the GPU box): bash tools/pmc_r3.sh OUTDIR GRAPH MASK [MASK...]   (PASSES env: which)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
G=$2
shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A P
P[1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
P[2]="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P[3]="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"
P[4]="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
P[5]="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
P[6]="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P[7]="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
P[8]="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum"
P[9]="TD_TD_BUSY_sum TD_TC_STALL_sum"
for M in "$@"; do
  for i in ${PASSES:-1 2 3 4 5 6 7 8 9}; do
    NLAM_K16=$M timeout -k 10 100 rocprofv3 --kernel-trace --pmc ${P[$i]} --output-format csv -d $OUT/m${M}_p$i -- python3 $R/tools/micro_m2m.py --iters 2 --graph $G > /dev/null 2> $OUT/m${M}_p$i.err
    echo "mask $M pass $i exit=$?"
  done
done
python3 $R/tools/pmc_summary.py $OUT/m*_p* > $OUT/summary.txt 2>&1
echo summary lines: $(wc -l < $OUT/summary.txt)
