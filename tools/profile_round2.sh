#!/bin/bash
# Round-2 profile on the GPU box: gpu tests, bench line, rocprofv3 kernel stats, SQ counters of the
# final kernels, PMC traffic per kernel (whole step) and per call site (m2m micro-benchmark).
# usage (repo root, under gpurun): bash tools/profile_round2.sh <tag> [extra bench args]
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
BARGS="--no-cpu-baseline --no-kernel-timing --no-fp32-compare $@"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 $BARGS > $OUT/stats.out 2> $OUT/stats.err; echo stats_exit=$?
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BARGS > /dev/null 2> $OUT/pmc_sq.err; echo sq_exit=$?
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BARGS > /dev/null 2> $OUT/pmc_sq2.err; echo sq2_exit=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BARGS > /dev/null 2> $OUT/pmc_fetch.err; echo fetch_exit=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph $BARGS > /dev/null 2> $OUT/pmc_write.err; echo write_exit=$?
# per call site: the m2m layer alone (its segment_sum launches are the m2m scatter-add)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/m2m_fetch -- python3 $R/tools/micro_m2m.py --iters 3 > /dev/null 2> $OUT/m2m_fetch.err; echo m2m_fetch_exit=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/m2m_write -- python3 $R/tools/micro_m2m.py --iters 3 > /dev/null 2> $OUT/m2m_write.err; echo m2m_write_exit=$?
python3 $R/tools/pmc_summary.py $OUT/pmc_sq $OUT/pmc_sq2 > $OUT/pmc_sq_summary.txt 2>&1
python3 $R/tools/make_traffic.py $OUT $OUT/traffic_step.json > /dev/null 2>&1
mkdir -p $OUT/m2m && cp -r $OUT/m2m_fetch $OUT/m2m/pmc_fetch && cp -r $OUT/m2m_write $OUT/m2m/pmc_write
python3 $R/tools/make_traffic.py $OUT/m2m $OUT/traffic_m2m_site.json > /dev/null 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
ls $OUT
