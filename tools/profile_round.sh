#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats, PMC traffic passes.
# usage (from the repo root, under gpurun): bash tools/profile_round.sh <tag>
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo bench_exit=$?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/stats.out 2> $OUT/stats.err; echo stats_exit=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/pmc_fetch.err; echo fetch_exit=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/pmc_write.err; echo write_exit=$?
# keep the merged output small: drop the per-dispatch traces
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -30
