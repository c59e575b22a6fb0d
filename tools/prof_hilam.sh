#!/bin/bash
# Kernel stats (rocprofv3 --kernel-trace --stats) and plain bench lines of the Hi-LAM configs:
# hidden 64 and 128 in the default arithmetic, hidden 256 in bf16 (BASELINE configs[2], [4]).
# usage (repo root, under gpurun): bash tools/prof_hilam.sh <tag>
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_hilam_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "64 bf16x3" "128 bf16x3" "256 bf16"; do
  set -- $cfg; D=$1; MODE=$2
  export NLAM_MFMA=$MODE
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_d$D -- python3 $R/bench.py --model hi_lam --hidden-dim $D --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-fp32-compare > $OUT/stats_d$D.out 2> $OUT/stats_d$D.err; echo "stats d=$D exit=$?"
  find $OUT/stats_d$D -name "*kernel_trace.csv" -delete
  python3 $R/bench.py --model hi_lam --hidden-dim $D --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-compare > $OUT/bench_d$D.json 2> $OUT/bench_d$D.err; echo "bench d=$D exit=$?"
done
ls $OUT
