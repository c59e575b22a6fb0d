#!/bin/bash
# Per-call-site kernel times and HBM traffic of one bench.py configuration (run on the GPU box):
#   tools/profile_sites.sh <key> <outdir> [bench.py args...]
# Three separate rocprofv3 passes of the same command (kernel trace; FETCH_SIZE; WRITE_SIZE --
# counters never share a pass with a trace), then tools/site_stats.py.
set -e
KEY=$1; OUT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --windows 1 --no-cpu-baseline --no-fp32-compare --no-other-configs $*"
NLAM_BENCH_DUMP_ORDER=$OUT/order.json timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace pass done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
echo "fetch pass done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
echo "write pass done"
python3 $R/tools/site_stats.py --order $OUT/order.json --trace $OUT/trace --fetch $OUT/fetch --write $OUT/write \
  --key $KEY --csv $OUT/sites.csv --traffic $OUT/traffic_sites.json
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
# raw traces are large: keep the summaries only
rm -rf $OUT/trace $OUT/fetch $OUT/write
