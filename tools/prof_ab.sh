#!/bin/bash
# rocprofv3 kernel stats of one bench.py configuration for each library build under
# neural-lam-dev_amd/build/ab/lib_<tag>.so (same box, HIP-graph replay): tools/prof_ab.sh "<tags>" <outdir> [bench args]
TAGS=$1; OUT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for L in $TAGS; do
  NLAM_LIB_PATH=$R/neural-lam-dev_amd/build/ab/lib_$L.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$L -o t -- python3 $R/bench.py --steps 20 --warmup 3 --windows 1 --no-cpu-baseline --no-fp32-compare --no-kernel-timing --no-other-configs "$@" > $OUT/$L.log 2>&1
  cp $(find $OUT/trace_$L -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$L.csv
  rm -rf $OUT/trace_$L
done
