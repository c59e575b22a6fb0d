#!/bin/bash
# Same-box A/B timing of library builds (neural-lam-dev_amd/build/ab/lib_<tag>.so, selected through
# NLAM_LIB_PATH): box-to-box variation of the step time is ~1.5 %, larger than most kernel tweaks.
#   tools/ab_libs.sh "<tags>" <reps> [bench.py args...]
TAGS=$1; REPS=$2; shift 2
for rep in $(seq $REPS); do for L in $TAGS; do
  NLAM_LIB_PATH=$PWD/neural-lam-dev_amd/build/ab/lib_$L.so python bench.py "$@" --no-cpu-baseline --no-fp32-compare --no-kernel-timing --windows 3 > gpurun_out/ab.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/ab.json'))
print('$L', round(d['ms_per_step'],4), round(d['windows']['min_ms_per_step'],4))
"; done; done
