# Same-box A/B of an environment switch:  tools/ab_env.sh VAR "<values>" <reps> [bench.py args...]
VAR=$1; VALS=$2; REPS=$3; shift 3
for rep in $(seq $REPS); do for L in $VALS; do
  env $VAR=$L python bench.py "$@" --no-other-configs --no-cpu-baseline --no-fp32-compare --no-kernel-timing --windows 3 > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -5 gpurun_out/ab.err
  python -c "
import json
d=json.load(open('gpurun_out/ab.json'))
print('$VAR=$L', round(d['ms_per_step'],4), round(d['windows']['min_ms_per_step'],4))
"; done; done
