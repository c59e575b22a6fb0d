R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
for rep in 1 2 3; do for L in 0 1; do
  NLAM_FUSE_LOSS=$L python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-fp32-compare --no-kernel-timing --windows 3 > gpurun_out/ab.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/ab.json'))
print('fuse_loss=$L', round(d['ms_per_step'],4), round(d['windows']['min_ms_per_step'],4))
"; done; done
cd /tmp && export TMPDIR=/tmp
for L in 1; do
NLAM_FUSE_LOSS=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$L -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-other-configs --no-cpu-baseline --no-fp32-compare --no-kernel-timing > $R/gpurun_out/prof$L.log 2>&1
echo "== fuse_loss=$L"
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/prof$L/p_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows:
    if 'state_step' in r['Name'] or 'wmse' in r['Name'] or 'ssl_' in r['Name']:
        print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
