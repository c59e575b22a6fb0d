#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on kernels with known byte counts (a streaming row
# copy and 256-byte-row gathers): tools/pmc_gather_calib.sh <outdir>.  Two --pmc passes, counters only.
set -e
OUT=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 $R/tools/pmc_gather_calib.py > $OUT/workload.json 2> $OUT/f.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python3 $R/tools/pmc_gather_calib.py > /dev/null 2> $OUT/w.log
python3 - <<PY
import csv, glob, json
wl = json.loads([l for l in open("$OUT/workload.json") if l.startswith("{")][-1])
def series(d, cname):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname and ("gather_rows" in r["Kernel_Name"] or "copy_rows" in r["Kernel_Name"]):
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
    rows.sort()
    return rows
fetch, write = series("$OUT/f", "FETCH_SIZE"), series("$OUT/w", "WRITE_SIZE")
lines = ["case; FETCH_SIZE KB x1024 per launch (raw, NOT doubled); ratio to 'every distinct row once'; ratio to 'every gathered row'; WRITE_SIZE x1024 per launch; ratio to written bytes"]
i = 0
for name, ent in wl.items():
    n = ent["launches"]
    f = [v for _, _, v in fetch[i:i + n]]; w = [v for _, _, v in write[i:i + n]]
    i += n
    fb, wb = sum(f[1:]) / (n - 1) * 1024, sum(w[1:]) / (n - 1) * 1024     # (first launch of a case: cold)
    if "gather" in name:
        a, b = ent["read_bytes_if_every_row_fetched_once"], ent["read_bytes_if_every_gather_fetched"]
        lines.append(f"{name}; {fb/1e6:.1f} MB; {fb/a:.3f}; {fb/b:.3f}; {wb/1e6:.1f} MB; {wb/ent['write_bytes']:.3f}")
    else:
        lines.append(f"{name}; {fb/1e6:.1f} MB; {fb/ent['read_bytes']:.3f}; -; {wb/1e6:.1f} MB; {wb/ent['write_bytes']:.3f}")
open("$OUT/calibration.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/f $OUT/w
