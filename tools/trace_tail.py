#!/usr/bin/env python3
"""Ordered tail of a rocprofv3 --kernel-trace CSV: the last N kernels with duration and the gap to
the previous one (tools/trace_tail.py <trace dir> [N] [--around NAME]).  With --around, only the
neighbours of kernels whose name contains NAME (e.g. copyBuffer) are printed."""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 110
around = sys.argv[sys.argv.index("--around") + 1] if "--around" in sys.argv else None
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
prev_end = None
for i, r in enumerate(rows):
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1000 if prev_end else 0
    prev_end = en
    if around and not any(around in rows[j]["Kernel_Name"] for j in range(max(0, i - 1), min(len(rows), i + 2))):
        continue
    print(f"{i:5d} {r['Kernel_Name'][:60]:60s} dur {(en - st) / 1000:8.2f} us gap {gap:7.2f}")
