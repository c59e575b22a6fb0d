#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 --kernel-trace CSV (kernel_trace.csv): calls, total and
average duration, plus the idle gaps between consecutive kernels of the steady state.
usage: trace_summary.py DIR [steps]   (DIR is searched for *kernel_trace.csv)"""
import csv, glob, os, sys, collections
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = collections.defaultdict(lambda: [0, 0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    t = tot[name]; t[0] += 1; t[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(t[1] for t in tot.values())
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{path}: {len(rows)} kernels, busy {busy/1e6:.2f} ms, span {span/1e6:.2f} ms")
for k, (c, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {k[:70]:70s} {c:6d} calls  {ns/1e6:8.3f} ms  avg {ns/c/1e3:8.1f} us")
# gaps in the last third of the trace (steady state: graph replays)
n = len(rows); tail = rows[2 * n // 3:]
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(tail, tail[1:])]
gaps = [g for g in gaps if g < 1_000_000]
if gaps:
    gs = sorted(gaps)
    print(f"steady-state gaps between consecutive kernels: n {len(gs)}  median {gs[len(gs)//2]/1e3:.1f} us  "
          f"mean {sum(gs)/len(gs)/1e3:.1f} us  p90 {gs[int(.9*len(gs))]/1e3:.1f} us  total {sum(gs)/1e6:.2f} ms")
    kt = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
    print(f"steady-state: kernel time {kt/1e6:.2f} ms vs gaps {sum(gs)/1e6:.2f} ms")
