#!/usr/bin/env python3
"""Compact view of one bench.py JSON line: step time, per-kernel totals (call sites merged), and
the top call sites.  usage: bench_summary.py FILE [N]"""
import json, sys, collections
path = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
line = [l for l in open(path) if l.startswith("{")][-1]
j = json.loads(line)
print(f"{j['config']['workload'][:60]} | {j['ms_per_step']:.2f} ms/step  value {j['value']:.3e}  mode {j.get('mfma_mode')}")
r = j.get("roofline") or {}
print("roofline:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()
                    if k in ("kernel", "avg_launch_us", "frac", "frac_hbm", "frac_mfma", "bound")})
ks = j.get("kernels", {})
if not ks and j.get("kernels_file"):   # round 5: the full table lives in a side file
    import os
    for cand in (j["kernels_file"], os.path.join(os.path.dirname(path), os.path.basename(j["kernels_file"]))):
        if os.path.exists(cand):
            ks = json.load(open(cand))["kernels"]
            break
tot = collections.defaultdict(lambda: [0.0, 0.0])
for k, v in ks.items():
    t = tot[k.split("@")[0]]
    t[0] += v["ms_per_step"]; t[1] += v["calls_per_step"]
print(f"kernel time {sum(t[0] for t in tot.values()):.2f} ms, launches {sum(t[1] for t in tot.values()):.0f}")
for k, (ms, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:n]:
    print(f"  {k:28s} {ms:7.3f} ms  {c:5.0f} calls")
print("top sites:")
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_step"])[:n]:
    us = 1e3 * v["ms_per_step"] / v["calls_per_step"]
    print(f"  {k:34s} {v['ms_per_step']:7.3f} ms  {v['calls_per_step']:4.0f} x {us:7.1f} us  {v['mb_per_step']/max(v['ms_per_step'],1e-9)/1e3:6.2f} TB/s")
