#!/usr/bin/env python3
"""Attribute rocprofv3 dispatches to C-ABI entry points AND call sites ("nlam_edge_bwd@m2m").

rocprofv3 names a dispatch by its HIP kernel template, so every call site of one template is
lumped together in its CSVs.  The launch sequence of a training step is deterministic, so
bench.py (NLAM_BENCH_DUMP_ORDER=<file>) writes the ordered list of "entry@site" launches of one
step, and this tool walks the kernel trace in dispatch order against that list (cyclically: every
step of the run -- warm-up, HIP-graph replays, eager profile steps -- repeats it).

  site_stats.py --order order.json --trace DIR [--fetch DIR --write DIR] [--key graph_lam-64]
                [--csv out.csv] [--traffic profiles/traffic_sites.json]

--trace: rocprofv3 --kernel-trace output (durations).  --fetch / --write: rocprofv3 --pmc FETCH_SIZE /
--pmc WRITE_SIZE outputs of the SAME command (separate passes, as the MI355X guide prescribes);
bytes = 2 x FETCH_SIZE KB (gfx950: 128-byte requests tallied at 64 B) + WRITE_SIZE KB.
"""
import argparse
import collections
import csv
import glob
import json
import re

FAMILIES = [  # (kernel-name regex, family); the first match wins, template arguments included
    (r"^fs_lin_fwd_kernel<\d+, \d+, \d+, true>", "lin_bwd_data"),   # (transposed weights = data gradient)
    (r"^grid_fwd16_kernel", "grid_encode_fwd"),
    (r"^node_fwd16_kernel", "node_fwd"), (r"^node_bwd16_kernel", "node_bwd"),
    (r"^node_outer16_kernel", "node_outer"),
    (r"^lin_fwd16_multi_kernel", "lin_fwd"), (r"^lin_bwd16_multi_kernel", "lin_bwd"),
    (r"^edge_fwd_kernel", "edge_fwd"), (r"^edge_bwd2?_kernel", "edge_bwd"),
    (r"^mlp_fwd(16)?(_multi)?_kernel", "mlp_fwd"), (r"^mlp_bwd(16)?(_multi)?_kernel", "mlp_bwd"),
    (r"^(lin_fwd(16|_b3)?|wide_lin_fwd|fs_lin_fwd)_kernel", "lin_fwd"),
    (r"^lin_bwd_data_kernel", "lin_bwd_data"), (r"^lin_bwd(16)?_kernel", "lin_bwd"),
    (r"^outer_bwd(16)?_kernel", "outer_bwd"), (r"^(wide|fs)_outer_kernel", "wide_outer"),
    (r"^(fs_)?tail_fwd(_multi)?_kernel", "tail_fwd"), (r"^(fs_)?tail_bwd(_multi)?_kernel", "tail_bwd"),
    (r"^reduce_slabs_multi_kernel", "reduce_slabs_multi"), (r"^reduce_slabs_kernel", "reduce_slabs"),
    (r"^segment_sum_", "segment_sum"), (r"^sum_batch", "sum_batch"), (r"^concat_rows", "concat_rows"),
    (r"^boundary_mix", "boundary_mix"), (r"^affine_residual", "affine_residual"),
    (r"^state_step_wmse_bwd", "state_step_wmse_bwd"), (r"^(state_step_wmse|ssl_final)", "state_step_wmse_fwd"),
    (r"^state_step_bwd", "state_step_bwd"), (r"^state_step", "state_step"),
    (r"^pack_segments", "pack_segments"),
    (r"^scale_cols", "scale_cols"), (r"^wmse_(partial|final)", "wmse_fwd"), (r"^wmse_bwd", "wmse_bwd"),
    (r"^gemm", "gemm"), (r"^silu_fwd", "silu_fwd"), (r"^silu_bwd", "silu_bwd"),
    (r"^layernorm_fwd", "layernorm_fwd"), (r"^layernorm_bwd", "layernorm_bwd"),
    (r"^colsum", "colsum"), (r"^gather_rows", "gather_rows"), (r"^add_rows", "add_rows"),
    (r"^nll_partial", "nll_fwd"), (r"^nll_bwd", "nll_bwd"), (r"^std_head_fwd", "std_head_fwd"),
    (r"^std_head_bwd", "std_head_bwd"),
]


def kernel_family(name):
    full = name.replace("void ", "").strip()
    n = full.split("<")[0].split("(")[0].strip()
    for rx, fam in FAMILIES:
        if re.match(rx, full if "<" in rx else n):
            return fam
    return None


def entry_family(entry):
    e = entry.split("@")[0]
    e = e[5:] if e.startswith("nlam_") else e
    if e != "reduce_slabs_multi" and e.endswith("_multi"):
        e = e[:-6]
    if e == "tail_fwd_pre":   # (nlam_tail_fwd_pre launches tail_fwd_kernel<..., PRE = true>)
        e = "tail_fwd"
    return e


def read_rows(d, pattern):
    rows = []
    for f in glob.glob(d + "/**/*" + pattern, recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def attribute(order, dispatches):
    """dispatches: list of (dispatch_id, kernel_name); returns {dispatch_id: entry@site}."""
    fams = [entry_family(e) for e in order]
    known = set(fams)
    out, ptr, L, unmatched = {}, 0, len(order), 0
    for did, kname in dispatches:
        fam = kernel_family(kname)
        if fam is None or fam not in known:
            continue
        if fam == fams[ptr % L]:
            out[did] = order[ptr % L]
            ptr += 1
        elif ptr > 0 and fam == fams[(ptr - 1) % L]:
            out[did] = order[(ptr - 1) % L]      # second kernel of a two-kernel entry point
        else:
            unmatched += 1                       # (e.g. the back-to-back micro-benchmark at the end)
    return out, unmatched


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", required=True)
    ap.add_argument("--trace", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--key", default="graph_lam-64")
    ap.add_argument("--csv")
    ap.add_argument("--traffic")
    a = ap.parse_args()
    order = json.load(open(a.order))
    tr = read_rows(a.trace, "kernel_trace.csv")
    tr.sort(key=lambda r: int(r["Dispatch_Id"]))
    amap, unmatched = attribute(order, [(int(r["Dispatch_Id"]), r["Kernel_Name"]) for r in tr])
    stat = collections.defaultdict(lambda: {"n": 0, "ns": 0.0, "kernels": set()})
    for r in tr:
        site = amap.get(int(r["Dispatch_Id"]))
        if site:
            s = stat[site]
            s["n"] += 1
            s["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            s["kernels"].add(r["Kernel_Name"].replace("void ", "").split("(")[0][:70])
    traffic = {}
    for which, d in (("fetch", a.fetch), ("write", a.write)):
        if not d:
            continue
        rows = read_rows(d, "counter_collection.csv")
        cname = "FETCH_SIZE" if which == "fetch" else "WRITE_SIZE"
        seen = {}
        for r in rows:
            seen[int(r["Dispatch_Id"])] = r["Kernel_Name"]
        pm, _ = attribute(order, sorted(seen.items()))
        acc = collections.defaultdict(list)
        for r in rows:
            if r["Counter_Name"] == cname and int(r["Dispatch_Id"]) in pm:
                acc[pm[int(r["Dispatch_Id"])]].append(float(r["Counter_Value"]))
        for site, v in acc.items():
            traffic.setdefault(site, {})[which] = sum(v) / len(v)
    print(f"{len(amap)} dispatches attributed to {len(stat)} call sites, {unmatched} unmatched")
    lines = [("site", "calls", "avg_us", "total_ms", "hbm_MB_per_launch", "kernel")]
    for site, s in sorted(stat.items(), key=lambda kv: -kv[1]["ns"]):
        t = traffic.get(site, {})
        mb = (2.0 * t["fetch"] + t["write"]) * 1024 / 1e6 if "fetch" in t and "write" in t else ""
        lines.append((site, s["n"], round(s["ns"] / s["n"] / 1e3, 2), round(s["ns"] / 1e6, 3),
                      round(mb, 2) if mb != "" else "", " | ".join(sorted(s["kernels"]))))
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            csv.writer(f).writerows(lines)
    for ln in lines[:25]:
        print(",".join(str(x) for x in ln[:5]))
    if a.traffic and traffic:
        try:
            table = json.load(open(a.traffic))
        except (OSError, ValueError):
            table = {}
        ent = table.setdefault(a.key, {})
        for site, t in traffic.items():
            if "fetch" in t and "write" in t:
                ent[site] = {"fetch_bytes_per_launch": 2.0 * t["fetch"] * 1024,
                             "write_bytes_per_launch": t["write"] * 1024,
                             "hbm_bytes_per_launch": (2.0 * t["fetch"] + t["write"]) * 1024}
        table["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) per launch, attributed to "
                          "call sites by launch order (tools/site_stats.py); FETCH_SIZE x2: gfx950 tallies "
                          "128-byte requests at 64 B")
        json.dump(table, open(a.traffic, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
