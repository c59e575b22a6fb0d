#!/usr/bin/env python3
"""nlam_edge_bwd alone on the MEPS graphs: the kernel families side by side in ONE process
(nlam_set_k16 selects which kernel the entry point dispatches to).  For every family: outputs
compared with the first one (gh, gPr, g_e, every weight gradient) and the launch time from one
event pair around `--iters` back-to-back launches.

  python tools/micro_edge_bwd.py [--graph m2m|g2m|m2g] [--batch 4] [--iters 30] [--masks ...]

masks: K16 bit masks (fused_params.h); default compares the round-3 default (511) with the
round-4 pipelined kernel (2047)."""
import argparse
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--graph", default="m2m", choices=["m2m", "g2m", "m2g"])
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--masks", type=int, nargs="+", default=[511, 2047])
ap.add_argument("--mean", action="store_true")
ap.add_argument("--no-geo", action="store_true")
ap.add_argument("--bsum", action="store_true",
                help="(g2m / m2g) also ask for dPe = sum_b gh[b]; checked against gh.sum(0)")
args = ap.parse_args()

from neural_lam_amd import graph, graphgen, ops  # noqa: E402
from neural_lam_amd._lib import lib  # noqa: E402
from neural_lam_amd.ops import mat  # noqa: E402
from neural_lam_amd.utils import load_graph  # noqa: E402

with tempfile.TemporaryDirectory() as tmp:
    graphgen.create_graph(tmp, graphgen.make_xy(238, 268))
    _, g = load_graph(tmp)
ei = g[f"{args.graph}_edge_index"]
if isinstance(ei, (list, tuple)):
    ei = ei[0]
send, rec, n_rec, n_send = graph.normalise_edge_index(ei)
if args.graph == "m2m":
    n_send = n_rec
t = graph.EdgeTables(send, rec, n_send, n_rec).cuda()
t.tag = args.graph
dev = torch.device("cuda")
B, d, M = args.batch, 64, int(send.shape[0])
upd = args.graph == "m2m"
torch.manual_seed(0)
e = torch.randn(B if upd else 1, M, d, device=dev)
ps = torch.randn(B, n_send, d, device=dev)
pr = torch.randn(B, n_rec, d, device=dev)
W1e = torch.randn(d, d, device=dev) / 8 if upd else None
W2 = torch.randn(d, d, device=dev) / 8
b2 = torch.randn(d, device=dev) / 8
gam = 1 + torch.randn(d, device=dev) / 8
g_agg = torch.randn(B, n_rec, d, device=dev)
geo = torch.randn(B, M, d, device=dev) if (upd and not args.no_geo) else None


def run():
    gh = torch.empty(B, M, d, device=dev)
    gpr = torch.empty(B, n_rec, d, device=dev)
    g_e = torch.empty(B, M, d, device=dev) if upd else (
        torch.empty(1, M, d, device=dev) if args.bsum else None)
    dW1e = torch.empty(d, d, device=dev) if upd else None
    dW2, db2 = torch.empty(d, d, device=dev), torch.empty(d, device=dev)
    dg, dbt = torch.empty(d, device=dev), torch.empty(d, device=dev)
    ops.fused_edge_bwd(t, mat(e), upd, mat(ps), mat(pr), W1e, W2, b2, gam, mat(g_agg),
                       mat(geo) if geo is not None else None, mat(gh), mat(gpr),
                       mat(g_e) if g_e is not None else None, args.mean, d, dW1e, dW2, db2, dg, dbt)
    out = {"gh": gh, "gpr": gpr, "dW2": dW2, "db2": db2, "dgamma": dg, "dbeta": dbt}
    if upd:
        out.update(g_e=g_e, dW1e=dW1e)
    elif args.bsum:
        out["dPe_minus_gh_sum"] = g_e - gh.sum(0, keepdim=True) + 1.0   # (== 1 where they agree)
    return out


res = {}
ref = None
for m in args.masks:
    lib.nlam_set_k16(m)
    out = run()
    torch.cuda.synchronize()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        run()
    en.record()
    torch.cuda.synchronize()
    # (the entry point includes the slab reduction launch: time the pair, report it as such)
    ent = {"us_per_call_incl_slab_reduce": s.elapsed_time(en) * 1e3 / args.iters}
    if ref is None:
        ref = out
    else:
        ent["max_rel_err_vs_first"] = {
            k: float((out[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30)) for k in out}
        ent["finite"] = all(bool(torch.isfinite(v).all()) for v in out.values())
    res[str(m)] = ent
print(json.dumps({"graph": args.graph, "B": B, "M": M, "n_rec": n_rec, "n_send": n_send,
                  "ntiles": int(t.ntiles), "mean": args.mean, "results": res}))
