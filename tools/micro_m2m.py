#!/usr/bin/env python3
"""Layer micro-benchmark: one m2m InteractionNet layer fwd+bwd on
the MEPS multiscale mesh (6,561 nodes, 57,616 edges), plus the bare aggregate
(segment-sum) kernel.  Prints one JSON line; used under rocprofv3 for PMC passes."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--graph", default="m2m", choices=["m2m", "g2m", "m2g"])
args = ap.parse_args()

from neural_lam_amd import graphgen, ops  # noqa: E402
from neural_lam_amd.interaction_net import InteractionNet  # noqa: E402
from neural_lam_amd.utils import load_graph  # noqa: E402

with tempfile.TemporaryDirectory() as tmp:
    graphgen.create_graph(tmp, graphgen.make_xy(238, 268))
    _, g = load_graph(tmp)
ei = g[f"{args.graph}_edge_index"]
torch.manual_seed(0)
upd = args.graph == "m2m"
net = InteractionNet(ei, args.dim, update_edges=upd).cuda()
net.tables.tag = args.graph
B, d, M = args.batch, args.dim, ei.shape[1]
n_s, n_r = net.tables.n_send, net.tables.n_rec
x_s = torch.randn(B, n_s, d, device="cuda", requires_grad=True)
x_r = x_s if args.graph == "m2m" else torch.randn(B, n_r, d, device="cuda", requires_grad=True)
e = torch.randn(B if upd else 1, M, d, device="cuda", requires_grad=True)


def step():
    out = net(x_s, x_r, e)
    if upd:
        (out[0].sum() + out[1].sum()).backward()
    else:
        out.sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    step()
torch.cuda.synchronize()
t_layer = (time.perf_counter() - t0) / args.iters

ops.PROFILER = ops.KernelProfiler()
for _ in range(args.iters):
    step()
stats = ops.PROFILER.collect()
ops.PROFILER = None

# bare aggregate kernel
msg = torch.randn(B, M, d, device="cuda")
agg = torch.empty(B, n_r, d, device="cuda")
t = net.tables
for _ in range(3):
    ops.segment_sum(ops.mat(msg), t.csr_rowptr, t.csr_eid, ops.mat(agg))
torch.cuda.synchronize()
s, eend = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50):
    ops.segment_sum(ops.mat(msg), t.csr_rowptr, t.csr_eid, ops.mat(agg))
eend.record()
torch.cuda.synchronize()
t_agg = s.elapsed_time(eend) / 50 / 1e3
agg_bytes = B * (4.0 * d * (M + n_r) + 4.0 * M + 4.0 * (n_r + 1))
print(json.dumps({
    "graph": args.graph, "B": B, "d": d, "M": M, "n_rec": n_r,
    "layer_fwd_bwd_us": t_layer * 1e6,
    "receiver_updates_per_s": B * n_r / t_layer,
    "aggregate_us": t_agg * 1e6, "aggregate_GBps": agg_bytes / t_agg / 1e9,
    "aggregate_frac_of_8TBps": agg_bytes / t_agg / 8e12,
    "kernels_us": {k: round(v["ms"] * 1e3 / args.iters, 2) for k, v in
                   sorted(stats.items(), key=lambda kv: -kv[1]["ms"])},
    "kernels_TFLOPs": {k: round(v["flops"] / (v["ms"] / 1e3) / 1e12, 2) for k, v in stats.items()
                       if v["flops"] > 0 and v["ms"] > 0},
}))
