#!/usr/bin/env python3
"""Per-kernel time of one InteractionNet forward + backward against the size of the graph
(N nodes, ~8 N edges, batch 4): what a launch costs when it is one tile per workgroup, and what a
tile costs once the device is full.  HIP events around every C-ABI launch (ops.KernelProfiler).

  python tools/micro_levels.py --hidden-dim 256 [--sizes 81 729 6561 26244]"""
import argparse, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--hidden-dim", type=int, default=256)
ap.add_argument("--sizes", type=int, nargs="+", default=[81, 729, 2187, 6561, 26244])
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
if args.hidden_dim == 256:
    os.environ.setdefault("NLAM_MFMA", "bf16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd import ops
from neural_lam_amd.interaction_net import InteractionNet

d, B = args.hidden_dim, 4
table = {}
for N in args.sizes:
    gen = torch.Generator().manual_seed(0)
    deg = torch.randint(8, 10, (N,), generator=gen)
    rec = torch.repeat_interleave(torch.arange(N), deg)
    M = rec.numel()
    send = (rec + torch.randint(-40, 40, (M,), generator=gen)) % N
    net = InteractionNet(torch.stack((send, rec)), d).cuda()
    x = torch.randn(B, N, d, device="cuda", requires_grad=True)
    e = torch.randn(B, M, d, device="cuda", requires_grad=True)
    for it in range(3 + args.iters):
        if it == 3:
            torch.cuda.synchronize()
            ops.PROFILER = ops.KernelProfiler()
        ox, oe = net(x, x, e)
        (ox.sum() + oe.sum()).backward()
    st = ops.PROFILER.collect()
    ops.PROFILER = None
    for k, v in st.items():
        table.setdefault(k, {})[N] = (v["ms"] / args.iters * 1e3, v["calls"] / args.iters)
print(f"hidden {d}, B = {B}: us per fwd+bwd of one InteractionNet (launches)")
print("%-34s" % "entry point" + "".join("%16s" % f"N={n}" for n in args.sizes))
tot = {n: 0.0 for n in args.sizes}
for k in sorted(table, key=lambda k: -sum(v[0] for v in table[k].values())):
    row = "%-34s" % k
    for n in args.sizes:
        us, c = table[k].get(n, (0.0, 0))
        tot[n] += us
        row += "%11.1f (%2d)" % (us, c)
    print(row)
print("%-34s" % "total" + "".join("%16.1f" % tot[n] for n in args.sizes))
