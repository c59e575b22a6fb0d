#!/usr/bin/env python3
"""Phase shares of the round-4 edge backward kernel (fused_edge2.hip; instrumented build,
NLAM_STAMP2=1).  usage: python tools/stamp_edge_bwd2.py [m2m|g2m|m2g]"""
import ctypes, os, sys, tempfile
os.environ["NLAM_STAMP2"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd import graphgen
from neural_lam_amd._lib import lib
from neural_lam_amd.interaction_net import InteractionNet
from neural_lam_amd.utils import load_graph

which = sys.argv[1] if len(sys.argv) > 1 else "m2m"
with tempfile.TemporaryDirectory() as tmp:
    graphgen.create_graph(tmp, graphgen.make_xy(238, 268))
    _, g = load_graph(tmp)
ei = g[f"{which}_edge_index"]
upd = which == "m2m"
net = InteractionNet(ei, 64, update_edges=upd).cuda()
B = 4
n_s, n_r = net.tables.n_send, net.tables.n_rec
x = torch.randn(B, n_s, 64, device="cuda", requires_grad=True)
xr = x if upd else torch.randn(B, n_r, 64, device="cuda", requires_grad=True)
e = torch.randn(B if upd else 1, ei.shape[1], 64, device="cuda", requires_grad=True)
buf = (ctypes.c_ulonglong * 8)()
for it in range(3):
    out = net(x, xr, e)
    if upd:
        (out[0].sum() + (out[1] * out[1]).sum()).backward()
    else:
        out.sum().backward()
    torch.cuda.synchronize()
    lib.nlam_debug_edge_bwd_stamps(buf, 1)
vals = [buf[i] for i in range(8)]
tot = sum(vals)
names = ["P0 rows landed + staged, h", "P1 GEMM1 silu S-planes GEMM2 stats", "P2 gm, LN bwd, dgamma/dbeta, GZ planes",
         "P3 next idx, dW2, db2, gh", "P4 GH tile + planes", "P4 W1e^T gh + gh stores", "P4 receiver sums",
         "P5 dW1e + g_e stores | (no upd) gh stores + sums"]
ntiles = net.tables.ntiles * B
for n, v in zip(names, vals):
    print(f"{n:48s} {100*v/max(tot,1):5.1f} %   {v/ntiles:9.0f} cycles/tile")
print("total cycles/tile (100 MHz s_memtime ticks x 1?)", tot / ntiles)
