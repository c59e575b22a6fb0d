#!/usr/bin/env python3
"""Phase shares of the m2m edge backward kernel (instrumented build, NLAM_STAMP=1)."""
import ctypes, os, sys, tempfile
os.environ["NLAM_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd import graphgen
from neural_lam_amd._lib import lib
from neural_lam_amd.interaction_net import InteractionNet
from neural_lam_amd.utils import load_graph

with tempfile.TemporaryDirectory() as tmp:
    graphgen.create_graph(tmp, graphgen.make_xy(238, 268))
    _, g = load_graph(tmp)
ei = g["m2m_edge_index"]
net = InteractionNet(ei, 64).cuda()
B = 4
x = torch.randn(B, net.tables.n_rec, 64, device="cuda", requires_grad=True)
e = torch.randn(B, ei.shape[1], 64, device="cuda", requires_grad=True)
buf = (ctypes.c_ulonglong * 8)()
for it in range(3):
    ox, oe = net(x, x, e)
    (ox.sum() + oe.sum()).backward()
    torch.cuda.synchronize()
    lib.nlam_debug_edge_bwd_stamps(buf, 1)
vals = [buf[i] for i in range(8)]
tot = sum(vals)
names = ["stage(gathers)", "recompute GEMM1+silu+GEMM2", "LN bwd + colsums", "dW2 + W2^T gz + silu'",
         "gh store + gPr reduce", "GH planes + dW1e outer", "issue next tile's gathers",
         "W1e^T gh + g_e store"]
ntiles = net.tables.ntiles * B
for n, v in zip(names, vals):
    print(f"{n:30s} {100*v/tot:5.1f} %   {v/ntiles:9.0f} cycles/tile")
print("total cycles/tile", tot / ntiles)
