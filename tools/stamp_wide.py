#!/usr/bin/env python3
"""Phase shares of the hidden-128 tail kernels (csrc/fused_wide.hip; NLAM_STAMP_WIDE=1) on the
level-0 same-level InteractionNet of Hi-LAM (6,561 nodes, 51,520 edges on the MEPS hierarchy;
here the multiscale m2m graph: 57,616 edges), batch 4.  s_memtime stamps summed over all waves."""
import ctypes, os, sys, tempfile
os.environ["NLAM_STAMP_WIDE"] = "1"
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
d, B = 128, 4
net = InteractionNet(ei, d).cuda()
N, M = net.tables.n_rec, ei.shape[1]
x = torch.randn(B, N, d, device="cuda", requires_grad=True)
e = torch.randn(B, M, d, device="cuda", requires_grad=True)
buf = (ctypes.c_ulonglong * 16)()
for it in range(3):
    ox, oe = net(x, x, e)
    (ox.sum() + (oe * oe).sum()).backward()
    torch.cuda.synchronize()
    lib.nlam_debug_fs_stamps(buf, 1 if it < 2 else 0)
vals = [buf[i] for i in range(16)]
names_f = ["slot tables, gathers a / b / c landed + summed, h stored", "h tile staged -> accumulator layout",
           "silu", "GEMM W2 silu(h) + b2", "LayerNorm + message tile", "receiver sums", "row stores (+ residual)"]
names_b = ["slot tables, h rows landed + staged", "gradient rows issued, silu, GEMM (z recomputed)",
           "gradient rows staged, LN backward, dbeta / dgamma sums", "gz tile + row stores",
           "h rows again + GEMM W2^T gz", "silu', gh tile + row stores", "receiver sums"]
ntile = B * net.tables.ntiles + B * ((N + 31) // 32)   # edge call + node call of the layer
for title, names, v in (("tail_fwd", names_f, vals[:7]), ("tail_bwd", names_b, vals[8:15])):
    tot = sum(v)
    print(f"{title} (edge call + node call of one layer): {tot / ntile:.0f} cycles per 32-row tile")
    for n, c in zip(names, v):
        print(f"   {n:58s} {100 * c / max(tot, 1):5.1f} %   {c / ntile:8.0f}")
