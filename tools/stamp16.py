#!/usr/bin/env python3
"""Per-segment cycle shares of the 16-row per-wave nlam_edge_bwd (diagnostic build, NLAM_STAMP16=1):
runs the layer micro-benchmark graph and prints the s_memtime sums per segment.
usage: NLAM_STAMP16=1 NLAM_K16=<mask> python tools/stamp16.py [m2m|g2m|m2g]"""
import ctypes
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from neural_lam_amd import graphgen  # noqa: E402
from neural_lam_amd._lib import lib  # noqa: E402
from neural_lam_amd.interaction_net import InteractionNet  # noqa: E402
from neural_lam_amd.utils import load_graph  # noqa: E402

graph = sys.argv[1] if len(sys.argv) > 1 else "m2m"
with tempfile.TemporaryDirectory() as tmp:
    graphgen.create_graph(tmp, graphgen.make_xy(238, 268))
    _, g = load_graph(tmp)
ei = g[f"{graph}_edge_index"]
upd = graph == "m2m"
torch.manual_seed(0)
net = InteractionNet(ei, 64, update_edges=upd).cuda()
B, d, M = 4, 64, ei.shape[1]
x_s = torch.randn(B, net.tables.n_send, d, device="cuda", requires_grad=True)
x_r = x_s if upd else torch.randn(B, net.tables.n_rec, d, device="cuda", requires_grad=True)
e = torch.randn(B if upd else 1, M, d, device="cuda", requires_grad=True)


def step():
    out = net(x_s, x_r, e)
    ((out[0].sum() + out[1].sum()) if upd else out.sum()).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.nlam_debug_k16_stamps(buf, 1)
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
lib.nlam_debug_k16_stamps(buf, 1)
names = ["e/g_e' landed+transposed", "prefetch issue + GEMM1", "Ps/Pr landed, h", "silu + S planes + GEMM2",
         "g + dbeta + LN bwd + dgamma", "GZ planes + db2 + dW2", "W2^T gz * silu'", "gh tile/stores/seg sums",
         "W1e^T gh + g_e stores", "loop overhead", "tail"]
tot = sum(buf[:11])
print(f"{graph}: total stamped cycles per launch {tot / n:.3e} (tiles {net.tables.ntiles} x B {B})")
for k, nm in enumerate(names):
    print(f"  {k:2d} {nm:32s} {buf[k] / n:12.3e}  {100.0 * buf[k] / max(tot, 1):5.1f} %")
