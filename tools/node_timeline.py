#!/usr/bin/env python3
"""Workgroup timeline of nlam_node_bwd on the MEPS m2m chain (NLAM_TIMELINE_NODE=1 in the environment):
where the ~35 us of a launch that moves 30 MB go.  Prints, in microseconds relative to the first
workgroup's start: start spread, and mean / max duration of each phase."""
import ctypes
import os
import sys
import tempfile

os.environ.setdefault("NLAM_TIMELINE_NODE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from neural_lam_amd import graphgen  # noqa: E402
from neural_lam_amd._lib import lib  # noqa: E402
from neural_lam_amd.interaction_net import InteractionNet  # noqa: E402
from neural_lam_amd.models.graph_lam import ProcessorSequential  # noqa: E402
from neural_lam_amd.utils import load_graph  # noqa: E402

with tempfile.TemporaryDirectory() as tmp:
    graphgen.create_graph(tmp, graphgen.make_xy(238, 268))
    _, g = load_graph(tmp)
ei = g["m2m_edge_index"]
torch.manual_seed(0)
proc = ProcessorSequential([InteractionNet(ei, 64) for _ in range(2)]).cuda()
B, d, M, N = 4, 64, ei.shape[1], 6561
x = torch.randn(B, N, d, device="cuda", requires_grad=True)
e = torch.randn(B, M, d, device="cuda", requires_grad=True)
for _ in range(3):
    ox, oe = proc(x, e)
    (ox.sum() + oe.sum()).backward()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8))()
lib.nlam_debug_node_timeline(buf)
nwg = int(lib.nlam_node_bwd_grid(B, N))
t = torch.tensor([[buf[w * 8 + k] for k in range(6)] for w in range(min(nwg, 256))], dtype=torch.float64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["weights -> LDS", "sender gather (tile 1)", "G = gP W + g_res", "node update bwd + rest of loop",
         "fold + slab"]
print(f"{nwg} workgroups; starts spread over {float(us[:, 0].max()):.2f} us; last end {float(us[:, 5].max()):.2f} us")
for k, nm in enumerate(names):
    dt = us[:, k + 1] - us[:, k]
    print(f"  {nm:34s} mean {float(dt.mean()):6.2f}  max {float(dt.max()):6.2f} us")
