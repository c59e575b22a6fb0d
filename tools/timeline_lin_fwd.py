#!/usr/bin/env python3
"""Where does a mesh-sized (26 k rows) projection launch spend its time?
NLAM_TIMELINE=1 python tools/timeline_lin_fwd.py"""
import ctypes, os, sys
os.environ["NLAM_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neural_lam_amd import ops
from neural_lam_amd._lib import lib
dev = "cuda"
for rows, nA, nB in ((26244, 64, 64), (26244, 64, 0), (255136, 64, 0)):
    x = torch.randn(1, rows, 64, device=dev)
    WA = torch.randn(nA, 64, device=dev); WB = torch.randn(nB, 64, device=dev) if nB else None
    out = torch.empty(1, rows, nA + nB, device=dev)
    for _ in range(3):
        ops.fused_lin_fwd(ops.mat(x), WA, None, WB, None, ops.mat(out))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.fused_lin_fwd(ops.mat(x), WA, None, WB, None, ops.mat(out)); e.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (3 * 1024))()
    assert lib.nlam_debug_lin_fwd_timeline(buf) == 0
    t = np.array(buf, dtype=np.float64).reshape(1024, 3)
    nwg = min(1024, (((rows + 31) // 32) + 3) // 4)
    t = t[:nwg]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    print(f"rows {rows} n_out {nA+nB}: event {s.elapsed_time(e)*1e3:.1f} us, {nwg} workgroups")
    print(f"  start  : min {us[:,0].min():.2f}  median {np.median(us[:,0]):.2f}  max {us[:,0].max():.2f} us")
    print(f"  prologue (weights->LDS): median {np.median(us[:,1]-us[:,0]):.2f}  max {(us[:,1]-us[:,0]).max():.2f} us")
    print(f"  tile loop              : median {np.median(us[:,2]-us[:,1]):.2f}  max {(us[:,2]-us[:,1]).max():.2f} us")
    print(f"  last workgroup exit at {us[:,2].max():.2f} us after the first start")

# ---- true per-launch cost in a dependent stream: 100 launches in one HIP graph
os.environ["NLAM_TIMELINE"] = "0"
for rows, nA, nB in ((26244, 64, 64), (26244, 64, 0), (255136, 64, 0)):
    x = torch.randn(1, rows, 64, device=dev)
    WA = torch.randn(nA, 64, device=dev); WB = torch.randn(nB, 64, device=dev) if nB else None
    out = torch.empty(1, rows, nA + nB, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            ops.fused_lin_fwd(ops.mat(x), WA, None, WB, None, ops.mat(out))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(100):
                ops.fused_lin_fwd(ops.mat(x), WA, None, WB, None, ops.mat(out))
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    print(f"graph of 100 x lin_fwd rows {rows} n_out {nA+nB}: {s.elapsed_time(e)*10:.2f} us per launch")
