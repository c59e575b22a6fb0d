#!/usr/bin/env python3
"""Split-bf16 (bf16x3) GEMM pilot on the projection kernel: accuracy vs float64 and graph-replay
time, with NLAM_BF16X3=0/1 (run once per setting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd import ops
dev = "cuda"
mode = os.environ.get("NLAM_MFMA", "fp32")
torch.manual_seed(0)
for rows, nA, nB in ((26244, 64, 64), (255136, 64, 0), (1020544, 64, 0)):
    x = torch.randn(1, rows, 64, device=dev)
    WA = torch.randn(nA, 64, device=dev) / 8
    WB = torch.randn(nB, 64, device=dev) / 8 if nB else None
    bA = torch.randn(nA, device=dev)
    out = torch.empty(1, rows, nA + nB, device=dev)
    ops.fused_lin_fwd(ops.mat(x), WA, bA, WB, None, ops.mat(out))
    n = min(rows, 20000)
    W = torch.cat([WA, WB], 0) if nB else WA
    ref = x[0, :n].double() @ W.double().T
    ref[:, :nA] += bA.double()
    err = (out[0, :n].double() - ref).abs().max().item() / ref.abs().max().item()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(50):
                ops.fused_lin_fwd(ops.mat(x), WA, bA, WB, None, ops.mat(out))
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    print(f"MFMA={mode} rows {rows:8d} n_out {nA+nB:3d}: {s.elapsed_time(e)*20:.2f} us/launch, max rel err {err:.2e}")
