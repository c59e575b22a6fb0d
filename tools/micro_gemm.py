#!/usr/bin/env python3
"""Generic nlam_gemm throughput on the shapes of the hidden_dim=128 path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd import ops

dev = "cuda"
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

for rows, k, n in [(1020544, 384, 128), (1020544, 128, 128), (255136, 256, 128), (100656, 384, 128)]:
    x = torch.randn(rows, k, device=dev); W = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    y = torch.empty(rows, n, device=dev); gy = torch.randn(rows, n, device=dev); gx = torch.empty(rows, k, device=dev)
    dW = torch.empty(n, k, device=dev); db = torch.empty(n, device=dev)
    fl = 2.0 * rows * k * n
    t1 = bench(lambda: ops.linear_fwd(ops.mat(x), W, b, ops.mat(y)))
    t2 = bench(lambda: ops.linear_bwd_data(ops.mat(gy), W, ops.mat(gx)))
    t3 = bench(lambda: ops.linear_bwd_weight(ops.mat(gy), ops.mat(x), dW, db))
    print(f"rows={rows} k={k} n={n}: fwd {fl/t1/1e12:.1f} TF/s ({t1*1e6:.0f}us)  bwd_data {fl/t2/1e12:.1f} ({t2*1e6:.0f}us)  bwd_weight+colsum {fl/t3/1e12:.1f} ({t3*1e6:.0f}us)")
