#!/usr/bin/env python3
"""Fixed vs per-row cost of the fused row kernels: times back-to-back launches for a
range of row counts (d = 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd import ops

d = 64
dev = "cuda"
gen = torch.Generator(device="cpu").manual_seed(0)
W1 = (torch.randn(d, 2 * d, generator=gen) / 11).to(dev)
Wsq = (torch.randn(d, d, generator=gen) / 8).to(dev)
Wp = (torch.randn(d, 3 * d, generator=gen) / 8).to(dev)
b = torch.randn(d, generator=gen).to(dev)
res = {}
for rows in (1024, 6561, 26244, 104976, 255136):
    x = torch.randn(1, rows, d, device=dev)
    agg = torch.randn(1, rows, d, device=dev)
    gy = torch.randn(1, rows, d, device=dev)
    out = torch.empty(1, rows, d, device=dev)
    P = torch.empty(1, rows, 2 * d, device=dev)
    gP = torch.randn(1, rows, 2 * d, device=dev)
    gx = torch.empty(1, rows, d, device=dev)
    gxb = torch.empty(1, rows, d, device=dev)
    dst = {"dW1": torch.empty(d, 2 * d, device=dev), "db1": torch.empty(d, device=dev),
           "dW2": torch.empty(d, d, device=dev), "db2": torch.empty(d, device=dev),
           "dgamma": torch.empty(d, device=dev), "dbeta": torch.empty(d, device=dev)}
    dst1 = dict(dst, dW1=torch.empty(d, d, device=dev))
    dWp = torch.empty(d, 3 * d, device=dev)
    cases = {
        "lin_fwd(64->128)": lambda: ops.fused_lin_fwd(ops.mat(x), Wp[:, d:2*d], None, Wp[:, 2*d:], b, ops.mat(P)),
        "mlp_fwd(k128)": lambda: ops.fused_mlp_fwd(ops.mat(x), ops.mat(agg), W1, b, Wsq, b, b, b, ops.mat(x), ops.mat(out), d, d),
        "mlp_fwd(k64)": lambda: ops.fused_mlp_fwd(ops.mat(x), None, Wsq, b, Wsq, b, b, b, ops.mat(x), ops.mat(out), d, d),
        "mlp_bwd(k128)+outer": lambda: ops.fused_mlp_bwd(ops.mat(x), ops.mat(agg), W1, b, Wsq, b, b, ops.mat(gy), ops.mat(gx), ops.mat(gxb), True, d, d, dst),
        "mlp_bwd(k64)": lambda: ops.fused_mlp_bwd(ops.mat(x), None, Wsq, b, Wsq, b, b, ops.mat(gy), ops.mat(gx), None, True, d, d, dst1),
        "lin_bwd(128->64)": lambda: ops.fused_lin_bwd(ops.mat(x), ops.mat(gP), Wp[:, d:2*d], Wp[:, 2*d:], ops.mat(gx), dWp[:, d:2*d], None, dWp[:, 2*d:], dst["db1"]),
    }
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.setdefault(name, {})[rows] = round(s.elapsed_time(e) / 20 * 1e3, 1)
for k, v in res.items():
    print(k, v)
