#!/usr/bin/env python3
"""Phase shares of the hidden-256 tail kernels (NLAM_STAMP=1, NLAM_MFMA=bf16) on an m2m-sized
InteractionNet: 6,561 nodes, ~57.6 k edges (in-degree 8-9), batch 4."""
import ctypes, os, sys
os.environ["NLAM_STAMP"] = "1"
os.environ.setdefault("NLAM_MFMA", "bf16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd._lib import lib
from neural_lam_amd.interaction_net import InteractionNet

gen = torch.Generator().manual_seed(0)
N, d, B = int(os.environ.get('N', '6561')), 256, 4
deg = torch.randint(8, 10, (N,), generator=gen)
rec = torch.repeat_interleave(torch.arange(N), deg)
M = rec.numel()
send = (rec + torch.randint(-90, 90, (M,), generator=gen)) % N
net = InteractionNet(torch.stack((send, rec)), d).cuda()
x = torch.randn(B, N, d, device="cuda", requires_grad=True)
e = torch.randn(B, M, d, device="cuda", requires_grad=True)
buf = (ctypes.c_ulonglong * 16)()
lib.nlam_debug_fs_stamps(buf, 1)
for it in range(3):
    ox, oe = net(x, x, e)
    (ox.sum() + oe.sum()).backward()
    torch.cuda.synchronize()
    if it < 2:
        lib.nlam_debug_fs_stamps(buf, 1)
lib.nlam_debug_fs_stamps(buf, 0)
names_f = ["rows issued + landed, h store, silu -> planes", "barrier", "residual issue + GEMM",
           "LayerNorm (2 barriers, z_keep store)", "output tile + tables + barrier",
           "aggregation + row stores", "prologue: header -> indices -> tables + barrier",
           "prologue: weight slice"]
vals = [buf[i] for i in range(8)]
tot = sum(vals)
ntile = B * ((M + 63) // 64) + B * ((N + 63) // 64)
nwg = min(256, B * ((M + 63) // 64)) + min(256, B * ((N + 63) // 64))
print(f"fs_tail_fwd (edge call + node call), M = {M}, B = {B}: {ntile} tiles on {nwg} workgroups")
for n, v in zip(names_f, vals):
    print(f"  {n:48s} {100 * v / max(tot, 1):5.1f} %  {v / nwg:9.0f} cycles / workgroup")
ntile = B * ((M + 63) // 64) + B * ((N + 63) // 64)
print(f"  total {tot / ntile:.0f} cycles per 64-row tile")
