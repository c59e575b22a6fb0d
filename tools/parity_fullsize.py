#!/usr/bin/env python3
"""One m2m InteractionNet at full MEPS size (6,561 mesh nodes, 57,616 edges, B = 1) at hidden
NLAM_WIDE_D in THIS process's arithmetic mode (NLAM_MFMA) against the CPU oracle: forward, input
gradients, every parameter gradient.  Used for the widths / modes a pytest process cannot switch
to (hidden 256 runs in bf16 only):  NLAM_MFMA=bf16 NLAM_WIDE_D=256 python tools/parity_fullsize.py
Bars: fp32-grade modes 1e-4 / 1e-3; bf16 1e-2 / 5e-2 (relative to max|ref| per tensor)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import nlam_oracle as orc
from neural_lam_amd import graphgen
from neural_lam_amd._lib import lib
from neural_lam_amd.interaction_net import InteractionNet
from neural_lam_amd.utils import load_graph

d = int(os.environ.get("NLAM_WIDE_D", "256"))
mode = {0: "fp32", 1: "bf16x3", 2: "bf16"}[int(lib.nlam_mfma_mode())]
fwd_bar, grad_bar = (1e-2, 5e-2) if mode == "bf16" else (1e-4, 1e-3)
print("mfma mode:", mode, "width:", d)


def rel(a, b):
    return float((a.detach().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30))


with tempfile.TemporaryDirectory() as tmp:
    graphgen.create_graph(tmp, graphgen.make_xy(238, 268))
    _, g = load_graph(tmp)
ei = g["m2m_edge_index"]
N, M = 6561, ei.shape[1]
torch.manual_seed(10)
net = InteractionNet(ei, d)
gen = torch.Generator().manual_seed(11)
with torch.no_grad():
    for p in net.parameters():
        if p.dim() == 1:
            p.add_(0.1 * torch.randn(p.shape, generator=gen))
sd = {f"n.{k}": v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
net = net.cuda()
x, e = torch.randn(1, N, d, generator=gen), torch.randn(1, M, d, generator=gen)
cx, ce = torch.randn(1, N, d, generator=gen), torch.randn(1, M, d, generator=gen)
xc, ec = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
rx, re_ = orc.interaction_net(sd, "n", ei, xc, xc, ec)
names = [k for k, _ in net.named_parameters()]
want = torch.autograd.grad((rx * cx).sum() + (re_ * ce).sum(), [xc, ec] + [sd[f"n.{k}"] for k in names])
xg, eg = x.cuda().requires_grad_(True), e.cuda().requires_grad_(True)
ox, oe = net(xg, xg, eg)
((ox * cx.cuda()).sum() + (oe * ce.cuda()).sum()).backward()
fwd = max(rel(ox, rx), rel(oe, re_))
gin = max(rel(xg.grad, want[0]), rel(eg.grad, want[1]))
gpar = max(rel(p.grad, w) for (k, p), w in zip(net.named_parameters(), want[2:]))
print(f"full-size m2m d{d} {N} nodes {M} edges: fwd {fwd:.2e}  input grads {gin:.2e}  "
      f"param grads {gpar:.2e}")
assert fwd < fwd_bar and gin < grad_bar and gpar < grad_bar, (fwd, gin, gpar)
print("full-size case passed")
