#!/usr/bin/env python3
"""Worst relative errors of the InteractionNet operator against the reference goldens of the
wide hidden sizes (128: wide kernels; 256: generic kernels) in the MFMA mode set by NLAM_MFMA."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from neural_lam_amd._lib import lib
from neural_lam_amd.interaction_net import InteractionNet
print("mfma mode:", {0: "fp32", 1: "bf16x3", 2: "bf16"}[int(lib.nlam_mfma_mode())])


def rel(a, b):
    return float((a.detach().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "op_d*.pt"))):
    if path.endswith("_bf16.pt"):      # the autocast fixtures: tools/parity_wide.py
        continue
    fx = torch.load(path, weights_only=False)
    if fx["d"] < 128:
        continue
    net = InteractionNet(fx["edge_index"], fx["d"], **fx["kwargs"])
    net.load_state_dict(fx["state_dict"], strict=True)
    net = net.cuda()
    s = fx["send"].cuda().requires_grad_(True)
    r = s if fx["shared"] else fx["rec"].cuda().requires_grad_(True)
    e = fx["edge"].cuda().requires_grad_(True)
    o_rec, o_edge = net(s, r, e)
    ((o_rec * fx["cot_rec"].cuda()).sum() + (o_edge * fx["cot_edge"].cuda()).sum()).backward()
    fwd = max(rel(o_rec, fx["out_rec"]), rel(o_edge, fx["out_edge"]))
    gin = max(rel(s.grad, fx["grad_send"]), rel(e.grad, fx["grad_edge"]))
    gpar = max(rel(p.grad, fx["grad_params"][k]) for k, p in net.named_parameters())
    print(f"{os.path.basename(path):24s} d{fx['d']:<4d} fwd {fwd:.2e}  input grads {gin:.2e}  "
          f"param grads {gpar:.2e}")
