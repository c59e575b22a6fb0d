#!/usr/bin/env python3
"""Worst-case relative errors of the fused models against the reference goldens, for the
MFMA mode set in NLAM_MFMA (fp32 | bf16x3 | bf16).  Tolerances: prediction 1e-4, grads 2e-3
(fp32 / bf16x3); 2e-2 for the plain-bf16 mode.  Hidden widths 64 and 128."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_models import build_model, rel, MODEL_FILES
from neural_lam_amd._lib import lib
print("mfma mode:", {0: "fp32", 1: "bf16x3", 2: "bf16"}[int(lib.nlam_mfma_mode())])
for path in MODEL_FILES:
    fx = torch.load(path, weights_only=False)
    if fx["cfg"]["hidden_dim"] not in (64, 128):
        continue
    with tempfile.TemporaryDirectory() as tmp:
        model = build_model(fx, tmp).cuda()
    batch = (fx["init_states"].cuda(), fx["target_states"].cuda(), fx["forcing"].cuda(), None)
    pred, _, _, _ = model.common_step(batch)
    loss = model.training_step(batch)
    loss.backward()
    ge = max(rel(p.grad, fx["grad_params"][k]) for k, p in model.named_parameters())
    print(f"{os.path.basename(path):32s} d{fx['cfg']['hidden_dim']:<4d} pred {rel(pred, fx['prediction']):.2e}  "
          f"loss {abs(float(loss) - fx['loss']) / abs(fx['loss']):.2e}  worst grad {ge:.2e}")
