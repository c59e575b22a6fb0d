#!/usr/bin/env python3
"""One full-size training step (MEPS 238 x 268 grid, B = 1, ar_steps = 1) of a BASELINE model in
THIS process's arithmetic mode against the CPU oracle: loss and every parameter gradient.
    python tools/parity_fullmodel.py graph_lam 64          # configs[1]
    python tools/parity_fullmodel.py hi_lam 128            # configs[2]
    NLAM_MFMA=bf16 python tools/parity_fullmodel.py hi_lam 256   # configs[4]
Bars: fp32-grade modes loss 1e-4 / gradients 2e-3; bf16 1e-2 / 5e-2 (relative to max|ref| per tensor)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import nlam_oracle as orc
from neural_lam_amd import synthetic
from neural_lam_amd._lib import lib
from neural_lam_amd.models import MODELS

kind, hidden = sys.argv[1], int(sys.argv[2])
mode = {0: "fp32", 1: "bf16x3", 2: "bf16"}[int(lib.nlam_mfma_mode())]
loss_bar, grad_bar = (1e-2, 5e-2) if mode == "bf16" else (1e-4, 2e-3)
print("mfma mode:", mode, "model:", kind, "hidden:", hidden, flush=True)


def rel(a, b):
    return float((a.detach().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30))


hier = kind != "graph_lam"
with tempfile.TemporaryDirectory() as tmp:
    ds, gname, info = synthetic.meps_setup(tmp, hierarchical=hier, n_levels=3 if hier else None)
    torch.manual_seed(42)
    model = MODELS[kind](synthetic.model_args(graph=gname, hidden_dim=hidden, processor_layers=4),
                         config=None, datastore=ds)
    _, graph = orc.load_graph(tmp + "/graph/" + gname)
sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()
      if v.dtype.is_floating_point}
data = {k: getattr(model, k).detach().clone() for k in
        ("grid_static_features", "diff_mean", "diff_std", "boundary_mask", "per_var_std")}
batch = synthetic.random_batch(1, 1, info["num_grid"], seed=7)
cfg = {"model": kind, "hidden_layers": 1, "processor_layers": 4, "mesh_aggr": "sum", "loss": "wmse"}
want, _ = orc.training_loss(sd, graph, cfg, data, batch[0], batch[1], batch[2])
names = [k for k, _ in model.named_parameters()]
grads = torch.autograd.grad(want, [sd[k] for k in names])
model = model.cuda()
loss = model.training_step(tuple(t.cuda() if t is not None else None for t in batch))
loss.backward()
lerr = abs(float(loss) - float(want)) / abs(float(want))
worst, wname = max((rel(p.grad, g), k) for (k, p), g in zip(model.named_parameters(), grads))
print(f"full-size {kind}-{hidden}: {info['num_grid']} grid nodes, {len(names)} parameter tensors: "
      f"loss rel {lerr:.2e}  worst parameter gradient {worst:.2e} ({wname})", flush=True)
assert lerr < loss_bar and worst < grad_bar, (lerr, worst, wname)
print("full-size model case passed")
