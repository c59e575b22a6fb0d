#!/usr/bin/env python3
"""Which torch (non C-ABI) device ops still run inside a train step, with call sites.
usage (GPU box): python tools/torch_profile.py [--model graph_lam]"""
import argparse, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="graph_lam")
ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
from neural_lam_amd import parallel, synthetic
from neural_lam_amd.models import MODELS
tmp = tempfile.mkdtemp()
hier = a.model != "graph_lam"
ds, gname, info = synthetic.meps_setup(tmp, hierarchical=hier, n_levels=3 if hier else None)
args = synthetic.model_args(graph=gname, hidden_dim=64, processor_layers=4)
dev = torch.device("cuda", 0)
model = MODELS[a.model](args, config=None, datastore=ds).to(dev)
flat = parallel.FlatParams(model)
opt = parallel.FlatAdamW(flat, lr=1e-3)
batch = synthetic.random_batch(a.batch, 1, info["num_grid"], seed=1, device=dev)

def step():
    flat.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    flat.pack_grads()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
             with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    site = ""
    for fr in (ev.stack or []):
        if "neural-lam-dev_amd" in fr or "bench.py" in fr or "torch_profile" in fr:
            site = fr.split("neural-lam-dev_amd/")[-1]
            break
    rows.append((ev.name, str(ev.input_shapes)[:70], site[:60], ev.device_time_total))
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for n, s, site, t in rows:
    k = (n, s, site)
    agg[k][0] += 1
    agg[k][1] += t
tot = 0.0
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:8.1f} us  x{c:<3d} {k[0]:22s} {k[1]:72s} {k[2]}")
    tot += t
print("total torch-op device time per step: %.1f us" % tot)
