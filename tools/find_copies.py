#!/usr/bin/env python3
"""Where do the memcpy / elementwise torch kernels of one training step come from?  Profiles one
eager step (torch.profiler, Python stacks) of the bench model and groups aten::copy_ / clone /
add / fill launches by their innermost repo frame.  usage: find_copies.py [bench.py model args]"""
import collections, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from neural_lam_amd import parallel, synthetic

sys.argv = [sys.argv[0]] + sys.argv[1:] + ["--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0)
tmp = tempfile.TemporaryDirectory()
model, info = bench.build(args, tmp.name)
model = model.to(dev)
flat = parallel.FlatParams(model)
reducer = parallel.GradAllReduce(flat)
opt = parallel.FlatAdamW(flat, lr=1e-3)
batch = synthetic.random_batch(args.batch, args.ar_steps, info["num_grid"], seed=100, device=dev)


def step():
    flat.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    reducer.reduce()
    opt.step(grad_scale=1.0)


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::clone", "aten::add_", "aten::add", "aten::fill_", "aten::zero_",
        "aten::cat", "aten::contiguous", "aten::mul", "aten::sum")
groups = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        frame = next((f for f in ev.stack if "/repo/" in f or "neural" in f), ev.stack[0] if ev.stack else "?")
        groups[(ev.name, frame.strip()[:110])] += 1
for (name, frame), c in groups.most_common(40):
    print(f"{c:5d}  {name:18s} {frame}")
