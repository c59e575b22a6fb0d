#!/usr/bin/env python3
"""cProfile of the HOST side of eager training steps (the multi-rank path launches eagerly so
that gradient buckets can be all-reduced from backward hooks): where do the microseconds per
launch go?  usage: host_profile.py [bench.py model args]"""
import cProfile, os, pstats, sys, tempfile, time
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


for _ in range(3):
    step()
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    step()
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / n
print(f"host issue time {host*1e3:.2f} ms/step, wall {total*1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
