"""Peak device memory of one training step (GB):  python tools/peak_mem.py [bench.py model args]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neural_lam_amd import synthetic

sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
if args.hidden_dim == 256:
    os.environ.setdefault("NLAM_MFMA", "bf16")
with tempfile.TemporaryDirectory() as tmp:
    model, info = bench.build(args, tmp)
model = model.cuda()
batch = synthetic.random_batch(args.batch, args.ar_steps, info["num_grid"], seed=100, device="cuda")
for it in range(2):
    for p in model.parameters():
        p.grad = None
    torch.cuda.reset_peak_memory_stats()
    loss = model.training_step(batch)
    loss.backward()
    torch.cuda.synchronize()
print(f"{args.model}-{args.hidden_dim} ar{args.ar_steps} B{args.batch}: peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
