#!/usr/bin/env python3
"""Where do the device-to-device copies of a training step come from?  Runs a few eager steps of
a bench model under torch.profiler and prints, for every CPU op that launched a `Memcpy DtoD` (or
a copy / fill kernel), its name, input shapes and the innermost Python frames.

  python tools/find_copies.py [--model graph_lam|hi_lam] [--hidden-dim 64]"""
import argparse
import collections
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="graph_lam")
    ap.add_argument("--hidden-dim", type=int, default=64)
    ap.add_argument("--processor-layers", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4)
    args = ap.parse_args()
    from neural_lam_amd import parallel, synthetic

    dev = torch.device("cuda", 0)
    with tempfile.TemporaryDirectory() as tmp:
        model, info = bench.build(args, tmp)
    model = model.to(dev)
    flat = parallel.FlatParams(model)
    batch = synthetic.random_batch(args.batch, 1, info["num_grid"], seed=100, device=dev)

    def step():
        flat.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        flat.pack_grads()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    dev_copies = collections.Counter(ev.name[:60] for ev in prof.events()
                                     if ev.device_type.name != "CPU" and ("Memcpy" in ev.name or "copy" in ev.name.lower()))
    print("device copy events:", dict(dev_copies))
    rows = collections.Counter()
    for ev in prof.events():
        if ev.device_type.name == "CPU" and ev.name in ("aten::copy_", "aten::clone", "aten::cat", "aten::contiguous",
                                                        "aten::fill_", "aten::zero_", "aten::add", "aten::add_"):
            stack = [f for f in (ev.stack or []) if "neural_lam" in f or "bench" in f or "tools/" in f][:2]
            rows[(ev.name, str(ev.input_shapes)[:90], " <- ".join(stack)[:200])] += 1
    for (op, shapes, stack), n in sorted(rows.items(), key=lambda kv: -kv[1]):
        print(f"{n:4d}  {op:18s} {shapes:90s} {stack}")


if __name__ == "__main__":
    main()
