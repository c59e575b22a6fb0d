#!/usr/bin/env python3
"""What this MI355X box actually delivers for the access patterns of the hot path, next to the
8 TB/s the roofline is priced against: (a) a streaming device-to-device copy (torch copy_ =
hipMemcpyAsync, and the library's own row copy), (b) a gather of 256-byte rows (hidden 64; 512 /
1024 bytes at 128 / 256) through a receiver-sorted sender index like the m2m layer's, (c) the
library's segment-sum on the same table.  50 launches between one HIP event pair each.
Prints JSON: {"case": GB/s}."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd import ops

dev = "cuda"
gen = torch.Generator().manual_seed(0)
B, N, M = 4, 26244, 230304          # GraphLAM-like m2m layer: nodes, edges (synthetic MEPS mesh)
res = {}


def timed(fn, nbytes, name, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    res[name] = {"us": round(us, 1), "GB/s": round(nbytes / us / 1e3, 0)}


for d in (64, 128, 256):
    src = torch.randn(B, M, d, device=dev)
    dst = torch.empty(B, M, d, device=dev)
    timed(lambda: dst.copy_(src), 8.0 * B * M * d, f"stream copy (hipMemcpy) {B}x{M}x{d}")
    timed(lambda: ops.copy_rows(ops.mat(src), ops.mat(dst)), 8.0 * B * M * d,
          f"stream copy (nlam_copy_rows) {B}x{M}x{d}")
    # receiver-sorted edges: sender of edge k is a random node near its receiver (mesh locality)
    rec = torch.sort(torch.randint(0, N, (M,), generator=gen)).values
    send = ((rec + torch.randint(-200, 200, (M,), generator=gen)) % N).to(torch.int32).to(dev)
    rnd = torch.randint(0, N, (M,), generator=gen).to(torch.int32).to(dev)
    x = torch.randn(B, N, d, device=dev)
    out = torch.empty(B, M, d, device=dev)
    # algorithmic bytes: every source row once + every gathered row written + the index
    nb = B * (4.0 * d * (N + M) + 4.0 * M)
    timed(lambda: ops.gather_rows(ops.mat(x), send, ops.mat(out)), nb, f"gather rows, mesh-local index, d={d}")
    timed(lambda: ops.gather_rows(ops.mat(x), rnd, ops.mat(out)), nb, f"gather rows, random index, d={d}")
    # segment-sum of the M edge rows into N receivers
    rowptr = torch.zeros(N + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(torch.bincount(rec, minlength=N), 0).to(torch.int32)
    rowptr = rowptr.to(dev)
    pos = torch.arange(M, dtype=torch.int32, device=dev)
    agg = torch.empty(B, N, d, device=dev)
    timed(lambda: ops.segment_sum(ops.mat(out), rowptr, pos, ops.mat(agg)),
          B * (4.0 * d * (M + N) + 4.0 * M + 4.0 * (N + 1)), f"segment sum, d={d}")
    del src, dst, x, out, agg
print(json.dumps(res, indent=1))
