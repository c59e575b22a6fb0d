#!/usr/bin/env python3
"""Workload of tools/pmc_gather_calib.sh: back-to-back launches of (a) the library's streaming row
copy and (b) its 256-byte-row gather through a mesh-local and a random index -- kernels whose HBM
bytes are known exactly -- so that the FETCH_SIZE / WRITE_SIZE counters can be calibrated for the
access shapes of the grid-side edge kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd import ops

dev = "cuda"
gen = torch.Generator().manual_seed(0)
d = 64
out = {}
# sizes of the m2g gather: 255,136 gathered rows per sample out of a 6,561-row (L2-resident) and
# out of a 63,784-row table; B = 4; and a table far beyond L2 + Infinity Cache (2 M rows = 512 MB)
for name, N, M in (("mesh table 6561 rows", 6561, 255136), ("grid table 63784 rows", 63784, 255136),
                   ("2M-row table", 2_000_000, 255136)):
    B = 4 if N < 1_000_000 else 1
    x = torch.randn(B, N, d, device=dev)
    o = torch.empty(B, M, d, device=dev)
    near = torch.sort(torch.randint(0, N, (M,), generator=gen)).values.to(torch.int32).to(dev)
    rnd = torch.randint(0, N, (M,), generator=gen).to(torch.int32).to(dev)
    for tag, idx in (("sorted index", near), ("random index", rnd)):
        for _ in range(5):
            ops.gather_rows(ops.mat(x), idx, ops.mat(o))
        torch.cuda.synchronize()
        uniq = int(torch.unique(idx).numel())
        out[f"gather_rows {name}, {tag}"] = {
            "launches": 5, "rows_gathered": B * M, "distinct_rows": B * uniq,
            "read_bytes_if_every_row_fetched_once": B * uniq * 4 * d + 4 * M,
            "read_bytes_if_every_gather_fetched": B * M * 4 * d + 4 * M,
            "write_bytes": B * M * 4 * d}
    del x, o
src = torch.randn(4, 255136, d, device=dev)
dst = torch.empty_like(src)
for _ in range(5):
    ops.copy_rows(ops.mat(src), ops.mat(dst))
torch.cuda.synchronize()
out["copy_rows 4x255136x64"] = {"launches": 5, "read_bytes": src.numel() * 4, "write_bytes": src.numel() * 4}
print(json.dumps(out))
