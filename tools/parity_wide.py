#!/usr/bin/env python3
"""Runs the operator / MLP parity cases of tests/test_gpu_wide.py in THIS process's arithmetic
mode and width (NLAM_MFMA, NLAM_WIDE_D) -- e.g. the hidden-256 feature-split kernels, which exist
in bf16 arithmetic only:  NLAM_MFMA=bf16 NLAM_WIDE_D=256 python tools/parity_wide.py
Prints one `ok <case>` line per case; exits non-zero on the first failure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import neural_lam_amd  # noqa: F401  (binds libnlam_hip.so)
from neural_lam_amd._lib import lib
import test_gpu_wide as T

print("mfma mode:", {0: "fp32", 1: "bf16x3", 2: "bf16"}[int(lib.nlam_mfma_mode())], "width:", T.D)
for shared, upd, aggr, B in T.INET_CASES + [(True, True, "mean", 5)]:
    T.test_wide_interaction_net_vs_oracle(shared, upd, aggr, B)
    print(f"ok inet shared={shared} upd={upd} aggr={aggr} B={B}", flush=True)
T.test_wide_stride0_batch_inputs_match_oracle()
print("ok inet stride-0 inputs", flush=True)
for blueprint, ln, res, rows, B in T.MLP_CASES:
    T.test_wide_mlp_vs_oracle(blueprint, ln, res, rows, B)
    print(f"ok mlp {blueprint} ln={ln} res={res} rows={rows} B={B}", flush=True)
print("all wide cases passed")
