#!/usr/bin/env python3
"""Phase shares of nlam_mlp_bwd on grid-sized rows (instrumented: NLAM_STAMP=1)."""
import ctypes, os, sys
os.environ["NLAM_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_lam_amd import utils
from neural_lam_amd._lib import lib

names = ["stage x, gy", "GEMM1 + silu", "GEMM2 + LN bwd + colsums", "planes + db2 + dW2",
         "W2^T gz + silu' + GA tile", "X again + db1 + dW1 / ga store", "W1^T ga + stores"]
buf = (ctypes.c_ulonglong * 8)()
for blueprint, ln, res, label in [([64, 64, 64], True, True, "encoding_grid_mlp (K=64, LN, residual)"),
                                  ([56, 64, 64], True, False, "grid_embedder (K=56)"),
                                  ([64, 64, 17], False, False, "output_map (17 out, no LN)")]:
    mlp = utils.make_mlp(blueprint, layer_norm=ln).cuda()
    x = torch.randn(4, 63784, blueprint[0], device="cuda", requires_grad=True)
    lib.nlam_debug_mlp_bwd_stamps(buf, 1)
    for it in range(3):
        y = mlp(x, res=x) if res else mlp(x)
        y.sum().backward()
        torch.cuda.synchronize()
        lib.nlam_debug_mlp_bwd_stamps(buf, 1)
    vals = [buf[i] for i in range(7)]
    tot = sum(vals)
    ntiles = 4 * ((63784 + 31) // 32)
    print(label)
    for n, v in zip(names, vals):
        print(f"  {n:34s} {100*v/max(tot,1):5.1f} %   {v/ntiles:9.0f} cycles/tile")
    print("  total cycles/tile", tot / ntiles)
