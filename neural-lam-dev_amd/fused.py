"""Fused gfx950 kernels for the BASELINE shapes (hidden_layers == 1,
hidden_dim in {64, 128}).  Until a shape is covered the callers use the generic
kernel sequences (generic.py); both are HIP."""
import os

FORCE_GENERIC = os.environ.get("NLAM_FORCE_GENERIC", "0") == "1"


def mlp_eligible(seq, x):
    return False


def inet_eligible(net, send_rep, rec_rep, edge_rep):
    return False
