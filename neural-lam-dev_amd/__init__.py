"""MI355X-native InteractionNet / GraphLAM / Hi-LAM hot path.

Importing the package binds libnlam_hip.so (see _lib.py) and raises if it is
missing: there is no CPU or eager fallback."""
from . import _lib  # noqa: F401
from .interaction_net import InteractionNet, SplitMLPs  # noqa: F401
from .utils import make_mlp, load_graph  # noqa: F401


# ---- arithmetic of a run (the reference's `--precision`, train_model.py:72-77,285) ----------
_PRECISIONS = {
    # Lightning names the reference accepts -> GEMM arithmetic of the HIP kernels
    "32": 1, "32-true": 1,            # fp32-grade: split-bf16 products (~2^-16), fp32 everywhere else
    "bf16-mixed": 2, "bf16": 2,       # bf16 products, fp32 accumulate / storage / LayerNorm
    "bf16x3": 1, "fp32-exact": 0, "fp32": 0,   # explicit names of the two fp32-grade forms
}


def set_precision(precision):
    """Select the GEMM arithmetic for the runs that follow (process-wide; call it before a model
    runs, not between a forward and its backward).  Returns the previous setting's name."""
    key = str(precision).lower()
    if key not in _PRECISIONS:
        raise ValueError(f"precision {precision!r} not one of {sorted(_PRECISIONS)}")
    prev = get_precision()
    _lib.check(_lib.lib.nlam_set_mfma_mode(_PRECISIONS[key]), "nlam_set_mfma_mode")
    return prev


def get_precision():
    return {0: "fp32-exact", 1: "32-true", 2: "bf16-mixed"}[int(_lib.lib.nlam_mfma_mode())]
