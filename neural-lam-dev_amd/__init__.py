"""MI355X-native InteractionNet / GraphLAM / Hi-LAM hot path.

Importing the package binds libnlam_hip.so (see _lib.py) and raises if it is
missing: there is no CPU or eager fallback."""
from . import _lib  # noqa: F401
from .interaction_net import InteractionNet, SplitMLPs  # noqa: F401
from .utils import make_mlp, load_graph  # noqa: F401
