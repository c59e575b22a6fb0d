from .ar_model import ARModel  # noqa: F401
from .base_graph_model import BaseGraphModel  # noqa: F401
from .base_hi_graph_model import BaseHiGraphModel  # noqa: F401
from .graph_lam import GraphLAM  # noqa: F401
from .hi_lam import HiLAM  # noqa: F401
from .hi_lam_parallel import HiLAMParallel  # noqa: F401

MODELS = {"graph_lam": GraphLAM, "hi_lam": HiLAM, "hi_lam_parallel": HiLAMParallel}
