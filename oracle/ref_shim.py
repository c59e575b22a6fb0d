"""
TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Loads the reference's own hot-path source files from /root/reference (where
they lie; nothing is copied) under import stand-ins for the third-party
packages that are not installed in the build container (torch_geometric,
pytorch_lightning, wandb, xarray, tueplots, ...).  This is what pins the
oracle (`oracle/nlam_oracle.py`) and generates the golden vectors under
`tests/golden/` (see `tests/golden/make_golden.py`).

The reference cannot travel to the GPU box: nothing under `-m gpu`, `smoke()`
or `bench.py` imports this module.  `available()` returns False when
/root/reference is absent, and every caller skips in that case.

Stand-in semantics (restated from the published behaviour of
torch-geometric==2.3.1, pinned in the reference's pyproject.toml:26):
  * MessagePassing.propagate(edge_index, x, edge_attr): x_j = x.index_select(
    -2, edge_index[0]); x_i = x.index_select(-2, edge_index[1]); out =
    message(x_j=, x_i=, edge_attr=); aggregate(out, edge_index[1], None,
    x.size(-2)).
  * MessagePassing.aggregate(inputs, index, ptr, dim_size): "sum" =
    zeros(dim_size).scatter_add_(-2, index, inputs); "mean" = sum /
    clamp(in_degree, min=1).
  * nn.Sequential(input_args, [(module, "a, b -> c, d"), ...]) registers
    children as module_<i> and threads named values through them.
"""
import importlib
import os
import sys
import types

import torch
from torch import nn

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "neural_lam"))


class _MessagePassing(nn.Module):
    def __init__(self, aggr="add", node_dim=-2, **kwargs):
        super().__init__()
        self.aggr = aggr
        self.node_dim = node_dim

    def propagate(self, edge_index, x=None, edge_attr=None, **kwargs):
        x_j = x.index_select(self.node_dim, edge_index[0])
        x_i = x.index_select(self.node_dim, edge_index[1])
        out = self.message(x_j=x_j, x_i=x_i, edge_attr=edge_attr)
        return self.aggregate(out, edge_index[1], None, x.size(self.node_dim))

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        dim_size = int(dim_size)
        size = list(inputs.shape)
        size[self.node_dim] = dim_size
        shape = [1] * inputs.dim()
        shape[self.node_dim] = -1
        idx = index.view(shape).expand_as(inputs)
        out = inputs.new_zeros(size).scatter_add_(self.node_dim, idx, inputs)
        if self.aggr == "mean":
            count = inputs.new_zeros(dim_size).scatter_add_(
                0, index, inputs.new_ones(index.shape[0])
            )
            out = out / count.clamp(min=1).view(shape)
        elif self.aggr not in ("sum", "add"):
            raise ValueError(self.aggr)
        return out


class _PygSequential(nn.Module):
    def __init__(self, input_args, modules):
        super().__init__()
        self._in = [a.strip() for a in input_args.split(",")]
        self._specs = []
        for i, (module, desc) in enumerate(modules):
            lhs, rhs = desc.split("->")
            self.add_module(f"module_{i}", module)
            self._specs.append(
                (
                    f"module_{i}",
                    [a.strip() for a in lhs.split(",")],
                    [a.strip() for a in rhs.split(",")],
                )
            )

    def forward(self, *args):
        env = dict(zip(self._in, args))
        out = None
        for name, lhs, rhs in self._specs:
            out = getattr(self, name)(*[env[a] for a in lhs])
            if not isinstance(out, tuple):
                out = (out,)
            env.update(zip(rhs, out))
        return out if len(out) > 1 else out[0]


class _LightningModule(nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Data:
    """Attribute bag standing in for torch_geometric.data.Data (only what create_graph.py
    touches: attribute access, assignment, clone())."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def clone(self):
        return _Data(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})


def _from_networkx(G):
    """torch_geometric.utils.convert.from_networkx of torch-geometric 2.3.1, restated: nodes
    are numbered in G.nodes() order, edges listed in G.edges() order, node / edge attributes
    collected per key in that order and turned into tensors (torch.stack for tensors,
    torch.tensor otherwise: numpy float64 attributes give float64 tensors)."""
    import networkx as nx
    from collections import defaultdict

    G = G.to_directed() if not nx.is_directed(G) else G
    mapping = dict(zip(G.nodes(), range(G.number_of_nodes())))
    edge_index = torch.empty((2, G.number_of_edges()), dtype=torch.long)
    for i, (src, dst) in enumerate(G.edges()):
        edge_index[0, i] = mapping[src]
        edge_index[1, i] = mapping[dst]
    data = defaultdict(list)
    node_attrs = list(next(iter(G.nodes(data=True)))[-1].keys()) if G.number_of_nodes() else []
    for _, feat in G.nodes(data=True):
        if set(feat.keys()) != set(node_attrs):
            raise ValueError("Not all nodes contain the same attributes")
        for k, v in feat.items():
            data[str(k)].append(v)
    edge_attrs = list(next(iter(G.edges(data=True)))[-1].keys()) if G.number_of_edges() else []
    for _, _, feat in G.edges(data=True):
        if set(feat.keys()) != set(edge_attrs):
            raise ValueError("Not all edges contain the same attributes")
        for k, v in feat.items():
            k = f"edge_{k}" if k in node_attrs else k
            data[str(k)].append(v)
    out = {}
    for k, v in data.items():
        if isinstance(v, (tuple, list)) and torch.is_tensor(v[0]):
            out[k] = torch.stack(v, dim=0)
        else:
            import numpy as np

            out[k] = torch.tensor(np.asarray(v))
    out["edge_index"] = edge_index.view(2, -1)
    d = _Data(**out)
    d.num_nodes = G.number_of_nodes()
    return d


def load_create_graph():
    """The reference's graph-creation tool (neural_lam/create_graph.py) under stand-ins for
    torch_geometric (from_networkx restated above; the pyg.utils calls are plot-only) and the
    config / datastore modules its CLI wrapper imports.  Returns the module."""
    load()
    utils_mod = _mod("torch_geometric.utils")
    conv = _mod("torch_geometric.utils.convert", from_networkx=_from_networkx)
    utils_mod.convert = conv
    sys.modules["torch_geometric"].utils = utils_mod
    sys.modules["neural_lam.config"].load_config_and_datastore = None
    base = _mod("neural_lam.datastore.base", BaseRegularGridDatastore=object)
    sys.modules["neural_lam.datastore"].base = base
    sys.modules["neural_lam.datastore"].__path__ = []
    import matplotlib

    matplotlib.use("Agg")
    return importlib.import_module("neural_lam.create_graph")


_loaded = None


def load():
    """Import the reference hot-path modules; returns a namespace of them."""
    global _loaded
    if _loaded is not None:
        return _loaded
    assert available(), "reference tree not present"

    # Parent packages without running neural_lam/__init__.py (it eagerly
    # imports datastores / W&B / Lightning that are not installed).
    pkg = _mod("neural_lam")
    pkg.__path__ = [os.path.join(REF_ROOT, "neural_lam")]
    mpkg = _mod("neural_lam.models")
    mpkg.__path__ = [os.path.join(REF_ROOT, "neural_lam", "models")]

    # third-party stand-ins
    _mod("tueplots", bundles=types.SimpleNamespace(), figsizes=types.SimpleNamespace())
    pyg_nn = _mod(
        "torch_geometric.nn", MessagePassing=_MessagePassing, Sequential=_PygSequential
    )
    _mod("torch_geometric", nn=pyg_nn)
    _mod(
        "pytorch_lightning",
        LightningModule=_LightningModule,
        LightningDataModule=object,
    )
    _mod("wandb")
    _mod("xarray", DataArray=object, Dataset=object)
    if "matplotlib" not in sys.modules:
        try:
            importlib.import_module("matplotlib.pyplot")
        except Exception:  # pragma: no cover
            _mod("matplotlib", pyplot=types.ModuleType("pyplot"))
            _mod("matplotlib.pyplot")

    # first-party modules off the hot path, as empty stand-ins
    _mod("neural_lam.vis")
    _mod("neural_lam.weather_dataset", WeatherDataset=object)
    _mod("neural_lam.config", NeuralLAMConfig=object)
    _mod("neural_lam.datastore", BaseDatastore=object)

    def get_state_feature_weighting(config, datastore):
        # uniform case of loss_weighting.py:52-71
        n = datastore.get_num_data_vars(category="state")
        return [1.0 / n] * n

    _mod(
        "neural_lam.loss_weighting",
        get_state_feature_weighting=get_state_feature_weighting,
    )

    ns = types.SimpleNamespace()
    ns.utils = importlib.import_module("neural_lam.utils")
    ns.interaction_net = importlib.import_module("neural_lam.interaction_net")
    ns.metrics = importlib.import_module("neural_lam.metrics")
    ns.ar_model = importlib.import_module("neural_lam.models.ar_model")
    ns.base_graph_model = importlib.import_module(
        "neural_lam.models.base_graph_model"
    )
    ns.graph_lam = importlib.import_module("neural_lam.models.graph_lam")
    ns.base_hi_graph_model = importlib.import_module(
        "neural_lam.models.base_hi_graph_model"
    )
    ns.hi_lam = importlib.import_module("neural_lam.models.hi_lam")
    ns.hi_lam_parallel = importlib.import_module(
        "neural_lam.models.hi_lam_parallel"
    )
    _loaded = ns
    return ns
