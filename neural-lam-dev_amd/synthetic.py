"""Synthetic stand-ins for the data side of the hot path: a duck-typed
datastore exposing exactly the fields ARModel.__init__ reads
(ar_model.py:40-48,54-76,121-125) and MEPS-shaped random batches
(SURVEY.md section 8d).  Used by bench.py, smoke() and the tests; there is no
network for real datasets."""
import types
from pathlib import Path

import numpy as np
import torch

from . import graphgen


class _DA:
    def __init__(self, values):
        self.values = np.asarray(values)

    def transpose(self, *dims):
        return self


class SyntheticDatastore:
    def __init__(self, root_path, static, state_mean, state_std, diff_mean, diff_std,
                 boundary_mask, n_forcing):
        self.root_path = Path(root_path)
        self._static = _DA(static)
        self._n = {"state": len(state_mean), "forcing": int(n_forcing),
                   "static": np.asarray(static).shape[1]}
        self._stats = types.SimpleNamespace(
            state_mean=_DA(state_mean), state_std=_DA(state_std),
            state_diff_mean=_DA(diff_mean), state_diff_std=_DA(diff_std),
        )
        self.boundary_mask = _DA(boundary_mask)

    def get_num_data_vars(self, category):
        return self._n[category]

    def get_vars_names(self, category):
        return [f"{category}_{i}" for i in range(self._n[category])]

    def get_dataarray(self, category, split):
        assert category == "static"
        return self._static

    def get_standardization_dataarray(self, category):
        assert category == "state"
        return self._stats


def model_args(**kw):
    """Namespace with the argparse fields the model classes read
    (train_model.py:29-209; tests/test_training.py:71-87)."""
    base = dict(graph="multiscale", hidden_dim=64, hidden_layers=1, processor_layers=4,
                mesh_aggr="sum", output_std=False, loss="wmse", lr=1e-3, restore_opt=False,
                n_example_pred=0, num_past_forcing_steps=1, num_future_forcing_steps=1,
                val_steps_to_log=[1], metrics_watch=[])
    base.update(kw)
    return types.SimpleNamespace(**base)


def meps_setup(root, nx=238, ny=268, hierarchical=False, n_levels=None, graph_name=None,
               n_state=17, n_forcing=6, n_static=4, boundary_width=10, seed=42):
    """MEPS-sized synthetic problem: 238 x 268 grid at
    10 km, multiscale or 3-level hierarchical graph in load_graph format,
    N(0,1) static features, unit statistics, 10-cell boundary frame."""
    graph_name = graph_name or ("hierarchical" if hierarchical else "multiscale")
    gdir = Path(root) / "graph" / graph_name
    info = graphgen.create_graph(str(gdir), graphgen.make_xy(nx, ny), n_levels, hierarchical)
    n_grid = info["num_grid"]
    gen = torch.Generator().manual_seed(seed)
    static = torch.randn(n_grid, n_static, generator=gen).numpy()
    mask = np.zeros((ny, nx), dtype=np.float32)
    w = boundary_width
    if w > 0:
        mask[:w, :] = mask[-w:, :] = 1
        mask[:, :w] = mask[:, -w:] = 1
    ds = SyntheticDatastore(root, static, np.zeros(n_state), np.ones(n_state), np.zeros(n_state),
                            np.ones(n_state), mask.reshape(-1), n_forcing)
    return ds, graph_name, info


def random_batch(batch_size, ar_steps, n_grid, n_state=17, n_forcing_window=18, seed=0,
                 device="cpu"):
    """(init_states (B,2,N,d), target_states (B,T,N,d), forcing (B,T,N,f), times)."""
    gen = torch.Generator().manual_seed(seed)
    init = torch.randn(batch_size, 2, n_grid, n_state, generator=gen)
    target = torch.randn(batch_size, ar_steps, n_grid, n_state, generator=gen)
    forcing = torch.randn(batch_size, ar_steps, n_grid, n_forcing_window, generator=gen)
    return init.to(device), target.to(device), forcing.to(device), None
