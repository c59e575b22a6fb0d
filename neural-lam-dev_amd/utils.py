"""Host-side mirror of the reference's neural_lam/utils.py for the hot path:
BufferList (utils.py:11-33), load_graph (utils.py:36-188) and make_mlp
(utils.py:191-214).  make_mlp returns a container with the reference's child
layout (Linear at even indices, SiLU between, LayerNorm last => identical
state_dict keys and identical default-init RNG stream) whose forward runs the
HIP kernels.
"""
import os

import torch
from torch import nn

from . import generic


class BufferList(nn.Module):
    """List of buffers b0, b1, ... (utils.py:11-33)."""

    def __init__(self, buffer_tensors, persistent=True):
        super().__init__()
        self.n_buffers = len(buffer_tensors)
        for i, t in enumerate(buffer_tensors):
            self.register_buffer(f"b{i}", t, persistent=persistent)

    def __getitem__(self, key):
        return getattr(self, f"b{key}")

    def __len__(self):
        return self.n_buffers

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def load_graph(graph_dir_path, device="cpu"):
    """Reads the 7 (+4 hierarchical) .pt files of a graph directory; all edge
    features are divided by the longest m2m edge (utils.py:104-113); flat graphs
    unwrap level 0 (utils.py:165-167).  Returns (hierarchical, dict)."""

    def ld(fn):
        return torch.load(os.path.join(graph_dir_path, fn), map_location=device, weights_only=True)

    m2m_ei = ld("m2m_edge_index.pt")
    m2m_f = ld("m2m_features.pt")
    mesh_f = ld("mesh_features.pt")
    n_levels = len(m2m_ei)
    assert len(m2m_f) == n_levels, "Inconsistent number of levels in mesh"
    assert len(mesh_f) == n_levels, "Inconsistent number of levels in mesh"
    hierarchical = n_levels > 1
    longest = max(torch.max(f[:, 0]) for f in m2m_f)
    out = {
        "g2m_edge_index": ld("g2m_edge_index.pt"),
        "m2g_edge_index": ld("m2g_edge_index.pt"),
        "g2m_features": ld("g2m_features.pt") / longest,
        "m2g_features": ld("m2g_features.pt") / longest,
    }
    if hierarchical:
        out["m2m_edge_index"] = BufferList(m2m_ei, persistent=False)
        out["m2m_features"] = BufferList([f / longest for f in m2m_f], persistent=False)
        out["mesh_static_features"] = BufferList(mesh_f, persistent=False)
        out["mesh_up_edge_index"] = BufferList(ld("mesh_up_edge_index.pt"), persistent=False)
        out["mesh_down_edge_index"] = BufferList(ld("mesh_down_edge_index.pt"), persistent=False)
        out["mesh_up_features"] = BufferList(
            [f / longest for f in ld("mesh_up_features.pt")], persistent=False
        )
        out["mesh_down_features"] = BufferList(
            [f / longest for f in ld("mesh_down_features.pt")], persistent=False
        )
    else:
        out["m2m_edge_index"] = m2m_ei[0]
        out["m2m_features"] = m2m_f[0] / longest
        out["mesh_static_features"] = mesh_f[0]
        for k in ("mesh_up_edge_index", "mesh_down_edge_index", "mesh_up_features",
                  "mesh_down_features"):
            out[k] = []
    return hierarchical, out


class HipMLP(nn.Sequential):
    """make_mlp's Sequential(Linear, SiLU, ..., Linear[, LayerNorm]) as a
    parameter container; forward = libnlam_hip.so kernels."""

    tag = "mlp"   # profiler label of this block's launches; the models set it

    def forward(self, x, res=None):
        from . import fused, ops, wide

        with ops.tag(self.tag):
            if fused.mlp_eligible(self, x):
                return fused.apply_mlp(self, x, res)
            if wide.mlp_eligible(self, x, res):
                return wide.apply_mlp(self, x, res)
            return generic.apply_mlp(self, x, res)


def make_mlp(blueprint, layer_norm=True):
    """utils.py:191-214."""
    hidden_layers = len(blueprint) - 2
    assert hidden_layers >= 0, "Invalid MLP blueprint"
    layers = []
    for layer_i, (dim1, dim2) in enumerate(zip(blueprint[:-1], blueprint[1:])):
        layers.append(nn.Linear(dim1, dim2))
        if layer_i != hidden_layers:
            layers.append(nn.SiLU())
    if layer_norm:
        layers.append(nn.LayerNorm(blueprint[-1]))
    return HipMLP(*layers)
