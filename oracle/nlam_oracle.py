"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain fp32 torch ops, functional style over a flat
``state_dict``) of the reference's InteractionNet / GraphLAM / Hi-LAM hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; the product package
(``neural-lam-dev_amd/``) never does and has no CPU fallback.

Parity status: PINNED.  ``tests/golden/make_golden.py`` runs the reference's own
source files (under the import stand-ins of ``oracle/ref_shim.py``) and this
oracle on identical inputs and commits the reference's outputs as golden
vectors; ``tests/test_oracle_golden.py`` checks this file against them and,
when /root/reference is present, against the live reference.

Third-party arithmetic restated here (not under /root/reference):
torch-geometric==2.3.1 (reference pyproject.toml:26) MessagePassing: gather =
``index_select`` on dim -2, "sum" = ``scatter_add_`` into zeros(num_rec),
"mean" = sum / clamp(in_degree, min=1).

Every function cites the reference file:line it follows.  Parameter names are
the reference's state_dict keys, so a reference checkpoint evaluates directly.
"""
import os

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# MLP blocks
# --------------------------------------------------------------------------
def mlp(sd, prefix, x, hidden_layers=1, layer_norm=True):
    """utils.py:191-214 make_mlp: Linear at index 2i (i=0..L), SiLU between,
    LayerNorm(eps=1e-5, affine) at index 2L+1."""
    for i in range(hidden_layers + 1):
        x = F.linear(x, sd[f"{prefix}.{2 * i}.weight"], sd[f"{prefix}.{2 * i}.bias"])
        if i != hidden_layers:
            x = F.silu(x)
    if layer_norm:
        k = 2 * hidden_layers + 1
        x = F.layer_norm(
            x, (x.shape[-1],), sd[f"{prefix}.{k}.weight"], sd[f"{prefix}.{k}.bias"], 1e-5
        )
    return x


def split_mlps(sd, prefix, x, chunk_sizes, hidden_layers=1):
    """interaction_net.py:149-163 SplitMLPs.forward: split rows (dim -2) into
    contiguous chunks, one MLP per chunk (keys ``<prefix>.mlps.<c>.<i>``)."""
    outs = []
    start = 0
    for c, n in enumerate(chunk_sizes):
        outs.append(
            mlp(sd, f"{prefix}.mlps.{c}", x[..., start : start + n, :], hidden_layers)
        )
        start += n
    return torch.cat(outs, dim=-2)


# --------------------------------------------------------------------------
# InteractionNet
# --------------------------------------------------------------------------
def normalise_edge_index(edge_index):
    """interaction_net.py:56-61: both rows start at 0; num_rec = max(rec)+1;
    senders re-based after the receivers.  Returns (sender_local, receiver,
    num_rec) with senders in [0, N_s) (i.e. *without* the +num_rec offset,
    which only addresses the [rec; send] concatenation of :102)."""
    ei = edge_index - edge_index.min(dim=1, keepdim=True)[0]
    num_rec = int(ei[1].max()) + 1
    return ei[0].clone(), ei[1].clone(), num_rec


def interaction_net(
    sd,
    prefix,
    edge_index,
    send_rep,
    rec_rep,
    edge_rep,
    update_edges=True,
    aggr="sum",
    hidden_layers=1,
    edge_chunk_sizes=None,
    aggr_chunk_sizes=None,
):
    """interaction_net.py:86-131 forward/message/aggregate.
    message input order is (edge, sender x_j, receiver x_i) (:121)."""
    send, rec, num_rec = normalise_edge_index(edge_index)
    assert rec_rep.shape[-2] == num_rec
    x_j = send_rep.index_select(-2, send)
    x_i = rec_rep.index_select(-2, rec)
    cat = torch.cat((edge_rep, x_j, x_i), dim=-1)
    if edge_chunk_sizes is None:
        msg = mlp(sd, f"{prefix}.edge_mlp", cat, hidden_layers)
    else:
        msg = split_mlps(sd, f"{prefix}.edge_mlp", cat, edge_chunk_sizes, hidden_layers)
    size = list(msg.shape)
    size[-2] = num_rec
    idx = rec.view([1] * (msg.dim() - 2) + [-1, 1]).expand_as(msg)
    agg = msg.new_zeros(size).scatter_add_(-2, idx, msg)
    if aggr == "mean":
        deg = torch.zeros(num_rec, dtype=msg.dtype).scatter_add_(
            0, rec, torch.ones(rec.shape[0], dtype=msg.dtype)
        )
        agg = agg / deg.clamp(min=1).view(-1, 1)
    else:
        assert aggr == "sum"
    cat2 = torch.cat((rec_rep, agg), dim=-1)
    if aggr_chunk_sizes is None:
        diff = mlp(sd, f"{prefix}.aggr_mlp", cat2, hidden_layers)
    else:
        diff = split_mlps(sd, f"{prefix}.aggr_mlp", cat2, aggr_chunk_sizes, hidden_layers)
    rec_out = rec_rep + diff
    if update_edges:
        return rec_out, edge_rep + msg
    return rec_out


# --------------------------------------------------------------------------
# graph files
# --------------------------------------------------------------------------
def load_graph(graph_dir):
    """utils.py:36-188 load_graph: 7(+4) .pt files; every edge-feature tensor is
    divided by the longest m2m edge (col 0) (:104-113); flat graphs unwrap
    level 0 (:165-167)."""

    def ld(fn):
        return torch.load(os.path.join(graph_dir, fn), map_location="cpu", weights_only=True)

    g = {}
    m2m_ei = ld("m2m_edge_index.pt")
    m2m_f = ld("m2m_features.pt")
    longest = max(float(f[:, 0].max()) for f in m2m_f)
    g["g2m_edge_index"] = ld("g2m_edge_index.pt")
    g["m2g_edge_index"] = ld("m2g_edge_index.pt")
    g["g2m_features"] = ld("g2m_features.pt") / longest
    g["m2g_features"] = ld("m2g_features.pt") / longest
    mesh_f = ld("mesh_features.pt")
    hierarchical = len(m2m_ei) > 1
    if hierarchical:
        g["m2m_edge_index"] = list(m2m_ei)
        g["m2m_features"] = [f / longest for f in m2m_f]
        g["mesh_static_features"] = list(mesh_f)
        g["mesh_up_edge_index"] = list(ld("mesh_up_edge_index.pt"))
        g["mesh_down_edge_index"] = list(ld("mesh_down_edge_index.pt"))
        g["mesh_up_features"] = [f / longest for f in ld("mesh_up_features.pt")]
        g["mesh_down_features"] = [f / longest for f in ld("mesh_down_features.pt")]
    else:
        g["m2m_edge_index"] = m2m_ei[0]
        g["m2m_features"] = m2m_f[0] / longest
        g["mesh_static_features"] = mesh_f[0]
    return hierarchical, g


# --------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------
def _expand(x, batch):
    """ar_model.py:204-209 expand_to_batch (stride-0 view)."""
    return x.unsqueeze(0).expand(batch, -1, -1)


def graphlam_process(sd, g, mesh_rep, cfg):
    """graph_lam.py:73-91 process_step: embed m2m edges, then
    processor_layers x InteractionNet with send = rec = mesh (:51-57)."""
    hl = cfg["hidden_layers"]
    e = _expand(mlp(sd, "m2m_embedder", g["m2m_features"], hl), mesh_rep.shape[0])
    for i in range(cfg["processor_layers"]):
        mesh_rep, e = interaction_net(
            sd,
            f"processor.module_{i}",
            g["m2m_edge_index"],
            mesh_rep,
            mesh_rep,
            e,
            update_edges=True,
            aggr=cfg.get("mesh_aggr", "sum"),
            hidden_layers=hl,
        )
    return mesh_rep


def hilam_process(sd, g, mesh_rep, cfg):
    """base_hi_graph_model.py:124-217 process_step + hi_lam.py:82-207
    (mesh_down_step / mesh_up_step / hi_processor_step)."""
    hl = cfg["hidden_layers"]
    B = mesh_rep.shape[0]
    L = len(g["mesh_static_features"])
    # base_hi_graph_model.py:137-164 embed upper levels + all edge sets
    nodes = [mesh_rep] + [
        _expand(mlp(sd, f"mesh_embedders.{l}", g["mesh_static_features"][l], hl), B)
        for l in range(1, L)
    ]
    same = [
        _expand(mlp(sd, f"mesh_same_embedders.{l}", g["m2m_features"][l], hl), B)
        for l in range(L)
    ]
    up = [
        _expand(mlp(sd, f"mesh_up_embedders.{l}", g["mesh_up_features"][l], hl), B)
        for l in range(L - 1)
    ]
    down = [
        _expand(mlp(sd, f"mesh_down_embedders.{l}", g["mesh_down_features"][l], hl), B)
        for l in range(L - 1)
    ]
    # :168-187 mesh init (up-sweep)
    for l in range(1, L):
        nodes[l], up[l - 1] = interaction_net(
            sd, f"mesh_init_gnns.{l - 1}", g["mesh_up_edge_index"][l - 1],
            nodes[l - 1], nodes[l], up[l - 1], hidden_layers=hl,
        )
    # hi_lam.py:165-207
    for p in range(cfg["processor_layers"]):
        # down sweep, hi_lam.py:82-124
        nodes[L - 1], same[L - 1] = interaction_net(
            sd, f"mesh_down_same_gnns.{p}.{L - 1}", g["m2m_edge_index"][L - 1],
            nodes[L - 1], nodes[L - 1], same[L - 1], hidden_layers=hl,
        )
        for l in range(L - 2, -1, -1):
            new, down[l] = interaction_net(
                sd, f"mesh_down_gnns.{p}.{l}", g["mesh_down_edge_index"][l],
                nodes[l + 1], nodes[l], down[l], hidden_layers=hl,
            )
            nodes[l], same[l] = interaction_net(
                sd, f"mesh_down_same_gnns.{p}.{l}", g["m2m_edge_index"][l],
                new, new, same[l], hidden_layers=hl,
            )
        # up sweep, hi_lam.py:126-163
        nodes[0], same[0] = interaction_net(
            sd, f"mesh_up_same_gnns.{p}.0", g["m2m_edge_index"][0],
            nodes[0], nodes[0], same[0], hidden_layers=hl,
        )
        for l in range(1, L):
            new, up[l - 1] = interaction_net(
                sd, f"mesh_up_gnns.{p}.{l - 1}", g["mesh_up_edge_index"][l - 1],
                nodes[l - 1], nodes[l], up[l - 1], hidden_layers=hl,
            )
            nodes[l], same[l] = interaction_net(
                sd, f"mesh_up_same_gnns.{p}.{l}", g["m2m_edge_index"][l],
                new, new, same[l], hidden_layers=hl,
            )
    # base_hi_graph_model.py:196-214 read-out (down-sweep, update_edges=False)
    for l in range(L - 2, -1, -1):
        nodes[l] = interaction_net(
            sd, f"mesh_read_gnns.{l}", g["mesh_down_edge_index"][l],
            nodes[l + 1], nodes[l], down[l], update_edges=False, hidden_layers=hl,
        )
    return nodes[0]


def hilam_parallel_process(sd, g, mesh_rep, cfg):
    """base_hi_graph_model.py:124-217 with hi_lam_parallel.py:26-99: one
    InteractionNet over the union of all mesh edges, SplitMLPs per edge set /
    per level."""
    hl = cfg["hidden_layers"]
    B = mesh_rep.shape[0]
    L = len(g["mesh_static_features"])
    nodes = [mesh_rep] + [
        _expand(mlp(sd, f"mesh_embedders.{l}", g["mesh_static_features"][l], hl), B)
        for l in range(1, L)
    ]
    same = [
        _expand(mlp(sd, f"mesh_same_embedders.{l}", g["m2m_features"][l], hl), B)
        for l in range(L)
    ]
    up = [
        _expand(mlp(sd, f"mesh_up_embedders.{l}", g["mesh_up_features"][l], hl), B)
        for l in range(L - 1)
    ]
    down = [
        _expand(mlp(sd, f"mesh_down_embedders.{l}", g["mesh_down_features"][l], hl), B)
        for l in range(L - 1)
    ]
    for l in range(1, L):
        nodes[l], up[l - 1] = interaction_net(
            sd, f"mesh_init_gnns.{l - 1}", g["mesh_up_edge_index"][l - 1],
            nodes[l - 1], nodes[l], up[l - 1], hidden_layers=hl,
        )
    ei_list = (
        list(g["m2m_edge_index"])
        + list(g["mesh_up_edge_index"])
        + list(g["mesh_down_edge_index"])
    )
    total_ei = torch.cat(ei_list, dim=1)
    sections = [ei.shape[1] for ei in ei_list]
    level_sizes = [f.shape[0] for f in g["mesh_static_features"]]
    x = torch.cat(nodes, dim=1)
    e = torch.cat(same + up + down, dim=1)
    for p in range(cfg["processor_layers"]):
        x, e = interaction_net(
            sd, f"processor.module_{p}", total_ei, x, x, e, hidden_layers=hl,
            edge_chunk_sizes=sections, aggr_chunk_sizes=level_sizes,
        )
    nodes = list(torch.split(x, level_sizes, dim=1))
    parts = torch.split(e, sections, dim=1)
    down = list(parts[2 * L - 1 :])
    for l in range(L - 2, -1, -1):
        nodes[l] = interaction_net(
            sd, f"mesh_read_gnns.{l}", g["mesh_down_edge_index"][l],
            nodes[l + 1], nodes[l], down[l], update_edges=False, hidden_layers=hl,
        )
    return nodes[0]


_PROCESSORS = {
    "graph_lam": graphlam_process,
    "hi_lam": hilam_process,
    "hi_lam_parallel": hilam_parallel_process,
}


def predict_step(sd, g, cfg, data, prev_state, prev_prev_state, forcing):
    """base_graph_model.py:106-177 predict_step.
    ``data``: dict with grid_static_features (N,ds), diff_mean, diff_std (d_f,).
    ``cfg``: model ("graph_lam"|"hi_lam"|"hi_lam_parallel"), hidden_layers,
    processor_layers, mesh_aggr, output_std."""
    hl = cfg["hidden_layers"]
    B = prev_state.shape[0]
    hier = cfg["model"] != "graph_lam"
    grid_features = torch.cat(
        (prev_state, prev_prev_state, forcing, _expand(data["grid_static_features"], B)),
        dim=-1,
    )
    grid_emb = mlp(sd, "grid_embedder", grid_features, hl)
    g2m_emb = mlp(sd, "g2m_embedder", g["g2m_features"], hl)
    m2g_emb = mlp(sd, "m2g_embedder", g["m2g_features"], hl)
    if hier:  # base_hi_graph_model.py:115-122
        mesh_emb = mlp(sd, "mesh_embedders.0", g["mesh_static_features"][0], hl)
    else:  # graph_lam.py:66-71
        mesh_emb = mlp(sd, "mesh_embedder", g["mesh_static_features"], hl)
    mesh_rep = interaction_net(
        sd, "g2m_gnn", g["g2m_edge_index"], grid_emb, _expand(mesh_emb, B),
        _expand(g2m_emb, B), update_edges=False, hidden_layers=hl,
    )
    grid_rep = grid_emb + mlp(sd, "encoding_grid_mlp", grid_emb, hl)
    mesh_rep = _PROCESSORS[cfg["model"]](sd, g, mesh_rep, cfg)
    grid_rep = interaction_net(
        sd, "m2g_gnn", g["m2g_edge_index"], mesh_rep, grid_rep,
        _expand(m2g_emb, B), update_edges=False, hidden_layers=hl,
    )
    net_out = mlp(sd, "output_map", grid_rep, hl, layer_norm=False)
    if cfg.get("output_std", False):
        delta, std_raw = net_out.chunk(2, dim=-1)
        pred_std = F.softplus(std_raw)
    else:
        delta, pred_std = net_out, None
    return prev_state + delta * data["diff_std"] + data["diff_mean"], pred_std


def unroll_prediction(sd, g, cfg, data, init_states, forcing, true_states):
    """ar_model.py:220-267: T-step rollout with boundary overwrite (:244-247)."""
    prev_prev, prev = init_states[:, 0], init_states[:, 1]
    preds, stds = [], []
    bm = data["boundary_mask"]  # (N,1)
    im = 1.0 - bm
    for t in range(forcing.shape[1]):
        pred, std = predict_step(sd, g, cfg, data, prev, prev_prev, forcing[:, t])
        new = bm * true_states[:, t] + im * pred
        preds.append(new)
        stds.append(std)
        prev_prev, prev = prev, new
    prediction = torch.stack(preds, dim=1)
    if cfg.get("output_std", False):
        return prediction, torch.stack(stds, dim=1)
    return prediction, data["per_var_std"]


def wmse(pred, target, pred_std, mask=None):
    """metrics.py:56-84 + :21-53 (average_grid=True, sum_vars=True)."""
    v = (pred - target) ** 2 / (pred_std**2)
    if mask is not None:
        v = v[..., mask, :]
    return v.mean(dim=-2).sum(dim=-1)


def mse(pred, target, pred_std, mask=None):
    """metrics.py:87-108: wmse with unit std."""
    return wmse(pred, target, torch.ones_like(pred_std), mask)


def training_loss(sd, g, cfg, data, init_states, target_states, forcing):
    """ar_model.py:287-298 training_step: mean over (B,T) of the masked loss."""
    pred, std = unroll_prediction(sd, g, cfg, data, init_states, forcing, target_states)
    fn = {"wmse": wmse, "mse": mse}[cfg.get("loss", "wmse")]
    mask = (1.0 - data["boundary_mask"])[:, 0].to(torch.bool)
    return torch.mean(fn(pred, target_states, std, mask)), pred
