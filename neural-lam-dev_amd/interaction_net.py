"""Drop-in for neural_lam/interaction_net.py (InteractionNet :10-131,
SplitMLPs :134-163): same constructor, forward signature, parameter names and
assertions; the arithmetic runs in libnlam_hip.so.
"""
import torch
from torch import nn

from . import fused, generic, utils, wide
from .graph import EdgeTables, normalise_edge_index


class SplitMLPs(nn.Module):
    """interaction_net.py:134-163: one MLP per contiguous row chunk (dim -2)."""

    def __init__(self, mlps, chunk_sizes):
        super().__init__()
        assert len(mlps) == len(chunk_sizes), "Number of MLPs must match the number of chunks"
        self.mlps = nn.ModuleList(mlps)
        self.chunk_sizes = chunk_sizes

    def forward(self, x):
        chunks = torch.split(x, self.chunk_sizes, dim=-2)
        outs = [mlp(c.contiguous()) for mlp, c in zip(self.mlps, chunks)]
        return torch.cat(outs, dim=-2)


def _blocks(mod, total_rows):
    """[(row0, row1, n_layers, has_ln, param_offset)], flat param list."""
    params, blocks = [], []
    if isinstance(mod, SplitMLPs):
        r0 = 0
        for mlp, n_rows in zip(mod.mlps, mod.chunk_sizes):
            n, has_ln, flat = generic.mlp_params(mlp)
            blocks.append((r0, r0 + n_rows, n, has_ln, len(params)))
            params += flat
            r0 += n_rows
        assert r0 == total_rows, "chunk sizes do not add up to the number of rows"
    else:
        n, has_ln, flat = generic.mlp_params(mod)
        blocks.append((0, total_rows, n, has_ln, 0))
        params += flat
    return blocks, params


class InteractionNet(nn.Module):
    """Interaction Network layer (Battaglia et al. 2016) as used by neural-lam.

    m_k   = edge_mlp([e_k, x_send(k), x_rec(k)])          (interaction_net.py:117-121)
    agg_i = sum / mean over in-edges of m_k                 (:124-131)
    x_i'  = x_i + aggr_mlp([x_i, agg_i])                    (:106-109)
    e_k'  = e_k + m_k  if update_edges                      (:111-113)
    """

    def __init__(self, edge_index, input_dim, update_edges=True, hidden_layers=1, hidden_dim=None,
                 edge_chunk_sizes=None, aggr_chunk_sizes=None, aggr="sum"):
        assert aggr in ("sum", "mean"), f"Unknown aggregation method: {aggr}"
        super().__init__()
        if hidden_dim is None:
            hidden_dim = input_dim
        send, rec, num_rec, num_send = normalise_edge_index(edge_index)
        self.num_rec = num_rec
        self.aggr = aggr
        # reference layout of the (non-persistent) edge_index buffer: senders
        # offset behind the receivers (interaction_net.py:59-62)
        self.register_buffer("edge_index", torch.stack((send + num_rec, rec)), persistent=False)
        self.tables = EdgeTables(send, rec, num_send, num_rec)
        if self.tables.virtual is not None and (hidden_dim or input_dim) not in (64, 128, 256):
            import warnings

            warnings.warn(
                f"neural_lam_amd: a receiver has {self.tables.max_in_degree} in-edges (> 32): at "
                "hidden widths other than 64 / 128 / 256 this InteractionNet runs on the generic HIP "
                "kernel sequence (several times slower) instead of the fused receiver-aligned tiles",
                RuntimeWarning, stacklevel=2)

        edge_mlp_recipe = [3 * input_dim] + [hidden_dim] * (hidden_layers + 1)
        aggr_mlp_recipe = [2 * input_dim] + [hidden_dim] * (hidden_layers + 1)
        if edge_chunk_sizes is None:
            self.edge_mlp = utils.make_mlp(edge_mlp_recipe)
        else:
            self.edge_mlp = SplitMLPs(
                [utils.make_mlp(edge_mlp_recipe) for _ in edge_chunk_sizes], edge_chunk_sizes
            )
        if aggr_chunk_sizes is None:
            self.aggr_mlp = utils.make_mlp(aggr_mlp_recipe)
        else:
            self.aggr_mlp = SplitMLPs(
                [utils.make_mlp(aggr_mlp_recipe) for _ in aggr_chunk_sizes], aggr_chunk_sizes
            )
        self.update_edges = update_edges
        self.input_dim, self.hidden_dim, self.hidden_layers = input_dim, hidden_dim, hidden_layers
        # SplitMLPs over the edges (HiLAMParallel): one table set per edge chunk, so that the
        # fused kernels run chunk by chunk with that chunk's weights (fused.apply_inet_split)
        self.chunk_tables = None
        if edge_chunk_sizes is not None:
            assert sum(edge_chunk_sizes) == send.shape[0], "edge chunk sizes do not add up"
            tabs, o = [], 0
            for m in edge_chunk_sizes:
                tabs.append(EdgeTables(send[o : o + m], rec[o : o + m], num_send, num_rec))
                o += m
            self.chunk_tables = nn.ModuleList(tabs)
            if hidden_dim == 64 and any(t.virtual is not None for t in tabs):
                import warnings

                warnings.warn(
                    "neural_lam_amd: an edge chunk of this SplitMLPs InteractionNet has a receiver "
                    "with more than 32 in-edges: at hidden 64 the SplitMLPs path then runs on the "
                    "generic HIP kernel sequence (several times slower) instead of the fused tiles",
                    RuntimeWarning, stacklevel=2)

    def forward(self, send_rep, rec_rep, edge_rep):
        if rec_rep.shape[-2] != self.num_rec:
            raise RuntimeError(
                f"rec_rep has {rec_rep.shape[-2]} rows but edge_index addresses {self.num_rec} "
                "receivers (interaction_net.py:58,106)"
            )
        if send_rep.shape[-2] < self.tables.n_send or edge_rep.shape[-2] != self.tables.M:
            raise RuntimeError("send_rep / edge_rep row counts do not match edge_index")
        # any leading dims (node dim = -2, interaction_net.py:86-115): every path below works
        # on (B, rows, d); expand_to_batch views pass through untouched (3-D already)
        lead = max((send_rep.shape[:-2], rec_rep.shape[:-2], edge_rep.shape[:-2]), key=len)

        def as3(t):
            return t if t.dim() == 3 else t.reshape(-1, t.shape[-2], t.shape[-1])

        same = send_rep is rec_rep
        s3, e3 = as3(send_rep), as3(edge_rep)
        r3 = s3 if same else as3(rec_rep)

        def restore(out):
            if len(lead) == 1:
                return out
            if self.update_edges:
                return (out[0].reshape(*lead, *out[0].shape[-2:]),
                        out[1].reshape(*lead, *out[1].shape[-2:]))
            return out.reshape(*lead, *out.shape[-2:])

        if fused.inet_eligible(self, s3, r3, e3):
            return restore(fused.apply_inet(self, s3, r3, e3))
        if wide.inet_eligible(self, s3, r3, e3):
            return restore(wide.apply_inet(self, s3, r3, e3))
        if fused.inet_split_eligible(self, s3, r3, e3):
            return restore(fused.apply_inet_split(self, s3, r3, e3))
        if wide.inet_split_eligible(self, s3, r3, e3):
            return restore(wide.apply_inet_split(self, s3, r3, e3))
        eb, ep = _blocks(self.edge_mlp, self.tables.M)
        ab, ap = _blocks(self.aggr_mlp, self.num_rec)
        ab = [(r0, r1, n, ln, off + len(ep)) for (r0, r1, n, ln, off) in ab]
        out = generic.InteractionNetGenericFunction.apply(
            s3, r3, e3, self.tables, self.update_edges,
            self.aggr == "mean", eb, ab, *(ep + ap),
        )
        return restore(out)
