"""Edge tables of one InteractionNet: the reference's index normalisation
(interaction_net.py:56-61) plus the receiver-sorted (CSR) / sender-sorted (CSC)
orderings the HIP kernels walk.  Built once on the host by
nlam_graph_build_host (csrc/graph_host.cpp) and kept as int32 device buffers.
"""
import ctypes

import numpy as np
import torch

from ._lib import check, lib


def normalise_edge_index(edge_index):
    """interaction_net.py:56-61: both rows re-based to 0; num_rec = max+1.
    Returns (send_local (M,), rec (M,), num_rec, num_send_span) as int64 CPU."""
    ei = edge_index.detach().to("cpu", torch.int64)
    if ei.dim() != 2 or ei.shape[0] != 2 or ei.shape[1] == 0:
        raise ValueError(f"edge_index must be (2, M) with M > 0, got {tuple(ei.shape)}")
    ei = ei - ei.min(dim=1, keepdim=True)[0]
    return ei[0].contiguous(), ei[1].contiguous(), int(ei[1].max()) + 1, int(ei[0].max()) + 1


class EdgeTables(torch.nn.Module):
    """int32 device tables (non-persistent buffers, like the reference's
    edge_index buffer, interaction_net.py:62)."""

    NAMES = ("send", "rec", "csr_rowptr", "csr_eid", "csr_send", "csr_rec", "csc_colptr",
             "csc_pos", "csc_eid", "inv_deg", "pos_of_eid")

    def __init__(self, send, rec, n_send, n_rec):
        super().__init__()
        M = int(send.shape[0])
        self.M, self.n_send, self.n_rec = M, int(n_send), int(n_rec)
        self.tag = "inet"  # profiler label; models set g2m / m2m / m2g / ...
        s = np.ascontiguousarray(send.numpy(), dtype=np.int64)
        r = np.ascontiguousarray(rec.numpy(), dtype=np.int64)
        out = {
            "csr_rowptr": np.empty(n_rec + 1, np.int32), "csr_eid": np.empty(M, np.int32),
            "csr_send": np.empty(M, np.int32), "csr_rec": np.empty(M, np.int32),
            "csc_colptr": np.empty(n_send + 1, np.int32), "csc_pos": np.empty(M, np.int32),
            "csc_eid": np.empty(M, np.int32), "inv_deg": np.empty(n_rec, np.float32),
        }
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        check(
            lib.nlam_graph_build_host(
                p(s), p(r), M, n_send, n_rec, p(out["csr_rowptr"]), p(out["csr_eid"]),
                p(out["csr_send"]), p(out["csr_rec"]), p(out["csc_colptr"]), p(out["csc_pos"]),
                p(out["csc_eid"]), p(out["inv_deg"]),
            ),
            "nlam_graph_build_host",
        )
        inv = np.empty(M, np.int32)
        inv[out["csr_eid"]] = np.arange(M, dtype=np.int32)
        out["pos_of_eid"] = inv
        out["send"] = s.astype(np.int32)
        out["rec"] = r.astype(np.int32)
        for k in self.NAMES:
            self.register_buffer(k, torch.from_numpy(out[k]), persistent=False)
        deg = np.diff(out["csr_rowptr"])
        self.max_in_degree = int(deg.max())
        # receiver-aligned 32-edge tiles for the fused edge kernels (None if a
        # receiver has more than 32 in-edges: such graphs take the generic path)
        self.ntiles = 0
        if self.max_in_degree <= 32:
            cap = n_rec + M // 32 + 2
            tiles = np.empty(4 * cap, np.int32)
            nt = lib.nlam_graph_tiles_host(p(out["csr_rowptr"]), n_rec, 32, 32, p(tiles), cap)
            if nt < 0:
                raise RuntimeError(lib.nlam_last_error().decode())
            self.ntiles = int(nt)
            self.register_buffer(
                "tiles", torch.from_numpy(tiles[: 4 * nt].copy()).view(nt, 4), persistent=False
            )
        else:
            import warnings

            warnings.warn(
                f"neural_lam_amd: a receiver has {self.max_in_degree} in-edges (> 32): this "
                "InteractionNet runs on the generic HIP kernel sequence (several times slower) "
                "instead of the fused receiver-aligned tiles", RuntimeWarning, stacklevel=3)
            self.tiles = None
