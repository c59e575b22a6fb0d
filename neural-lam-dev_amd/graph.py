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
        # receiver-aligned 32-edge tiles for the fused edge kernels (None if a receiver has more
        # than 32 in-edges: such graphs run the same kernels on virtual receivers, below)
        self.ntiles = 0
        self.virtual = None
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
            self._sender_parts(tiles[: 4 * nt].reshape(nt, 4), out["csr_send"], n_send)
        else:
            # a receiver with more than 32 in-edges does not fit one receiver-aligned tile: its
            # segment is cut into VIRTUAL receivers of <= 32 consecutive CSR positions; the fused
            # edge kernels run on the virtual graph and a (node-sized) second stage folds the
            # virtual rows back (VirtualReceivers) -- at hidden 64, 128 and 256 alike (wide.Tiling takes
            # the virtual tiling).  Only the hidden-64 SplitMLPs path (fused.inet_split_eligible)
            # still falls back to the generic kernels for such graphs, with a RuntimeWarning.
            self.tiles = None
            self.virtual = VirtualReceivers(out, n_rec, M)


    # Tables of nlam_edge_bwd_parts (include/nlam_hip.h): the distinct senders of every tile get
    # the slots 0, 1, ... in ascending sender order; `part_slot` holds the slot of every CSR
    # position (bits 0-7; bits 8-15: the number of distinct senders of its tile), `pcsc_colptr` / `pcsc_rows` list, per sender, the partial rows 16 * tile + slot that
    # carry its sums (ascending tile order: the reduction order is fixed).  Graphs with a tile of
    # more than 16 distinct senders do not get them (has_sender_parts = False).
    PART_SLOTS = 16

    def _sender_parts(self, tiles, csr_send, n_send):
        self.has_sender_parts = False
        nt = tiles.shape[0]
        ne = (tiles[:, 1] - tiles[:, 0]).astype(np.int64)
        if nt == 0 or int(ne.sum()) != self.M:
            return
        tile_of_pos = np.repeat(np.arange(nt, dtype=np.int64), ne)
        pos = (np.repeat(tiles[:, 0].astype(np.int64), ne) + np.arange(self.M, dtype=np.int64)
               - np.repeat(np.cumsum(ne) - ne, ne))               # CSR positions, tile by tile
        key = tile_of_pos * int(n_send) + csr_send[pos].astype(np.int64)
        pairs, inverse = np.unique(key, return_inverse=True)       # sorted by (tile, sender)
        pair_tile = pairs // int(n_send)
        pair_send = pairs - pair_tile * int(n_send)
        first = np.searchsorted(pair_tile, np.arange(nt, dtype=np.int64), side="left")
        pair_slot = np.arange(len(pairs), dtype=np.int64) - first[pair_tile]
        if len(pairs) == 0 or int(pair_slot.max()) >= self.PART_SLOTS:
            return
        # (packed with the tile's number of distinct senders in bits 8-15: the kernel reads it
        # from the first slot instead of reducing over the wave)
        ns_tile = np.bincount(pair_tile, minlength=nt)
        part_slot = np.zeros(self.M, np.int32)
        part_slot[pos] = (pair_slot[inverse] | (ns_tile[tile_of_pos] << 8)).astype(np.int32)
        order = np.lexsort((pair_tile, pair_send))                  # by sender, then tile
        colptr = np.zeros(int(n_send) + 1, np.int64)
        np.cumsum(np.bincount(pair_send, minlength=int(n_send)), out=colptr[1:])
        rows = (self.PART_SLOTS * pair_tile + pair_slot)[order]
        self.n_sender_parts = int(len(pairs))
        for k, v in (("part_slot", part_slot), ("pcsc_colptr", colptr.astype(np.int32)),
                     ("pcsc_rows", rows.astype(np.int32))):
            self.register_buffer(k, torch.from_numpy(np.ascontiguousarray(v)), persistent=False)
        self.has_sender_parts = True


class VirtualReceivers(torch.nn.Module):
    """Receiver-aligned tiles for graphs with in-degree > 32 (reference: any edge_index is legal,
    interaction_net.py:56-62; create_graph.py g2m radii give 13-17 in-edges, a finer grid more).
    Receiver i with deg_i in-edges becomes ceil(deg_i / 32) virtual receivers that own consecutive
    chunks of its CSR segment (receivers without in-edges keep one empty virtual receiver).  The
    fused edge kernels see `rowptr_v` / `csr_rec_v` / `tiles`; `real_of_virt` expands per-receiver
    rows (Pr, g_agg) to virtual rows, `rowptr2` (n_rec + 1, in virtual ids) folds per-virtual
    sums (agg, gPr) back -- a fixed order, so results stay deterministic."""

    def __init__(self, out, n_rec, M):
        super().__init__()
        rowptr = out["csr_rowptr"].astype(np.int64)
        deg = np.diff(rowptr)
        nv_per = np.maximum(1, -(-deg // 32))
        rowptr2 = np.zeros(n_rec + 1, np.int64)
        np.cumsum(nv_per, out=rowptr2[1:])
        n_virt = int(rowptr2[-1])
        real_of_virt = np.repeat(np.arange(n_rec, dtype=np.int64), nv_per)
        k_in_real = np.arange(n_virt, dtype=np.int64) - rowptr2[real_of_virt]
        start = rowptr[real_of_virt] + 32 * k_in_real
        end = np.minimum(start + 32, rowptr[real_of_virt + 1])
        rowptr_v = np.empty(n_virt + 1, np.int32)
        rowptr_v[:-1] = start
        rowptr_v[-1] = M
        assert np.all(end[:-1] == start[1:]) and end[-1] == M
        csr_rec_v = np.repeat(np.arange(n_virt, dtype=np.int32), (end - start).astype(np.int64))
        self.n_rec = n_virt
        self.M = M
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        cap = n_virt + M // 32 + 2
        tiles = np.empty(4 * cap, np.int32)
        nt = lib.nlam_graph_tiles_host(p(rowptr_v), n_virt, 32, 32, p(tiles), cap)
        if nt < 0:
            raise RuntimeError(lib.nlam_last_error().decode())
        self.ntiles = int(nt)
        bufs = {"tiles": tiles[: 4 * nt].copy().reshape(nt, 4), "csr_rowptr": rowptr_v,
                "csr_rec": csr_rec_v, "real_of_virt": real_of_virt.astype(np.int32),
                "rowptr2": rowptr2.astype(np.int32)}
        for k, v in bufs.items():
            self.register_buffer(k, torch.from_numpy(np.ascontiguousarray(v)), persistent=False)
