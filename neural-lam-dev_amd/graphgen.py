"""
Synthetic graph generator in the reference's on-disk ``load_graph`` format.

Produces the same *graph family* as the reference's offline tool
(neural_lam/create_graph.py:157-535) without networkx / torch_geometric:
  * square mesh levels of n = 3^k nodes per side placed at cell centres of the
    grid's bounding box (create_graph.py:116-126), 8-neighbour bidirectional
    edges (:128-149), edge features [len, dx, dy] = pos[sender] - pos[receiver];
  * multiscale: coarser levels are merged onto the finest level's nodes through
    the centre-child map i -> 3 i + 1 (create_graph.py:372-386);
  * hierarchical: levels keep separate node ids (mesh levels first, offset per
    level, grid last), each lower-level node gets one down edge from its
    nearest upper-level node; up edges are the reversed pairs with the SAME
    features (create_graph.py:283-341);
  * g2m: every grid node within 0.67 x mesh spacing of a bottom-level mesh node
    sends to it (create_graph.py:424-477); m2g: every grid node receives from its
    4 nearest bottom-level mesh nodes (create_graph.py:490-519);
  * mesh node features = pos / max|grid xy| (create_graph.py:236,410).
Files: {m2m,g2m,m2g}_{edge_index,features}.pt, mesh_features.pt and, if
hierarchical, mesh_{up,down}_{edge_index,features}.pt  (create_graph.py:84-107).

Edge *order* inside a file is sender-major here; the reference's order is
networkx' adjacency order.  Consumers must treat edge_index as given.
Pinned against the reference tool itself: tests/golden/make_graph_golden.py runs
create_graph.py from /root/reference and tests/test_host_logic.py compares edge sets,
edge features and mesh features of four grids (incl. equidistant-neighbour ties).
"""
import os

import numpy as np
import torch


def make_xy(nx, ny, spacing=10000.0):
    """Regular grid coordinates, shape (nx, ny, 2), as a datastore's
    ``get_xy(stacked=False)`` returns them."""
    x = np.arange(nx, dtype=np.float64) * spacing
    y = np.arange(ny, dtype=np.float64) * spacing
    gx, gy = np.meshgrid(x, y, indexing="ij")
    return np.stack((gx, gy), axis=-1)


def _level_positions(xy, n):
    xm, xM = xy[:, 0, 0].min(), xy[:, 0, 0].max()
    ym, yM = xy[0, :, 1].min(), xy[0, :, 1].max()
    dx, dy = (xM - xm) / n, (yM - ym) / n
    lx = np.linspace(xm + dx / 2, xM - dx / 2, n)
    ly = np.linspace(ym + dy / 2, yM - dy / 2, n)
    gx, gy = np.meshgrid(lx, ly, indexing="ij")
    return np.stack((gx, gy), axis=-1).reshape(n * n, 2)  # id = i*n + j


_NEIGH = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


def _level_edges(n):
    """Directed 8-neighbour edges of an n x n lattice, sender-major."""
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    i, j = i.reshape(-1), j.reshape(-1)
    send, rec = [], []
    for di, dj in _NEIGH:
        ii, jj = i + di, j + dj
        ok = (ii >= 0) & (ii < n) & (jj >= 0) & (jj < n)
        send.append((i * n + j)[ok])
        rec.append((ii * n + jj)[ok])
    send, rec = np.concatenate(send), np.concatenate(rec)
    order = np.lexsort((rec, send))
    return send[order], rec[order]


def _edge_features(pos_send, pos_rec):
    vdiff = pos_send - pos_rec
    length = np.sqrt((vdiff**2).sum(axis=1, keepdims=True))
    return torch.from_numpy(np.concatenate((length, vdiff), axis=1)).to(torch.float32)


def _ei(send, rec):
    return torch.from_numpy(np.stack((send, rec)).astype(np.int64))


def create_graph(graph_dir, xy, n_max_levels=None, hierarchical=False):
    """Write a graph directory for grid coordinates ``xy`` (nx, ny, 2).
    Returns a dict of sizes."""
    # the reference's class (create_graph.py:299,445,493): scipy.spatial.KDTree defaults to
    # leafsize 10, cKDTree to 16, and equidistant 4th neighbours are resolved by tree order
    from scipy.spatial import KDTree as cKDTree

    os.makedirs(graph_dir, exist_ok=True)
    nx_ref = 3
    nlev = int(np.log(max(xy.shape[:2])) / np.log(nx_ref))
    nleaf = nx_ref**nlev
    mesh_levels = nlev - 1
    if n_max_levels:
        mesh_levels = min(mesh_levels, n_max_levels)
    sides = [nleaf // nx_ref**lev for lev in range(1, mesh_levels + 1)]
    level_pos = [_level_positions(xy, n) for n in sides]
    pos_max = np.abs(xy).max()

    info = {"level_sides": sides}
    if hierarchical:
        sizes = [n * n for n in sides]
        first = np.concatenate(([0], np.cumsum(sizes)[:-1])).astype(np.int64)
        m2m_ei, m2m_f = [], []
        for n, pos, off in zip(sides, level_pos, first):
            s, r = _level_edges(n)
            m2m_ei.append(_ei(s + off, r + off))
            m2m_f.append(_edge_features(pos[s], pos[r]))
        up_ei, up_f, down_ei, down_f = [], [], [], []
        for l in range(mesh_levels - 1):
            tree = cKDTree(level_pos[l + 1])
            parent = tree.query(level_pos[l], 1)[1]
            child = np.arange(sizes[l])
            order = np.lexsort((child, parent))
            s, r = parent[order], child[order]
            feat = _edge_features(level_pos[l + 1][s], level_pos[l][r])
            down_ei.append(_ei(s + first[l + 1], r + first[l]))
            up_ei.append(_ei(r + first[l], s + first[l + 1]))
            down_f.append(feat)
            up_f.append(feat.clone())
        torch.save(up_ei, os.path.join(graph_dir, "mesh_up_edge_index.pt"))
        torch.save(down_ei, os.path.join(graph_dir, "mesh_down_edge_index.pt"))
        torch.save(up_f, os.path.join(graph_dir, "mesh_up_features.pt"))
        torch.save(down_f, os.path.join(graph_dir, "mesh_down_features.pt"))
        mesh_pos = [torch.from_numpy(p).to(torch.float32) / torch.tensor(pos_max)
                    for p in level_pos]
        num_mesh_total = int(sum(sizes))
        info["m2m_edges"] = [int(e.shape[1]) for e in m2m_ei]
        info["updown_edges"] = [int(e.shape[1]) for e in up_ei]
    else:
        n0 = sides[0]
        send_all, rec_all, feat_all = [], [], []
        for lev, (n, pos) in enumerate(zip(sides, level_pos)):
            s, r = _level_edges(n)
            feat_all.append(_edge_features(pos[s], pos[r]))

            def to_fine(idx, n=n, lev=lev):
                i, j = idx // n, idx % n
                for _ in range(lev):
                    i, j = nx_ref * i + 1, nx_ref * j + 1
                return i * n0 + j

            send_all.append(to_fine(s))
            rec_all.append(to_fine(r))
            if lev > 0:
                # networkx.compose (create_graph.py:386) lets the coarser level's node
                # attributes overwrite the finer ones: a merged node carries the position of
                # the COARSEST level it belongs to (equal up to rounding, which decides
                # equidistant-neighbour ties and the last bits of g2m / m2g features)
                merged_pos = level_pos[0] if lev == 1 else merged_pos
                merged_pos = merged_pos.copy()
                merged_pos[to_fine(np.arange(n * n))] = pos
        send, rec = np.concatenate(send_all), np.concatenate(rec_all)
        feat = torch.cat(feat_all, dim=0)
        order = np.argsort(send, kind="stable")
        m2m_ei = [_ei(send[order], rec[order])]
        m2m_f = [feat[torch.from_numpy(order)]]
        if len(sides) > 1:
            level_pos[0] = merged_pos
        # float32 positions divided in float32 (create_graph.py:400,410)
        mesh_pos = [torch.from_numpy(level_pos[0]).to(torch.float32) / torch.tensor(pos_max)]
        num_mesh_total = n0 * n0
        info["m2m_edges"] = [int(m2m_ei[0].shape[1])]
    torch.save(m2m_ei, os.path.join(graph_dir, "m2m_edge_index.pt"))
    torch.save(m2m_f, os.path.join(graph_dir, "m2m_features.pt"))
    torch.save(mesh_pos, os.path.join(graph_dir, "mesh_features.pt"))

    # grid nodes: id = num_mesh_total + a*Nx + b with pos xy[b, a]
    Nx, Ny = xy.shape[:2]
    grid_pos = xy.transpose(1, 0, 2).reshape(Ny * Nx, 2)
    bottom = level_pos[0]
    n0 = sides[0]
    # create_graph.py:428-433: distance between bottom-level nodes (0, 1, 0) and (0, 0, 0),
    # i.e. the spacing along the FIRST lattice index (x); node id = i * n0 + j
    dm = np.sqrt(((bottom[n0] - bottom[0]) ** 2).sum())

    gtree = cKDTree(grid_pos)
    neigh = gtree.query_ball_point(bottom, dm * 0.67)
    rec = np.concatenate([np.full(len(nb), m, dtype=np.int64) for m, nb in enumerate(neigh)])
    send = np.concatenate([np.asarray(sorted(nb), dtype=np.int64) for nb in neigh])
    order = np.lexsort((rec, send))
    send, rec = send[order], rec[order]
    torch.save(_ei(send + num_mesh_total, rec), os.path.join(graph_dir, "g2m_edge_index.pt"))
    torch.save(
        _edge_features(grid_pos[send], bottom[rec]), os.path.join(graph_dir, "g2m_features.pt")
    )
    info["g2m_edges"] = int(send.shape[0])

    mtree = cKDTree(bottom)
    nn4 = mtree.query(grid_pos, 4)[1]  # (N_grid, 4)
    rec = np.repeat(np.arange(grid_pos.shape[0], dtype=np.int64), 4)
    send = nn4.reshape(-1).astype(np.int64)
    order = np.lexsort((rec, send))
    send, rec = send[order], rec[order]
    torch.save(_ei(send, rec + num_mesh_total), os.path.join(graph_dir, "m2g_edge_index.pt"))
    torch.save(
        _edge_features(bottom[send], grid_pos[rec]), os.path.join(graph_dir, "m2g_features.pt")
    )
    info["m2g_edges"] = int(send.shape[0])
    info["num_grid"] = int(grid_pos.shape[0])
    info["num_mesh"] = [int(p.shape[0]) for p in mesh_pos]
    return info
