"""CPU tests of the host side: C-ABI library loads and exports every declared
symbol, graph tables, module/state_dict layout vs the golden state_dicts,
loud failure on CPU tensors."""
import glob
import os
import re
import tempfile

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    from neural_lam_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "nlam_hip.h")).read()
    declared = set(re.findall(r"\b(nlam_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(_lib.lib, name), f"libnlam_hip.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # the header's version macro, the library's and the binding's agree (a stale prebuilt .so
    # selected through NLAM_LIB_PATH is refused at import: _lib._load)
    ver = int(re.search(r"#define\s+NLAM_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert _lib.lib.nlam_abi_version() == ver == _lib.ABI_VERSION


def test_ctypes_struct_mirrors_have_the_library_layout():
    """inet_seq.Args / Grads mirror nlam_inet_args / nlam_inet_grads field by field: same size as
    the C structs the library was compiled with (the import itself refuses a mismatch)."""
    import ctypes

    from neural_lam_amd import _lib, inet_seq

    assert ctypes.sizeof(inet_seq.Args) == _lib.lib.nlam_sizeof_inet_args()
    assert ctypes.sizeof(inet_seq.Grads) == _lib.lib.nlam_sizeof_inet_grads()
    # the two flags added in round 5 are the last fields, after the six output pointers
    names = [f[0] for f in inet_seq.Args._fields_]
    assert names[-2:] == ["ps_given", "pr_given"] and names[-8:-2] == ["P", "Pr", "Pe", "agg", "e_out", "rec_out"]


def test_graph_tables_match_numpy():
    from neural_lam_amd.graph import EdgeTables, normalise_edge_index

    gen = torch.Generator().manual_seed(0)
    M, n_s, n_r = 500, 37, 23
    ei = torch.stack((torch.randint(0, n_s, (M,), generator=gen) + 100,
                      torch.randint(0, n_r, (M,), generator=gen) + 5))
    ei[0, 0], ei[1, 0], ei[1, 1], ei[0, 1] = 100, 5, 5 + n_r - 1, 100 + n_s - 1
    send, rec, num_rec, num_send = normalise_edge_index(ei)
    assert num_rec == n_r and num_send == n_s
    t = EdgeTables(send, rec, num_send, num_rec)
    s, r = send.numpy(), rec.numpy()
    order = np.argsort(r, kind="stable")
    assert np.array_equal(t.csr_eid.numpy(), order)
    assert np.array_equal(t.csr_send.numpy(), s[order])
    assert np.array_equal(t.csr_rec.numpy(), r[order])
    assert np.array_equal(np.diff(t.csr_rowptr.numpy()), np.bincount(r, minlength=n_r))
    # CSC lists hold CSR positions, ascending, grouped by sender
    pos = t.csc_pos.numpy()
    colptr = t.csc_colptr.numpy()
    assert np.array_equal(np.diff(colptr), np.bincount(s, minlength=n_s))
    for j in range(n_s):
        seg = pos[colptr[j]:colptr[j + 1]]
        assert np.all(t.csr_send.numpy()[seg] == j) and np.all(np.diff(seg) > 0)
    assert np.array_equal(t.csc_eid.numpy(), t.csr_eid.numpy()[pos])
    deg = np.maximum(np.bincount(r, minlength=n_r), 1)
    assert np.allclose(t.inv_deg.numpy(), 1.0 / deg)


def test_graph_build_rejects_bad_input():
    from neural_lam_amd._lib import NlamError
    from neural_lam_amd.graph import EdgeTables

    with pytest.raises(NlamError, match="out of"):
        EdgeTables(torch.tensor([0, 5]), torch.tensor([0, 1]), 3, 2)
    with pytest.raises(ValueError):
        from neural_lam_amd.graph import normalise_edge_index

        normalise_edge_index(torch.zeros(2, 0, dtype=torch.int64))


@pytest.mark.parametrize("path", [p for p in sorted(glob.glob(os.path.join(GOLDEN, "model_*.pt")))
                                  if not p.endswith("_bf16.pt")],
                         ids=lambda p: os.path.basename(p)[:-3])
def test_model_state_dict_layout_matches_reference(path):
    """Same parameter names / shapes as the reference's checkpoint, and -- with
    the same seed -- the same default-init values (construction order)."""
    from test_gpu_models import build_model

    fx = torch.load(path, weights_only=False)
    with tempfile.TemporaryDirectory() as tmp:
        torch.manual_seed(42)  # the seed make_golden.py constructs the reference model with
        model = build_model(fx, tmp, load_weights=False)
    sd = model.state_dict()
    assert list(sd.keys()) == list(fx["state_dict"].keys())
    for k, v in sd.items():
        assert v.shape == fx["state_dict"][k].shape, k
        assert torch.equal(v, fx["state_dict"][k]), f"default init differs at {k}"


def test_cpu_tensors_fail_loudly():
    from neural_lam_amd.interaction_net import InteractionNet

    net = InteractionNet(torch.tensor([[3, 4, 5, 3], [0, 1, 2, 2]]), 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.randn(1, 3, 8), torch.randn(1, 3, 8), torch.randn(1, 4, 8))
    mlp = __import__("neural_lam_amd").make_mlp([3, 8, 8])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mlp(torch.randn(5, 3))


def test_eval_metrics_match_reference_and_closed_forms():
    """wmae / mae / nll / crps_gauss (reference metrics.py:111-237): against torch.distributions
    closed forms, and against the reference's own metrics.py when it is present."""
    import math
    from neural_lam_amd import metrics as M

    gen = torch.Generator().manual_seed(0)
    pred = torch.randn(2, 3, 50, 4, generator=gen)
    target = torch.randn(2, 3, 50, 4, generator=gen)
    std = torch.rand(4, generator=gen) + 0.5
    std_full = torch.rand(2, 3, 50, 4, generator=gen) + 0.3
    mask = torch.rand(50, generator=gen) < 0.7
    normal = torch.distributions.Normal(pred, std_full)
    want_nll = (-normal.log_prob(target))[..., mask, :].mean(-2).sum(-1)
    assert torch.allclose(M.nll(pred, target, std_full, mask=mask), want_nll, rtol=1e-5, atol=1e-6)
    z = (target - pred) / std
    sn = torch.distributions.Normal(0.0, 1.0)
    want_crps = (-std * (math.pi ** -0.5 - 2 * torch.exp(sn.log_prob(z)) - z * (2 * sn.cdf(z) - 1)))
    assert torch.allclose(M.crps_gauss(pred, target, std, average_grid=False, sum_vars=False),
                          want_crps, rtol=1e-5, atol=1e-6)
    assert torch.allclose(M.mae(pred, target, std, sum_vars=False), (pred - target).abs().mean(-2))
    assert torch.allclose(M.wmae(pred, target, std), ((pred - target).abs() / std).mean(-2).sum(-1))
    assert set(M.DEFINED_METRICS) == {"mse", "mae", "wmse", "wmae", "nll", "crps_gauss"}
    from oracle import ref_shim
    if ref_shim.available():
        ref = ref_shim.load().metrics
        for name in M.DEFINED_METRICS:
            for kw in (dict(), dict(mask=mask, sum_vars=False), dict(average_grid=False)):
                a = M.get_metric(name)(pred, target, std_full, **kw)
                b = ref.get_metric(name)(pred, target, std_full, **kw)
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (name, kw)


GRAPH_FIXTURES = sorted(__import__("glob").glob(os.path.join(GOLDEN, "graph_*.pt")))


@pytest.mark.parametrize("path", GRAPH_FIXTURES, ids=[os.path.basename(p)[:-3] for p in GRAPH_FIXTURES])
def test_graphgen_matches_reference_create_graph(path):
    """graphgen.py against the REFERENCE's own create_graph.py (fixtures written by
    tests/golden/make_graph_golden.py from /root/reference): identical edge sets, bit-identical
    per-edge features and mesh node features, for flat and hierarchical graphs incl. a
    non-square grid with equidistant-neighbour ties (create_graph.py:157-535)."""
    import tempfile

    from neural_lam_amd import graphgen

    fx = torch.load(path, weights_only=False)
    nx, ny, sp, nml, hier = fx["case"]
    ref = fx["graph"]

    def canon(ei, ft):
        ei = ei.to(torch.int64)
        order = torch.argsort(ei[0] * (int(ei.max()) + 1) + ei[1])
        return ei[:, order].to(torch.int32), ft[order].to(torch.float32)

    with tempfile.TemporaryDirectory() as tmp:
        graphgen.create_graph(tmp, graphgen.make_xy(nx, ny, sp), nml, hier)
        for name, want in ref.items():
            if name == "mesh_features":
                got = torch.load(os.path.join(tmp, "mesh_features.pt"), weights_only=False)
                assert len(got) == len(want)
                for a, b in zip(got, want):
                    assert torch.equal(a, b)
                continue
            ei = torch.load(os.path.join(tmp, f"{name}_edge_index.pt"), weights_only=False)
            ft = torch.load(os.path.join(tmp, f"{name}_features.pt"), weights_only=False)
            if isinstance(want["edge_index"], list):
                assert len(ei) == len(want["edge_index"])
                pairs = zip(ei, ft, want["edge_index"], want["features"])
            else:
                pairs = [(ei, ft, want["edge_index"], want["features"])]
            for a, fa, b, fb in pairs:
                ca, cfa = canon(a, fa)
                assert torch.equal(ca, b), name
                assert torch.equal(cfa, fb), name
        if not hier:
            assert not os.path.exists(os.path.join(tmp, "mesh_up_edge_index.pt"))


def test_multi_problem_launch_shares_follow_the_work():
    """Workgroups per problem of a multi-problem launch (hidden 128 / 256 projections): one round of
    the device in all, proportional to the problems' trip rounds, at least one and never more than
    a problem has rounds.  (Equal shares cost Hi-LAM-256 1.5 ms per step: DESIGN.md 4.10.)"""
    import ctypes

    from neural_lam_amd._lib import lib

    def shares(rounds, cap=256):
        n = len(rounds)
        out = (ctypes.c_int64 * n)()
        assert lib.nlam_debug_multi_shares(n, (ctypes.c_int64 * n)(*rounds), cap, out) == 0
        return list(out)

    # level-0 same-level net of Hi-LAM at hidden 256: edge rows 8 x the node rows (64-row tiles)
    g = shares([3220, 410, 410])
    assert sum(g) == 256 and g[1] == g[2] and 7.0 < g[0] / g[1] < 8.6, g
    # everything fits: a workgroup per round
    assert shares([10, 3, 1]) == [10, 3, 1]
    # a tiny problem beside a huge one still gets a workgroup; the sum stays one round
    g = shares([100000, 1])
    assert g[1] == 1 and sum(g) == 256, g
    # never more workgroups than rounds, even for the problem that takes the remainder
    g = shares([300, 2, 2])
    assert g[0] <= 300 and g[1] <= 2 and sum(g) <= 256, g
    # the at-least-one bump of small problems must not push the total past the cap
    for rounds in ([1000, 1, 1], [19000, 30, 40], [5000, 1, 1, 1, 1, 1, 1, 1]):
        g = shares(rounds)
        assert sum(g) <= 256 and min(g) >= 1, (rounds, g)
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(1, 9))
        rounds = [int(x) for x in rng.integers(1, 20000, n)]
        g = shares(rounds)
        assert all(1 <= gi <= ri for gi, ri in zip(g, rounds)), (rounds, g)
        assert sum(g) <= max(256, n), (rounds, g)
        if sum(rounds) > 256:
            assert sum(g) >= 256 - n, (rounds, g)   # (the device is filled)
            big = max(range(n), key=lambda k: rounds[k])
            for k in range(n):   # proportional within rounding
                assert abs(g[k] - 256 * rounds[k] / sum(rounds)) <= (1 if k != big else n + 1), (rounds, g)


def test_sender_partial_tables_against_a_brute_force_walk():
    """EdgeTables._sender_parts (the tables of nlam_edge_bwd_parts): every CSR position's slot is
    the rank of its sender among the distinct senders of its tile; every sender's list holds
    exactly the rows 16 * tile + slot of the tiles it appears in, ascending; a graph with a tile
    of more than 16 distinct senders gets no tables."""
    import torch

    from neural_lam_amd.graph import EdgeTables

    gen = torch.Generator().manual_seed(0)
    n_r, n_s, M = 500, 60, 2000
    rec = torch.arange(M) % n_r
    send = (rec // 10 + torch.randint(0, 3, (M,), generator=gen)) % n_s
    t = EdgeTables(send, rec, n_s, n_r)
    assert t.has_sender_parts and 0 < t.n_sender_parts < M
    tiles, cs, ps = t.tiles.numpy(), t.csr_send.numpy(), t.part_slot.numpy()
    for p0, p1, _, _ in tiles:
        u = sorted(set(cs[p0:p1].tolist()))
        assert len(u) <= EdgeTables.PART_SLOTS
        assert [int(ps[p]) & 255 for p in range(p0, p1)] == [u.index(cs[p]) for p in range(p0, p1)]
        assert all(int(ps[p]) >> 8 == len(u) for p in range(p0, p1))
    colptr, rows = t.pcsc_colptr.numpy(), t.pcsc_rows.numpy()
    assert colptr[0] == 0 and colptr[-1] == t.n_sender_parts == len(rows)
    for s in range(n_s):
        want = sorted({16 * ti + (int(ps[p]) & 255) for ti, (p0, p1, _, _) in enumerate(tiles)
                       for p in range(p0, p1) if cs[p] == s})
        assert rows[colptr[s]:colptr[s + 1]].tolist() == want
    # random senders: 32-edge tiles with ~30 distinct senders -> no tables
    t2 = EdgeTables(torch.randint(0, 1000, (M,), generator=gen), rec, 1000, n_r)
    assert not t2.has_sender_parts and not hasattr(t2, "part_slot")
