"""GPU tests of the fused gfx950 kernels (C ABI) against plain torch fp32."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.detach().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize(
    "hid,ka,kb,n_out,ln,res,B,rows",
    [
        (64, 64, 0, 64, True, False, 1, 100),
        (64, 64, 64, 64, True, True, 3, 77),     # node update [x_r | agg] + residual
        (64, 3, 0, 64, True, False, 1, 1000),    # edge-feature embedder
        (64, 2, 0, 64, True, False, 1, 33),
        (64, 56, 0, 64, True, False, 2, 250),    # grid embedder
        (64, 64, 0, 17, False, False, 2, 129),   # output map
        (64, 64, 0, 64, True, True, 2, 64),      # encoding_grid_mlp with residual
        (128, 3, 0, 128, True, False, 1, 45),
        (128, 128, 0, 17, False, False, 1, 40),
    ],
)
def test_fused_mlp_fwd(hid, ka, kb, n_out, ln, res, B, rows):
    from neural_lam_amd import ops

    gen = torch.Generator().manual_seed(hid + ka + kb + n_out + rows)
    k_in = ka + kb
    xa = torch.randn(B, rows, ka, generator=gen)
    xb = torch.randn(B, rows, kb, generator=gen) if kb else None
    W1 = torch.randn(hid, k_in, generator=gen) / k_in ** 0.5
    b1 = torch.randn(hid, generator=gen)
    W2 = torch.randn(n_out, hid, generator=gen) / hid ** 0.5
    b2 = torch.randn(n_out, generator=gen)
    gam = 1 + 0.1 * torch.randn(n_out, generator=gen)
    bet = 0.1 * torch.randn(n_out, generator=gen)
    r = torch.randn(B, rows, n_out, generator=gen) if res else None
    x = torch.cat((xa, xb), -1) if kb else xa
    want = F.linear(F.silu(F.linear(x, W1, b1)), W2, b2)
    if ln:
        want = F.layer_norm(want, (n_out,), gam, bet, 1e-5)
    if res:
        want = want + r
    dev = "cuda"
    out = torch.full((B, rows, n_out), float("nan"), device=dev)
    ops.fused_mlp_fwd(
        ops.mat(xa.to(dev)), ops.mat(xb.to(dev)) if kb else None, W1.to(dev), b1.to(dev),
        W2.to(dev), b2.to(dev), gam.to(dev) if ln else None, bet.to(dev) if ln else None,
        ops.mat(r.to(dev)) if res else None, ops.mat(out), hid, n_out,
    )
    assert rel(out, want) < 1e-5


def test_fused_mlp_fwd_strided_views_and_broadcast():
    """sources that are column slices of wider buffers, a batch-invariant
    (stride-0) first source, and a large row count (persistent loop)."""
    from neural_lam_amd import ops

    gen = torch.Generator().manual_seed(9)
    B, rows, d = 3, 20000, 64
    wide = torch.randn(rows, 3 * d, generator=gen)          # batch-invariant
    agg = torch.randn(B, rows, d, generator=gen)
    Wfull = torch.randn(d, 2 * d, generator=gen) / 11.0
    b1, b2 = torch.randn(d, generator=gen), torch.randn(d, generator=gen)
    W2 = torch.randn(d, d, generator=gen) / 8.0
    gam, bet = torch.rand(d, generator=gen) + 0.5, torch.randn(d, generator=gen)
    xr = wide[:, d : 2 * d]
    x = torch.cat((xr.unsqueeze(0).expand(B, -1, -1), agg), -1)
    want = xr + F.layer_norm(F.linear(F.silu(F.linear(x, Wfull, b1)), W2, b2), (d,), gam, bet, 1e-5)
    wd = wide.cuda()
    out = torch.empty(B, rows, d, device="cuda")
    xm = ops.mat(wd.unsqueeze(0).expand(B, -1, -1), d, d)
    ops.fused_mlp_fwd(xm, ops.mat(agg.cuda()), Wfull.cuda(), b1.cuda(), W2.cuda(), b2.cuda(),
                      gam.cuda(), bet.cuda(), xm, ops.mat(out), d, d)
    assert rel(out, want) < 1e-5


@pytest.mark.parametrize("k_in,nA,nB,B,rows", [(64, 64, 64, 2, 333), (64, 64, 0, 1, 50),
                                               (128, 128, 0, 2, 100), (3, 64, 0, 1, 70)])
def test_fused_lin_fwd(k_in, nA, nB, B, rows):
    from neural_lam_amd import ops

    gen = torch.Generator().manual_seed(k_in + nA + nB)
    x = torch.randn(B, rows, k_in, generator=gen)
    Wfull = torch.randn(nA, 3 * k_in, generator=gen) / k_in ** 0.5   # column slices as weights
    WA, WB = Wfull[:, k_in : 2 * k_in], Wfull[:, 2 * k_in :]
    bB = torch.randn(max(nB, 1), generator=gen)
    want = [x @ WA.T]
    if nB:
        want.append(x @ WB.T + bB)
    want = torch.cat(want, -1)
    Wd = Wfull.cuda()
    out = torch.empty(B, rows, nA + nB, device="cuda")
    ops.fused_lin_fwd(ops.mat(x.cuda()), Wd[:, k_in : 2 * k_in], None,
                      Wd[:, 2 * k_in :] if nB else None, bB.cuda() if nB else None, ops.mat(out))
    assert rel(out, want) < 1e-5


def _edge_ref(e_term, ps, pr, send, rec, W2, b2, gam, bet, n_rec, mean):
    h = e_term + ps[:, send] + pr[:, rec]
    m = F.layer_norm(F.linear(F.silu(h), W2, b2), (h.shape[-1],), gam, bet, 1e-5)
    agg = torch.zeros(h.shape[0], n_rec, h.shape[-1]).index_add_(1, rec, m)
    if mean:
        deg = torch.zeros(n_rec).index_add_(0, rec, torch.ones(rec.shape[0])).clamp(min=1)
        agg = agg / deg.view(1, -1, 1)
    return m, agg


@pytest.mark.parametrize("k_in,nA,nB,B,rows,add", [
    (64, 64, 64, 2, 333, True), (64, 64, 0, 3, 50, True), (64, 64, 0, 1, 4100, False),
    (3, 64, 0, 1, 70, False)])
def test_fused_lin_bwd(k_in, nA, nB, B, rows, add):
    """gx = gy [WA; WB] (+ gx_add), dW = gy^T x, db = colsum(gy) -- interaction_net.py:109
    first Linear of edge_mlp, split by operand."""
    from neural_lam_amd import ops

    gen = torch.Generator().manual_seed(k_in + nA + nB + rows)
    x = torch.randn(B, rows, k_in, generator=gen)
    gy = torch.randn(B, rows, nA + nB, generator=gen)
    W = torch.randn(nA + nB, k_in, generator=gen) / k_in ** 0.5
    ga = torch.randn(B, rows, k_in, generator=gen) if add else None
    want_gx = gy @ W + (ga if add else 0)
    want_dW = torch.einsum("brn,brk->nk", gy, x)
    want_db = gy.sum((0, 1))
    dev = "cuda"
    Wd = W.to(dev)
    gx = torch.full((B, rows, k_in), float("nan"), device=dev)
    dW = torch.full((nA + nB, k_in), float("nan"), device=dev)
    db = torch.full((nA + nB,), float("nan"), device=dev)
    ops.fused_lin_bwd(ops.mat(x.to(dev)), ops.mat(gy.to(dev)), Wd[:nA], Wd[nA:] if nB else None,
                      ops.mat(gx), dW[:nA], db[:nA], dW[nA:] if nB else None,
                      db[nA:] if nB else None, gx_add=ops.mat(ga.to(dev)) if add else None)
    assert rel(gx, want_gx) < 2e-5
    assert rel(dW, want_dW) < 2e-5
    assert rel(db, want_db) < 2e-5


@pytest.mark.parametrize("d,egemm,mean,B,n_s,n_r,M", [
    (64, True, False, 2, 50, 50, 400), (64, False, True, 3, 80, 30, 333),
    (64, True, True, 1, 20, 700, 900),
    (128, False, False, 2, 30, 60, 250),
])
def test_fused_edge_fwd(d, egemm, mean, B, n_s, n_r, M):
    from neural_lam_amd import ops
    from neural_lam_amd.graph import EdgeTables

    gen = torch.Generator().manual_seed(d + M)
    send = torch.randint(0, n_s, (M,), generator=gen)
    rec = torch.randint(0, n_r, (M,), generator=gen)
    rec[rec == 3] = 4                      # an empty receiver
    g = EdgeTables(send, rec, n_s, n_r).cuda()
    assert g.ntiles > 0
    e = torch.randn(B if egemm else 1, M, d, generator=gen)
    ps = torch.randn(B, n_s, d, generator=gen)
    pr = torch.randn(1 if not egemm else B, n_r, d, generator=gen)   # batch-invariant in (b)
    W1e = torch.randn(d, d, generator=gen) / d ** 0.5
    W2 = torch.randn(d, d, generator=gen) / d ** 0.5
    b2, gam, bet = (torch.randn(d, generator=gen) for _ in range(3))
    e_term = e @ W1e.T if egemm else e
    m, agg = _edge_ref(e_term, ps, pr.expand(B, -1, -1), send, rec, W2, b2, gam, bet, n_r, mean)
    agg_d = torch.full((B, n_r, d), float("nan"), device="cuda")
    eo_d = torch.full((B, M, d), float("nan"), device="cuda") if egemm else None
    ed, psd, prd = e.cuda(), ps.cuda(), pr.cuda()
    ops.fused_edge_fwd(
        g, ops.mat(ed if egemm else ed.expand(B, -1, -1)), egemm, ops.mat(psd),
        ops.mat(prd.expand(B, -1, -1)), W1e.cuda() if egemm else None, W2.cuda(), b2.cuda(),
        gam.cuda(), bet.cuda(), ops.mat(agg_d), ops.mat(eo_d) if egemm else None, mean, d)
    assert rel(agg_d, agg) < 2e-5
    if egemm:
        assert rel(eo_d, e + m) < 2e-5


@pytest.mark.parametrize(
    "ka,kb,n_out,ln,res_a,B,rows,need_gx",
    [
        (64, 0, 64, True, True, 2, 300, True),      # encoding_grid_mlp (+residual from x)
        (64, 64, 64, True, True, 3, 77, True),      # node update [x_r | agg]
        (3, 0, 64, True, False, 1, 1000, False),    # feature embedder (no input grad)
        (56, 0, 64, True, False, 2, 250, False),    # grid embedder
        (64, 0, 17, False, False, 2, 129, True),    # output map
    ],
)
def test_fused_mlp_bwd(ka, kb, n_out, ln, res_a, B, rows, need_gx):
    from neural_lam_amd import ops

    hid = 64
    gen = torch.Generator().manual_seed(ka + kb + n_out + rows)
    k_in = ka + kb
    xa = torch.randn(B, rows, ka, generator=gen, requires_grad=True)
    xb = torch.randn(B, rows, kb, generator=gen, requires_grad=True) if kb else None
    W1 = (torch.randn(hid, k_in, generator=gen) / k_in ** 0.5).requires_grad_(True)
    b1 = torch.randn(hid, generator=gen, requires_grad=True)
    W2 = (torch.randn(n_out, hid, generator=gen) / hid ** 0.5).requires_grad_(True)
    b2 = torch.randn(n_out, generator=gen, requires_grad=True)
    gam = (1 + 0.1 * torch.randn(n_out, generator=gen)).requires_grad_(True)
    bet = (0.1 * torch.randn(n_out, generator=gen)).requires_grad_(True)
    gy = torch.randn(B, rows, n_out, generator=gen)
    x = torch.cat((xa, xb), -1) if kb else xa
    y = F.linear(F.silu(F.linear(x, W1, b1)), W2, b2)
    if ln:
        y = F.layer_norm(y, (n_out,), gam, bet, 1e-5)
    if res_a:
        y = y + xa[..., :n_out]
    y.backward(gy)
    dev = "cuda"
    gxa = torch.full((B, rows, ka), float("nan"), device=dev) if need_gx else None
    gxb = torch.full((B, rows, kb), float("nan"), device=dev) if (need_gx and kb) else None
    dst = {"dW1": torch.empty(hid, k_in, device=dev), "db1": torch.empty(hid, device=dev),
           "dW2": torch.empty(n_out, hid, device=dev), "db2": torch.empty(n_out, device=dev),
           "dgamma": torch.empty(n_out, device=dev), "dbeta": torch.empty(n_out, device=dev)}
    ops.fused_mlp_bwd(
        ops.mat(xa.detach().to(dev)), ops.mat(xb.detach().to(dev)) if kb else None,
        W1.detach().to(dev), b1.detach().to(dev), W2.detach().to(dev), b2.detach().to(dev),
        gam.detach().to(dev) if ln else None, ops.mat(gy.to(dev)),
        ops.mat(gxa) if need_gx else None, ops.mat(gxb) if gxb is not None else None,
        res_a, hid, n_out, dst)
    dW1, db1, dW2, db2, dg, dbt = (dst[k] for k in ("dW1", "db1", "dW2", "db2", "dgamma", "dbeta"))
    tol = 2e-5 * max(1.0, (B * rows) ** 0.5 / 8)
    assert rel(dW1, W1.grad) < tol and rel(db1, b1.grad) < tol
    assert rel(dW2, W2.grad) < tol and rel(db2, b2.grad) < tol
    if ln:
        assert rel(dg, gam.grad) < tol and rel(dbt, bet.grad) < tol
    if need_gx:
        assert rel(gxa, xa.grad) < 2e-5
        if kb:
            assert rel(gxb, xb.grad) < 2e-5


def test_fused_lin_bwd_batch_invariant_x_sums_gy_on_load():
    """x (1, rows, k) with per-sample gy (B, rows, n): gx = (sum_b gy_b) W, dW = (sum_b gy_b)^T x
    -- the backward of expand_to_batch (ar_model.py:204-209) folded into the load."""
    from neural_lam_amd import ops

    for B, rows in ((4, 1000), (5, 77), (2, 4100)):
        gen = torch.Generator().manual_seed(B * rows)
        x = torch.randn(1, rows, 64, generator=gen)
        gy = torch.randn(B, rows, 64, generator=gen)
        W = torch.randn(64, 64, generator=gen) / 8
        gsum = gy.sum(0, keepdim=True)
        dev = "cuda"
        gx = torch.full((1, rows, 64), float("nan"), device=dev)
        dW = torch.full((64, 64), float("nan"), device=dev)
        db = torch.full((64,), float("nan"), device=dev)
        xm, gm = ops.mat(x.to(dev)), ops.mat(gy.to(dev))
        assert ops.lin_bwd_can_sum(xm, gm)
        ops.fused_lin_bwd(xm, gm, W.to(dev), None, ops.mat(gx), dW, db, None, None,
                          sum_gy_batch=True)
        assert rel(gx, gsum @ W) < 2e-5
        assert rel(dW, torch.einsum("brn,brk->nk", gsum, x)) < 2e-5
        assert rel(db, gsum.sum((0, 1))) < 2e-5


def test_fused_paths_are_taken_for_baseline_shapes():
    """d=64, hidden_layers=1: the fused kernels must be the ones that run."""
    from neural_lam_amd import fused, ops, utils
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(0)
    ei = torch.stack((torch.randint(0, 30, (200,), generator=gen),
                      torch.randint(0, 30, (200,), generator=gen)))
    ei[0, 0], ei[1, 0], ei[1, 1] = 0, 0, 29
    net = InteractionNet(ei, 64).cuda()
    x = torch.randn(2, 30, 64, device="cuda", requires_grad=True)
    e = torch.randn(2, 200, 64, device="cuda", requires_grad=True)
    assert fused.inet_eligible(net, x, x, e)
    ops.PROFILER = ops.KernelProfiler()
    try:
        o_x, o_e = net(x, x, e)
        (o_x.sum() + o_e.sum()).backward()
        mlp = utils.make_mlp([3, 64, 64]).cuda()
        mlp(torch.randn(50, 3, device="cuda")).sum().backward()
        stats = ops.PROFILER.collect()
    finally:
        ops.PROFILER = None
    names = {k.split("@")[0] for k in stats}
    assert {"nlam_edge_fwd", "nlam_edge_bwd", "nlam_mlp_fwd", "nlam_mlp_bwd", "nlam_lin_fwd"} <= names
    # projections backward: nlam_lin_bwd, or (shared nodes, split-bf16 mode) the node-side pair
    assert "nlam_lin_bwd" in names or {"nlam_node_bwd", "nlam_node_outer"} <= names
    assert "nlam_gemm" not in names


def test_fused_path_is_taken_for_split_mlps():
    """SplitMLPs at d=64 (HiLAMParallel's operator, interaction_net.py:134-163): the fused
    kernels run chunk by chunk; no generic GEMM may appear."""
    from neural_lam_amd import fused, ops
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(1)
    ei = torch.stack((torch.randint(0, 30, (200,), generator=gen),
                      torch.randint(0, 30, (200,), generator=gen)))
    ei[0, 0], ei[1, 0], ei[1, 1] = 0, 0, 29
    net = InteractionNet(ei, 64, edge_chunk_sizes=[90, 60, 50], aggr_chunk_sizes=[10, 20]).cuda()
    x = torch.randn(2, 30, 64, device="cuda", requires_grad=True)
    e = torch.randn(2, 200, 64, device="cuda", requires_grad=True)
    assert not fused.inet_eligible(net, x, x, e) and fused.inet_split_eligible(net, x, x, e)
    ops.PROFILER = ops.KernelProfiler()
    try:
        o_x, o_e = net(x, x, e)
        (o_x.sum() + o_e.sum()).backward()
        stats = ops.PROFILER.collect()
    finally:
        ops.PROFILER = None
    names = {k.split("@")[0] for k in stats}
    assert {"nlam_edge_fwd", "nlam_edge_bwd", "nlam_mlp_fwd", "nlam_mlp_bwd"} <= names
    assert "nlam_gemm" not in names
    assert stats["nlam_edge_fwd@inet"]["calls"] == 3 and stats["nlam_mlp_fwd@inet"]["calls"] == 2


def test_sender_partials_match_the_gh_rows_form(monkeypatch):
    """nlam_edge_bwd_parts (per-tile sender partial sums of gh instead of the gh rows, gathered by
    the projection backward through the (tile, sender) lists) against the gh-rows form
    (NLAM_SENDER_PARTS=0) on an m2g-like layer: no edge update, batch-invariant edge operand and
    receivers' Pr, separate nodes, B = 4 (the batch-sum form must be the one that runs), through
    the C++ sequencer and launch by launch; every gradient to split-bf16 summation accuracy
    (the partial sums come from the indicator product that also forms the receiver sums)."""
    from neural_lam_amd import inet_seq, ops
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(11)
    n_r, n_s, B = 65536, 7000, 4      # (8,192 tiles: the batch-sum form is balanced from ~8 tiles per wave)
    rec = torch.arange(4 * n_r) // 4                     # four in-edges per receiver (create_graph.py:508)
    send = ((rec // 9) + torch.randint(0, 3, (4 * n_r,), generator=gen)) % n_s
    torch.manual_seed(5)
    net = InteractionNet(torch.stack((send + n_r, rec)), 64, update_edges=False).cuda()
    t = net.tables
    assert t.has_sender_parts and t.n_sender_parts < t.M
    assert ops.lib.nlam_edge_bwd_parts_supported(t.ntiles, B, 64) == 1
    xs = torch.randn(B, n_s, 64, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    xr = torch.randn(B, n_r, 64, device="cuda", generator=torch.Generator("cuda").manual_seed(2))
    e1 = torch.randn(1, 4 * n_r, 64, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    cot = torch.randn(B, n_r, 64, device="cuda", generator=torch.Generator("cuda").manual_seed(4))

    def run(parts, seq):
        monkeypatch.setenv("NLAM_SENDER_PARTS", "1" if parts else "0")
        monkeypatch.setattr(inet_seq, "ENABLED", seq)
        for p in net.parameters():
            p.grad = None
        a, b, c = (v.clone().requires_grad_(True) for v in (xs, xr, e1))
        ops.PROFILER = None if seq else ops.KernelProfiler()
        try:
            out = net(a, b, c.expand(B, -1, -1))
            (out * cot).sum().backward()
            stats = ops.PROFILER.collect() if ops.PROFILER is not None else {}
        finally:
            ops.PROFILER = None
        return [out.detach(), a.grad, b.grad, c.grad] + [p.grad.clone() for p in net.parameters()], stats

    want, _ = run(False, True)
    for seq in (True, False):
        got, stats = run(True, seq)
        for a, b in zip(got, want):
            scale = float(b.abs().max()) + 1e-30
            assert float((a - b).abs().max()) <= 2e-5 * scale, float((a - b).abs().max()) / scale
    # byte accounting of the launch-by-launch form: the gather reads one row per (tile, sender) pair
    assert any(k.startswith("nlam_edge_bwd") for k in stats)
