"""hidden_dim 128 ("wide") kernels through the C ABI: operator and MLP parity with the CPU oracle
(oracle/nlam_oracle.py, pinned by the reference goldens), all call shapes the models use
(shared / separate senders, sum / mean with empty receivers, update_edges on / off, stride-0
batch-invariant inputs, narrow embedder inputs, the 17-wide output map), and proof that the wide
kernels -- not the generic GEMM sequence -- are what runs.  fp32 bars (default bf16x3
arithmetic): forward 1e-4, gradients 1e-3 relative to max|ref|.

The same cases run at hidden 256 on the feature-split kernels (csrc/fused_fs.hip) in bf16
arithmetic -- NLAM_WIDE_D=256 NLAM_MFMA=bf16, in a process of its own (tools/parity_wide.py,
driven by test_gpu_mfma_modes.py) -- with bf16 bars: forward 1e-2, gradients 5e-2 of max|ref|
(measured worst: 4e-3 forward, 2e-2 on a Hi-LAM parameter gradient; two CPU bf16-autocast runs of
the reference that differ only in thread count are 1e-3 .. 1e-2 apart themselves)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

D = int(os.environ.get("NLAM_WIDE_D", "128"))
_BF16 = os.environ.get("NLAM_MFMA", "") == "bf16"
FWD_BAR, GRAD_BAR = (1e-2, 5e-2) if _BF16 else (1e-4, 1e-3)


def rel(a, b):
    return float((a.detach().cpu() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30))


def _edges(gen, n_s, n_r, M, shared, empty_receivers=False, empty_span=0):
    rec = torch.randint(0, n_r, (M,), generator=gen)
    send = torch.randint(0, n_s, (M,), generator=gen)
    if empty_receivers:   # (mean clamps the count to 1); in-degrees stay <= 32
        for r_empty in (7, 20, 30):
            rec[rec == r_empty] = r_empty + 1
    if empty_span:        # receivers 1 .. empty_span without in-edges: whole tiles with no edge
        lo = torch.arange(M) % (n_r - empty_span - 1) + empty_span + 1
        rec = torch.where((rec >= 1) & (rec <= empty_span), lo, rec)
    rec[0], rec[1], send[2] = 0, n_r - 1, 0
    return torch.stack((send + (0 if shared else n_r), rec))


@pytest.mark.parametrize("shared,upd,aggr,B", [(True, True, "sum", 2), (False, False, "mean", 3),
                                                (False, True, "mean", 1), (True, False, "sum", 2),
                                                (True, True, "mean", 5)])
def test_wide_interaction_net_vs_oracle(shared, upd, aggr, B):
    _inet_case(shared, upd, aggr, B)


@pytest.mark.parametrize("shared", [False, True])
def test_wide_tiles_of_empty_receivers(shared):
    """70 consecutive receivers without in-edges: receiver-aligned tiles that hold no edge at
    all (ne == 0) must leave zero aggregates / gradients and read nothing out of range."""
    _inet_case(shared, True, "mean", 2, empty_span=70)


@pytest.mark.parametrize("shared,upd,aggr", [(True, True, "sum"), (False, False, "mean")])
def test_wide_high_in_degree_runs_on_virtual_receivers(shared, upd, aggr):
    """Receivers with 60+ / 33 / exactly 64 in-edges: the wide kernels run on virtual receivers of
    <= 32 edges (graph.VirtualReceivers) with a fold-back stage, not on the generic GEMM sequence."""
    _inet_case(shared, upd, aggr, 2, high_degree=True)


def _inet_case(shared, upd, aggr, B, empty_span=0, high_degree=False):
    import nlam_oracle as orc
    from neural_lam_amd import wide
    from neural_lam_amd.interaction_net import InteractionNet

    d = D
    gen = torch.Generator().manual_seed(7 + B)
    n_s, n_r, M = (45, 45, 410) if shared else (70, 38, 333)
    if empty_span:
        n_s, n_r, M = (120, 120, 700) if shared else (90, 110, 600)
    ei = _edges(gen, n_s, n_r, M, shared, empty_receivers=(aggr == "mean"), empty_span=empty_span)
    if high_degree:
        snd, rcv = ei[0] - (0 if shared else n_r), ei[1].clone()
        rcv[10:72] = 9          # 60+ in-edges
        rcv[72:105] = 12        # 33+
        rcv[rcv == 15] = 14
        rcv[105:169] = 15       # exactly 64
        rcv[0], rcv[1] = 0, n_r - 1
        ei = torch.stack((snd + (0 if shared else n_r), rcv))
    torch.manual_seed(5)
    net = InteractionNet(ei, d, update_edges=upd, aggr=aggr)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    sd = {f"n.{k}": v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    send = torch.randn(B, n_s, d, generator=gen)
    rec = send if shared else torch.randn(B, n_r, d, generator=gen)
    edge = torch.randn(B, M, d, generator=gen)
    cr, ce = torch.randn(B, n_r, d, generator=gen), torch.randn(B, M, d, generator=gen)

    def run(fwd, s, r, e, cr, ce):
        out = fwd(s, r, e)
        o_r, o_e = out if upd else (out, None)
        loss = (o_r * cr).sum() + ((o_e * ce).sum() if upd else 0.0)
        loss.backward()
        return o_r, o_e

    sc = send.clone().requires_grad_(True)
    rc = sc if shared else rec.clone().requires_grad_(True)
    ec = edge.clone().requires_grad_(True)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    w_r, w_e = run(lambda s, r, e: orc.interaction_net(osd, "n", ei, s, r, e, update_edges=upd,
                                                       aggr=aggr), sc, rc, ec, cr, ce)
    sg = send.cuda().requires_grad_(True)
    rg = sg if shared else rec.cuda().requires_grad_(True)
    eg = edge.cuda().requires_grad_(True)
    assert wide.inet_eligible(net, sg, rg, eg)
    if high_degree:
        assert net.tables.virtual is not None and net.tables.max_in_degree >= 64
    g_r, g_e = run(net, sg, rg, eg, cr.cuda(), ce.cuda())
    assert rel(g_r, w_r) < FWD_BAR, ("rec", rel(g_r, w_r))
    if upd:
        assert rel(g_e, w_e) < FWD_BAR, ("edge", rel(g_e, w_e))
    assert rel(sg.grad, sc.grad) < GRAD_BAR and rel(eg.grad, ec.grad) < GRAD_BAR, (
        rel(sg.grad, sc.grad), rel(eg.grad, ec.grad))
    if not shared:
        assert rel(rg.grad, rc.grad) < GRAD_BAR, rel(rg.grad, rc.grad)
    for k, p in net.named_parameters():
        assert rel(p.grad, osd[f"n.{k}"].grad) < GRAD_BAR, (k, rel(p.grad, osd[f"n.{k}"].grad))


@pytest.mark.parametrize("upd,aggr", [(True, "sum"), (False, "mean")])
def test_wide_split_mlps_vs_oracle(upd, aggr):
    """SplitMLPs at this width (HiLAMParallel's operator, hi_lam_parallel.py:26-53,
    interaction_net.py:134-163): one edge MLP per edge chunk, one node MLP per node range -- the
    wide kernels run chunk by chunk (wide.apply_inet_split); forward and every gradient vs the CPU
    oracle, and no generic GEMM may run."""
    import nlam_oracle as orc
    from neural_lam_amd import ops, wide
    from neural_lam_amd.interaction_net import InteractionNet

    d, B = D, 2
    gen = torch.Generator().manual_seed(31)
    n, M = 50, 400
    ei = _edges(gen, n, n, M, shared=True, empty_receivers=(aggr == "mean"))
    kw = dict(update_edges=upd, aggr=aggr, edge_chunk_sizes=[150, 130, 120], aggr_chunk_sizes=[20, 30])
    torch.manual_seed(6)
    net = InteractionNet(ei, d, **kw)
    sd = {f"n.{k}": v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.cuda()
    x = torch.randn(B, n, d, generator=gen)
    e = torch.randn(B, M, d, generator=gen)
    cr, ce = torch.randn(B, n, d, generator=gen), torch.randn(B, M, d, generator=gen)
    xc, ec = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
    want = orc.interaction_net(sd, "n", ei, xc, xc, ec, **kw)
    wl = (want[0] * cr).sum() + (want[1] * ce).sum() if upd else (want * cr).sum()
    names = [k for k, _ in net.named_parameters()]
    wg = torch.autograd.grad(wl, [xc, ec] + [sd[f"n.{k}"] for k in names])
    xg, eg = x.cuda().requires_grad_(True), e.cuda().requires_grad_(True)
    assert wide.inet_split_eligible(net, xg, xg, eg)
    ops.PROFILER = ops.KernelProfiler()
    try:
        got = net(xg, xg, eg)
        gl = (got[0] * cr.cuda()).sum() + (got[1] * ce.cuda()).sum() if upd else (got * cr.cuda()).sum()
        gl.backward()
        ran = {k.split("@")[0] for k in ops.PROFILER.collect()}
    finally:
        ops.PROFILER = None
    assert "nlam_tail_fwd" in ran and "nlam_gemm" not in ran, ran
    g0, w0 = (got[0], want[0]) if upd else (got, want)
    assert rel(g0, w0) < FWD_BAR, rel(g0, w0)
    if upd:
        assert rel(got[1], want[1]) < FWD_BAR
    assert rel(xg.grad, wg[0]) < GRAD_BAR and rel(eg.grad, wg[1]) < GRAD_BAR
    for (k, p), w in zip(net.named_parameters(), wg[2:]):
        assert rel(p.grad, w) < GRAD_BAR, (k, rel(p.grad, w))


def test_wide_stride0_batch_inputs_match_oracle():
    """g2m-style call at d = 128: receiver and edge reps are stride-0 expands (expand_to_batch)."""
    import nlam_oracle as orc
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(21)
    d, B, n_s, n_r, M = D, 3, 60, 20, 150
    ei = _edges(gen, n_s, n_r, M, shared=False)
    torch.manual_seed(3)
    net = InteractionNet(ei, d, update_edges=False)
    sd = {f"n.{k}": v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    send = torch.randn(B, n_s, d, generator=gen)
    rec1, edge1 = torch.randn(n_r, d, generator=gen), torch.randn(M, d, generator=gen)
    sc, rc, ec = (t.clone().requires_grad_(True) for t in (send, rec1, edge1))
    want = orc.interaction_net(sd, "n", ei, sc, rc.unsqueeze(0).expand(B, -1, -1),
                               ec.unsqueeze(0).expand(B, -1, -1), update_edges=False)
    (want ** 2).sum().backward()
    sg, rg, eg = (t.cuda().requires_grad_(True) for t in (send, rec1, edge1))
    got = net(sg, rg.unsqueeze(0).expand(B, -1, -1), eg.unsqueeze(0).expand(B, -1, -1))
    (got ** 2).sum().backward()
    assert rel(got, want) < FWD_BAR
    assert (rel(sg.grad, sc.grad) < GRAD_BAR and rel(rg.grad, rc.grad) < GRAD_BAR
            and rel(eg.grad, ec.grad) < GRAD_BAR)


MLP_CASES = [
    ([3, D, D], True, False, 77, 1),        # edge / mesh embedders (narrow static features)
    ([17, D, D], True, False, 100, 2),      # grid embedder (unaligned 17-wide rows)
    ([48, D, D], True, False, 70, 2),       # grid embedder, aligned 48-wide rows
    ([D, D, D], True, True, 131, 2),        # encoding_grid_mlp with fused residual
    ([D, D, 17], False, False, 90, 2),      # output map: no LayerNorm, 17 columns
    ([D, D, 5], False, False, 33, 1),
]


@pytest.mark.parametrize("blueprint,ln,res,rows,B", MLP_CASES)
def test_wide_mlp_vs_oracle(blueprint, ln, res, rows, B):
    import nlam_oracle as orc
    from neural_lam_amd import ops, utils, wide

    gen = torch.Generator().manual_seed(sum(blueprint) + rows)
    torch.manual_seed(9)
    mlp = utils.make_mlp(blueprint, layer_norm=ln)
    with torch.no_grad():
        for p in mlp.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    sd = {f"m.{k}": v.detach().clone().requires_grad_(True) for k, v in mlp.state_dict().items()}
    mlp = mlp.cuda()
    x = torch.randn(B, rows, blueprint[0], generator=gen)
    cot = torch.randn(B, rows, blueprint[-1], generator=gen)
    xc = x.clone().requires_grad_(True)
    want = orc.mlp(sd, "m", xc, 1, layer_norm=ln)
    if res:
        want = xc + want
    (want * cot).sum().backward()
    xg = x.cuda().requires_grad_(True)
    assert wide.mlp_eligible(mlp, xg, xg if res else None)
    ops.PROFILER = ops.KernelProfiler()
    try:
        got = mlp(xg, res=xg) if res else mlp(xg)
        (got * cot.cuda()).sum().backward()
        stats = ops.PROFILER.collect()
    finally:
        ops.PROFILER = None
    names = {k.split("@")[0] for k in stats}
    assert {"nlam_tail_fwd", "nlam_tail_bwd", "nlam_wide_outer"} <= names
    assert rel(got, want) < FWD_BAR
    assert rel(xg.grad, xc.grad) < GRAD_BAR
    for k, p in mlp.named_parameters():
        assert rel(p.grad, sd[f"m.{k}"].grad) < GRAD_BAR, k


INET_CASES = [(True, True, "sum", 2), (False, False, "mean", 3), (False, True, "mean", 1),
              (True, False, "sum", 2)]


@pytest.mark.skipif(D != 128, reason="launch budget of the 128 path")
def test_wide_paths_are_taken_at_hidden_128():
    """d = 128, hidden_layers = 1: the wide kernels must be the ones that run -- no generic GEMM
    on the 128-wide operands (narrow embedder inputs may use it for their first layer only)."""
    from neural_lam_amd import fused, ops, wide
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(0)
    ei = torch.stack((torch.randint(0, 30, (200,), generator=gen),
                      torch.randint(0, 30, (200,), generator=gen)))
    ei[0, 0], ei[1, 0], ei[1, 1] = 0, 0, 29
    net = InteractionNet(ei, 128).cuda()
    x = torch.randn(2, 30, 128, device="cuda", requires_grad=True)
    e = torch.randn(2, 200, 128, device="cuda", requires_grad=True)
    assert not fused.inet_eligible(net, x, x, e) and wide.inet_eligible(net, x, x, e)
    ops.PROFILER = ops.KernelProfiler()
    try:
        o_x, o_e = net(x, x, e)
        (o_x.sum() + o_e.sum()).backward()
        stats = ops.PROFILER.collect()
    finally:
        ops.PROFILER = None
    names = {k.split("@")[0] for k in stats}
    assert {"nlam_lin_fwd_multi", "nlam_tail_fwd", "nlam_tail_fwd_pre", "nlam_tail_bwd",
            "nlam_lin_bwd_data_multi", "nlam_wide_outer_multi", "nlam_segment_sum"} <= names
    # launch budget of one wide InteractionNet (forward 3: the node update's aggregate projection
    # rides inside its tail launch; backward 9 incl. the slab reduction)
    assert sum(v["calls"] for v in stats.values()) <= 13, stats
    assert "nlam_gemm" not in names and "nlam_layernorm_fwd" not in names and "nlam_lin_fwd" not in names
    assert stats["nlam_tail_fwd@inet"]["calls"] == 1 and stats["nlam_tail_fwd_pre@inet"]["calls"] == 1
    assert stats["nlam_tail_bwd@inet"]["calls"] == 2


@pytest.mark.skipif(D != 128, reason="hidden 128 only")
@pytest.mark.parametrize("B,n_r,expect_pre", [(2, 30, True), (4, 8192, True), (1, 32 * 1024 - 5, True),
                                              (4, 8193, False), (3, 517, True)])
def test_node_update_with_the_projection_inside_the_tail(B, n_r, expect_pre, monkeypatch):
    """nlam_tail_fwd_pre (h = a + agg V1b^T inside the node tail; one row tile per wave, i.e. at
    most 1024 tiles) against the two-launch sequence it replaces (nlam_lin_fwd + nlam_tail_fwd):
    outputs and every gradient of an InteractionNet to fp32 rounding order (the product is
    accumulated onto `a` instead of beside it), at the tile-count boundary on both sides, with a
    batch-invariant receiver operand, and the fall-back above the boundary."""
    from neural_lam_amd import ops
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(n_r)
    M = 4 * n_r if n_r < 10000 else n_r + 100
    rec = torch.arange(M) % n_r
    send = torch.randint(0, n_r, (M,), generator=gen)
    torch.manual_seed(3)
    net = InteractionNet(torch.stack((send, rec)), 128).cuda()
    x = torch.randn(1 if B == 3 else B, n_r, 128, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    e = torch.randn(B, M, 128, device="cuda", generator=torch.Generator("cuda").manual_seed(2))

    def run(on):
        monkeypatch.setenv("NLAM_TAIL_PRE", "1" if on else "0")
        for p in net.parameters():
            p.grad = None
        xs, es = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
        ops.PROFILER = ops.KernelProfiler()
        try:
            o_x, o_e = net(xs.expand(B, -1, -1) if xs.shape[0] == 1 else xs, xs.expand(B, -1, -1)
                           if xs.shape[0] == 1 else xs, es)
            (o_x.square().sum() + o_e.sum()).backward()
            stats = ops.PROFILER.collect()
        finally:
            ops.PROFILER = None
        return (o_x.detach(), o_e.detach(), xs.grad, es.grad, *[p.grad.clone() for p in net.parameters()]), stats

    got, s_on = run(True)
    want, s_off = run(False)
    assert any(k.startswith("nlam_tail_fwd_pre") for k in s_on) == expect_pre
    assert not any(k.startswith("nlam_tail_fwd_pre") for k in s_off)
    if expect_pre:
        assert sum(v["calls"] for v in s_on.values()) == sum(v["calls"] for v in s_off.values()) - 1
    for a, b in zip(got, want):
        assert rel(a.cpu(), b.cpu()) < 2e-5, rel(a.cpu(), b.cpu())


@pytest.mark.skipif(D != 128, reason="model-level check at hidden 128")
@pytest.mark.parametrize("ar_steps", [1, 2])
def test_deferred_slab_reductions_leave_the_training_step_unchanged(ar_steps, monkeypatch):
    """Hi-LAM-128 training step with the slab reductions of the hidden-128 layers flushed once per
    AR step (glue.DeferGrad + ops.slab_batch(defer=True)) and with one reduction launch per layer
    (NLAM_DEFER_REDUCE=0): identical loss, parameter gradients equal to fp32 summation order (the
    same reductions, later), fewer reduction launches, nothing left pending, and no parameter without a gradient."""
    import tempfile

    import numpy as np

    from neural_lam_amd import graphgen, ops, synthetic, wide
    from neural_lam_amd.models import MODELS

    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(38, 35, 5000.0), 3, True)
        n = info["num_grid"]
        gen = torch.Generator().manual_seed(0)
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(n, 4, generator=gen).numpy(), np.zeros(7), np.ones(7), np.zeros(7),
            np.ones(7), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        torch.manual_seed(1)
        model = MODELS["hi_lam"](synthetic.model_args(graph="g", hidden_dim=128, processor_layers=2),
                                 config=None, datastore=ds).cuda()
    batch = synthetic.random_batch(2, ar_steps, n, n_state=7, n_forcing_window=6, device="cuda")

    def run(on, outer_rows=1 << 40):
        monkeypatch.setenv("NLAM_DEFER_REDUCE", "1" if on else "0")
        monkeypatch.setattr(wide, "DEFER_OUTER_ROWS", outer_rows)
        for p in model.parameters():
            p.grad = None
        ops.PROFILER = ops.KernelProfiler()
        try:
            loss = model.training_step(batch)
            assert not wide.PROXY
            loss.backward()
            stats = ops.PROFILER.collect()
        finally:
            ops.PROFILER = None
        assert not ops._DEFERRED and not wide._DEFERRED_OUTERS and not ops._FLUSH_HOOKS
        assert all(p.grad is not None for p in model.parameters())
        n_red = sum(v["calls"] for k, v in stats.items() if k.startswith("nlam_reduce_slabs"))
        return float(loss.detach()), [p.grad.clone() for p in model.parameters()], n_red

    l_on, g_on, r_on = run(True)
    l_off, g_off, r_off = run(False)
    # the weight-gradient problems of the step merged 24 to a launch (default) against one launch
    # per layer: the same products (a slab count per problem that depends on what shares its
    # launch: fp32 summation order only)
    l_lay, g_lay, _ = run(True, outer_rows=0)
    assert l_lay == l_on
    for (k, _), a, b in zip(model.named_parameters(), g_on, g_lay):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 5e-6 * scale, (k, float((a - b).abs().max()) / scale)
    # (every merged weight-gradient launch is followed by the reduction of its own slabs; a
    # multi-step rollout embeds the static features up front, undeferred)
    assert r_on <= 0.6 * r_off, (r_on, r_off)
    assert l_on == l_off
    # (the same slab sums; which reduction kernel form takes a segment depends on what shares its
    # launch, so the order of the fp32 additions -- not the set of addends -- may differ)
    for (k, _), a, b in zip(model.named_parameters(), g_on, g_off):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-6 * scale, (k, float((a - b).abs().max()) / scale)


@pytest.mark.skipif(D != 128, reason="hidden 128 only")
def test_static_embedders_in_multi_problem_launches_match_one_launch_each(monkeypatch):
    """wide.embed_many (the tails of all static-feature embedders of a Hi-LAM-128 model in
    multi-problem launches, forward and backward: nlam_mlp_tail_{fwd,bwd}_multi) against one
    launch per embedder (NLAM_EMBED_MULTI=0): embeddings bitwise equal (the same kernel body per
    problem), loss equal, parameter gradients to fp32 summation order (dgamma / dbeta slab counts
    follow the share of the launch), fewer launches."""
    import tempfile

    import numpy as np

    from neural_lam_amd import graphgen, ops, synthetic
    from neural_lam_amd.models import MODELS

    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(38, 35, 5000.0), 3, True)
        n = info["num_grid"]
        gen = torch.Generator().manual_seed(0)
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(n, 4, generator=gen).numpy(), np.zeros(7), np.ones(7), np.zeros(7),
            np.ones(7), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        torch.manual_seed(1)
        model = MODELS["hi_lam"](synthetic.model_args(graph="g", hidden_dim=128, processor_layers=2),
                                 config=None, datastore=ds).cuda()
    batch = synthetic.random_batch(2, 1, n, n_state=7, n_forcing_window=6, device="cuda")
    from neural_lam_amd import fused

    def run(on):
        monkeypatch.setenv("NLAM_EMBED_MULTI", "1" if on else "0")
        with torch.no_grad():
            emb = fused.embed_many(model.static_embedders())
        for p in model.parameters():
            p.grad = None
        ops.PROFILER = ops.KernelProfiler()
        try:
            loss = model.training_step(batch)
            loss.backward()
            stats = ops.PROFILER.collect()
        finally:
            ops.PROFILER = None
        return emb, float(loss.detach()), [p.grad.clone() for p in model.parameters()], stats

    e_on, l_on, g_on, s_on = run(True)
    e_off, l_off, g_off, s_off = run(False)
    assert any(k.startswith("nlam_tail_fwd_multi") for k in s_on) and any(
        k.startswith("nlam_tail_bwd_multi") for k in s_on)
    assert not any(k.startswith(("nlam_tail_fwd_multi", "nlam_tail_bwd_multi")) for k in s_off)
    assert sum(v["calls"] for v in s_on.values()) <= sum(v["calls"] for v in s_off.values()) - 12
    assert set(e_on) == set(e_off)
    for k in e_on:
        assert torch.equal(e_on[k], e_off[k]), k
    assert l_on == l_off
    for (k, _), a, b in zip(model.named_parameters(), g_on, g_off):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 5e-6 * scale, (k, float((a - b).abs().max()) / scale)
