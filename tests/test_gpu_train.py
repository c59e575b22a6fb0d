"""GPU tests of the training-step glue: flat AdamW kernel vs torch.optim.AdamW,
FlatParams packing, and a short GraphLAM training run whose loss decreases."""
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flat_adamw_matches_torch():
    from neural_lam_amd import parallel

    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Linear(13, 3))
    mine = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Linear(13, 3))
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda()
    flat = parallel.FlatParams(mine)
    opt = parallel.FlatAdamW(flat, lr=1e-2)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.9, 0.95))
    gen = torch.Generator().manual_seed(1)
    for _ in range(5):
        x = torch.randn(11, 7, generator=gen)
        topt.zero_grad()
        ref(x).pow(2).sum().backward()
        topt.step()
        flat.zero_grad()
        mine(x.cuda()).pow(2).sum().backward()
        flat.pack_grads()
        opt.step()
    for (k, a), (_, b) in zip(ref.state_dict().items(), mine.state_dict().items()):
        assert torch.allclose(a, b.cpu(), rtol=1e-5, atol=1e-6), k


def test_pack_grads_one_launch_matches_concatenation():
    """FlatParams.pack_grads on device tensors (nlam_pack_segments, one launch): every gradient
    lands bit-exactly in its slice -- odd sizes, a slice larger than one chunk, a gradient that
    is an unaligned view, a missing gradient (zeros), sub-ranges, and a second step whose
    gradients live elsewhere (the address table is rebuilt) -- and the padding stays zero."""
    from neural_lam_amd import parallel

    torch.manual_seed(3)
    shapes = [(7,), (13, 5), (64, 192), (1,), (129, 131), (3, 3)]
    params = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(*s)) for s in shapes]).cuda()
    flat = parallel.FlatParams(params)
    for step in range(2):
        big = torch.randn(10_000, device="cuda")
        for i, p in enumerate(flat.params):
            p.grad = torch.randn_like(p) if i != 3 else None
        flat.params[1].grad = big[3 : 3 + 65].view(13, 5)          # 4-byte aligned view
        flat.grad.fill_(-7.0) if step == 0 else None               # stale content is overwritten
        if step == 0:   # (the pads are never written: they must start as zeros)
            for i, p in enumerate(flat.params):
                end = flat.offsets[i + 1] if i + 1 < len(flat.params) else flat.numel
                flat.grad[flat.offsets[i] + p.numel() : end] = 0
        flat.pack_grads()
        for i, p in enumerate(flat.params):
            got = flat.grad[flat.offsets[i] : flat.offsets[i] + p.numel()]
            want = p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), device="cuda")
            assert torch.equal(got, want), (step, i)
            end = flat.offsets[i + 1] if i + 1 < len(flat.params) else flat.numel
            assert not flat.grad[flat.offsets[i] + p.numel() : end].any()
    # a sub-range leaves the other slices alone
    before = flat.grad.clone()
    flat.params[2].grad = torch.ones_like(flat.params[2])
    flat.params[0].grad = torch.ones_like(flat.params[0])
    flat.pack_grads(2, 3)
    a, b = flat.span(2, 3)
    assert torch.equal(flat.grad[:a], before[:a]) and torch.equal(flat.grad[b:], before[b:])
    assert bool((flat.grad[a : a + 64 * 192] == 1).all())


def test_graphlam_training_loss_decreases():
    from neural_lam_amd import graphgen, parallel, synthetic
    from neural_lam_amd.models import GraphLAM
    import numpy as np

    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(30, 28, 5000.0), None, False)
        n = info["num_grid"]
        gen = torch.Generator().manual_seed(0)
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(n, 1, generator=gen).numpy(), np.zeros(5), np.ones(5), np.zeros(5),
            np.ones(5), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        torch.manual_seed(1)
        model = GraphLAM(synthetic.model_args(graph="g", hidden_dim=64, processor_layers=2),
                         config=None, datastore=ds).cuda()
    flat = parallel.FlatParams(model)
    opt = parallel.FlatAdamW(flat, lr=2e-3)
    batch = synthetic.random_batch(2, 2, n, n_state=5, n_forcing_window=6, device="cuda")
    losses = []
    for _ in range(12):
        flat.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        flat.pack_grads()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.8 * losses[0], losses


@pytest.mark.parametrize("model_name", ["graph_lam", "hi_lam"])
def test_ar_checkpointing_matches_plain_rollout(model_name):
    """args.ar_checkpoint recomputes each predict_step in backward: same loss
    and the same parameter gradients as the plain BPTT rollout (ar_model.py:220-267).  Hi-LAM:
    the level representations are joined through glue.tee inside the recomputed step as well."""
    from neural_lam_amd import graphgen, synthetic
    from neural_lam_amd.models import MODELS
    import numpy as np

    GraphLAM = MODELS[model_name]
    hier = model_name != "graph_lam"
    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(30, 28, 5000.0),
                                     3 if hier else None, hier)
        n = info["num_grid"]
        gen = torch.Generator().manual_seed(0)
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(n, 1, generator=gen).numpy(), np.zeros(5), np.ones(5), np.zeros(5),
            np.ones(5), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        models = []
        for ck in (False, True):
            torch.manual_seed(1)
            models.append(GraphLAM(
                synthetic.model_args(graph="g", hidden_dim=64, processor_layers=2, ar_checkpoint=ck),
                config=None, datastore=ds).cuda())
    batch = synthetic.random_batch(2, 4, n, n_state=5, n_forcing_window=6, device="cuda")
    losses, grads, peaks = [], [], []
    for m in models:
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = m.training_step(batch)
        loss.backward()
        torch.cuda.synchronize()
        peaks.append(torch.cuda.max_memory_allocated() - base)
        losses.append(float(loss.detach()))
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
    # ar_steps = 4: the recomputed rollout holds ONE step's activations at a time
    assert peaks[1] < 0.6 * peaks[0], peaks
    assert abs(losses[0] - losses[1]) <= 1e-6 * abs(losses[0])
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max() + 1e-30), k


def test_validation_and_test_steps_metrics():
    """validation_step / test_step / epoch-end aggregation (reference ar_model.py:324-452,
    610-696, without plotting): logged losses and aggregated RMSE / MAE equal a CPU
    recomputation from the model's own rollout."""
    from neural_lam_amd import graphgen, metrics, synthetic
    from neural_lam_amd.models import GraphLAM
    import numpy as np

    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(30, 28, 5000.0), None, False)
        n = info["num_grid"]
        gen = torch.Generator().manual_seed(0)
        std = np.array([1.0, 2.0, 0.5, 1.5, 3.0])
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(n, 1, generator=gen).numpy(), np.zeros(5), std, np.zeros(5),
            np.ones(5), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        torch.manual_seed(1)
        model = GraphLAM(synthetic.model_args(graph="g", hidden_dim=64, processor_layers=1,
                                              val_steps_to_log=[1, 2, 5]),
                         config=None, datastore=ds).cuda()
    batch = synthetic.random_batch(3, 2, n, n_state=5, n_forcing_window=6, device="cuda")
    with torch.no_grad():
        pred, target, pred_std, _ = model.common_step(batch)
    p, t, s = pred.cpu(), target.cpu(), pred_std.cpu()
    mask = model.interior_mask_bool.cpu()
    step_loss = metrics.wmse(p, t, s, mask=mask).mean(0)
    model.validation_step(batch, 0)
    logged = model.last_logged
    assert set(logged) == {"val_loss_unroll1", "val_loss_unroll2", "val_mean_loss"}   # 5 > T: clipped
    assert abs(logged["val_loss_unroll2"] - float(step_loss[1])) < 1e-5 * abs(float(step_loss[1]))
    assert abs(logged["val_mean_loss"] - float(step_loss.mean())) < 1e-5 * float(step_loss.mean())
    model.on_validation_epoch_end()
    want_rmse = torch.sqrt(metrics.mse(p, t, s, mask=mask, sum_vars=False).mean(0)) * torch.tensor(std, dtype=torch.float32)
    got = model.eval_results["val"]["val_rmse"].cpu()
    assert got.shape == (2, 5) and torch.allclose(got, want_rmse, rtol=1e-4, atol=1e-6)
    assert model.val_metrics["mse"] == []
    model.args.val_steps_to_log = [1, 2]
    model.test_step(batch, 0)
    model.test_step(batch, 1)
    out = model.on_test_epoch_end()
    want_mae = metrics.mae(p, t, s, mask=mask, sum_vars=False).mean(0) * torch.tensor(std, dtype=torch.float32)
    assert torch.allclose(out["test_mae"].cpu(), want_mae, rtol=1e-4, atol=1e-6)
    assert torch.allclose(out["test_rmse"].cpu(), want_rmse, rtol=1e-4, atol=1e-6)
    assert out["test_mean_spatial_loss"].shape == (2, n)
    # checkpoints from before the encoder refactoring (ar_model.py:698-716)
    ckpt = {"state_dict": {"g2m_gnn.grid_mlp.0.weight": torch.zeros(1), "x": torch.zeros(1)}}
    model.on_load_checkpoint(ckpt)
    assert set(ckpt["state_dict"]) == {"encoding_grid_mlp.0.weight", "x"}


def test_output_std_head_and_nll_vs_reference_fixture():
    """output_std branch (base_graph_model.py:161-177) + nll training loss (metrics.py:166-190,
    ar_model.py:294-298) as HIP kernels vs the fixture computed with the reference's
    metrics.nll (tests/golden/make_golden.py: make_output_std_case)."""
    import os

    from conftest import GOLDEN
    from neural_lam_amd import glue

    fx = torch.load(os.path.join(GOLDEN, "aux_output_std.pt"), weights_only=False)
    x = fx["net_out"].cuda().requires_grad_(True)
    prev, target = fx["prev"].cuda(), fx["target"].cuda()
    state, std = glue.StdHead.apply(prev, x, fx["diff_std"].cuda(), fx["diff_mean"].cuda())
    keep = fx["interior"].float().cuda()
    lead = state.numel() // (state.shape[-1] * state.shape[-2])
    loss = glue.MaskedNLL.apply(state, target, std, keep, 1.0 / (float(keep.sum()) * lead))
    loss.backward()

    def rel(a, b):
        return float((a.detach().cpu() - b).abs().max() / (b.abs().max() + 1e-30))

    assert rel(state, fx["state"]) < 1e-6 and rel(std, fx["pred_std"]) < 1e-6
    assert abs(float(loss) - fx["loss"]) < 1e-5 * abs(fx["loss"])
    assert rel(x.grad, fx["grad_net_out"]) < 1e-5


@pytest.mark.parametrize("widths", [(17, 17, 18, 4), (5, 5, 6, 1), (40, 40, 30)])
def test_concat_rows_matches_torch_cat(widths):
    """Grid feature concat of predict_step (base_graph_model.py:116-124) as one kernel: equals
    torch.cat, reads a stride-0 expand_to_batch source in place, gradients are the slices."""
    from neural_lam_amd import glue
    from neural_lam_amd.models.ar_model import ARModel

    B, N = 3, 211
    gen = torch.Generator().manual_seed(sum(widths))
    srcs = [torch.randn(B, N, w, generator=gen).cuda().requires_grad_(True) for w in widths[:-1]]
    static = torch.randn(N, widths[-1], generator=gen).cuda().requires_grad_(True)
    exp = ARModel.expand_to_batch(static, B)
    got = glue.ConcatRows.apply(*srcs, exp)
    want = torch.cat([*srcs, static.unsqueeze(0).expand(B, -1, -1)], dim=-1)
    assert torch.equal(got, want)
    cot = torch.randn(B, N, sum(widths), generator=gen).cuda()
    g_got = torch.autograd.grad((got * cot).sum(), [*srcs, static])
    g_want = torch.autograd.grad((want * cot).sum(), [*srcs, static])
    for a, b in zip(g_got, g_want):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("prev_grad", [False, True])
def test_state_step_matches_residual_then_boundary_mix(prev_grad):
    """glue.StateStep (one kernel each way) on SLICED batch tensors against the two-step torch
    expression of base_graph_model.py:174-177 + ar_model.py:244-247, values and gradients."""
    from neural_lam_amd import glue

    gen = torch.Generator().manual_seed(5)
    B, N, F = 3, 517, 17
    init = torch.randn(B, 2, N, F, generator=gen).cuda()
    targets = torch.randn(B, 4, N, F, generator=gen).cuda()
    net = torch.randn(B, N, F, generator=gen).cuda().requires_grad_(True)
    mask = (torch.rand(N, 1, generator=gen) < 0.3).float().cuda()
    std, mean = (torch.rand(F, generator=gen) + 0.5).cuda(), torch.randn(F, generator=gen).cuda()
    prev = init[:, 1]
    if prev_grad:
        prev = prev.clone().requires_grad_(True)
    got = glue.StateStep.apply(prev, net, targets[:, 2], mask, std, mean)
    w = torch.randn(B, N, F, generator=gen).cuda()
    (got * w).sum().backward()
    g_net, g_prev = net.grad.clone(), (prev.grad.clone() if prev_grad else None)
    net.grad = None
    if prev_grad:
        prev.grad = None
    want = mask * targets[:, 2] + (1.0 - mask) * (prev + net * std + mean)
    (want * w).sum().backward()
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    assert torch.allclose(g_net, net.grad, rtol=1e-6, atol=1e-7)
    if prev_grad:
        assert torch.allclose(g_prev, prev.grad, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("prev_grad,use_state,use_loss,N", [(False, False, True, 517), (True, True, True, 517),
                                                             (False, True, False, 33), (True, True, True, 70001)])
def test_state_step_loss_matches_state_step_then_masked_wmse(prev_grad, use_state, use_loss, N):
    """glue.StateStepLoss (state step + this AR step's loss term, one kernel each way) against
    (a) glue.StateStep followed by glue.MaskedWMSE and (b) the torch expression in float64;
    values and gradients, with the new state and / or the loss term feeding the objective (the
    last rollout step has no consumer of its state; a term may go unused), twice in a row (the
    last-block ticket of the loss reduction must re-arm itself)."""
    from neural_lam_amd import glue

    gen = torch.Generator().manual_seed(6)
    B, F = 3, 17
    init = torch.randn(B, 2, N, F, generator=gen).cuda()
    targets = torch.randn(B, 4, N, F, generator=gen).cuda()
    mask = (torch.rand(N, 1, generator=gen) < 0.3).float().cuda()
    keep = (1.0 - mask[:, 0]).contiguous()
    wf = (torch.rand(F, generator=gen) + 0.2).cuda()
    std, mean = (torch.rand(F, generator=gen) + 0.5).cuda(), torch.randn(F, generator=gen).cuda()
    cot = torch.randn(B, N, F, generator=gen).cuda()
    lscale = 1.0 / (float(keep.sum()) * B * 2)

    def run(kind):
        net = torch.randn(B, N, F, generator=torch.Generator().manual_seed(7)).cuda().requires_grad_(True)
        prev = init[:, 1].clone().requires_grad_(True) if prev_grad else init[:, 1]
        truth = targets[:, 2]
        if kind == "fused":
            new, loss = glue.StateStepLoss.apply(prev, net, truth, mask, std, mean, keep, wf, lscale)
        elif kind == "split":
            new = glue.StateStep.apply(prev, net, truth, mask, std, mean)
            loss = glue.MaskedWMSE.apply(new, truth, keep, wf, lscale)
        else:
            new = mask.double() * truth.double() + (1.0 - mask.double()) * (
                prev.double() + net.double() * std.double() + mean.double())
            loss = (keep.double()[None, :, None] * wf.double() * (new - truth.double()) ** 2).sum() * lscale
        obj = (3.0 * loss if use_loss else 0.0) + ((new * cot).sum() if use_state else 0.0)
        obj.backward()
        return new.detach(), loss.detach(), net.grad, (prev.grad if prev_grad else None)

    for _ in range(2):
        got, split, want = run("fused"), run("split"), run("f64")
        assert torch.equal(got[0], split[0])
        assert torch.allclose(got[0].double(), want[0], rtol=1e-6, atol=1e-6)
        assert abs(float(got[1]) - float(want[1])) <= 2e-6 * abs(float(want[1]))
        assert abs(float(split[1]) - float(want[1])) <= 2e-6 * abs(float(want[1]))
        for a, b, c in zip(got[2:], split[2:], want[2:]):
            if a is None:
                assert b is None and c is None
                continue
            scale = float(c.abs().max()) + 1e-30
            assert float((a.double() - c.double()).abs().max()) <= 2e-6 * scale
            assert float((a - b).abs().max()) <= 2e-6 * scale


@pytest.mark.parametrize("model_name,ar_steps", [("graph_lam", 1), ("graph_lam", 3), ("hi_lam", 2)])
def test_training_step_is_unchanged_by_the_fused_loss_term(model_name, ar_steps, monkeypatch):
    """One training step with each AR step's loss term taken from the state-step kernel
    (NLAM_FUSE_LOSS default) and with the separate loss kernel over the stacked prediction: the
    same loss and parameter gradients to fp32 summation-order accuracy, fewer launches."""
    import tempfile

    import numpy as np

    from neural_lam_amd import graphgen, ops, synthetic
    from neural_lam_amd.models import MODELS

    hier = model_name != "graph_lam"
    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(38, 35, 5000.0),
                                     3 if hier else None, hier)
        n = info["num_grid"]
        gen = torch.Generator().manual_seed(0)
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(n, 4, generator=gen).numpy(), np.zeros(7), np.ones(7), np.zeros(7),
            np.ones(7), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        torch.manual_seed(1)
        model = MODELS[model_name](synthetic.model_args(graph="g", hidden_dim=64, processor_layers=2),
                                   config=None, datastore=ds).cuda()
    batch = synthetic.random_batch(2, ar_steps, n, n_state=7, n_forcing_window=6, device="cuda")

    def run(on):
        monkeypatch.setenv("NLAM_FUSE_LOSS", "1" if on else "0")
        for p in model.parameters():
            p.grad = None
        ops.PROFILER = ops.KernelProfiler()
        try:
            loss = model.training_step(batch)
            loss.backward()
            stats = ops.PROFILER.collect()
        finally:
            ops.PROFILER = None
        assert model._loss_tap is None and not model._loss_terms
        return float(loss.detach()), [p.grad.clone() for p in model.parameters()], stats

    l_on, g_on, s_on = run(True)
    l_off, g_off, s_off = run(False)
    assert sum(v["calls"] for k, v in s_on.items() if k.startswith("nlam_state_step_wmse_fwd")) == ar_steps
    assert not [k for k in s_on if k.split("@")[0] in ("nlam_state_step", "nlam_state_step_bwd",
                                                        "nlam_wmse_fwd", "nlam_wmse_bwd")], sorted(s_on)
    assert not any(k.startswith("nlam_state_step_wmse") for k in s_off)
    n_on, n_off = (sum(v["calls"] for v in s.values()) for s in (s_on, s_off))
    assert n_on <= n_off - 2, (n_on, n_off)   # (the separate loss is one entry over the stacked prediction)
    assert abs(l_on - l_off) <= 2e-6 * abs(l_off)
    for (k, _), a, b in zip(model.named_parameters(), g_on, g_off):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale, (k, float((a - b).abs().max()) / scale)


@pytest.mark.parametrize("n,shape", [(2, (3, 50, 64)), (7, (4, 737, 64)), (11, (2, 33, 5))])
def test_sum_many_matches_chain_of_adds(n, shape):
    """glue.sum_many (nlam_sum_many: one pass over up to 8 terms, chained beyond) against the
    chain of additions it replaces in the SplitMLPs operator: same fixed order, so bit-identical
    for up to 8 terms; every term's gradient is the incoming gradient."""
    from neural_lam_amd import glue

    gen = torch.Generator().manual_seed(n)
    terms = [torch.randn(*shape, generator=gen).cuda().requires_grad_(True) for _ in range(n)]
    got = glue.sum_many(terms)
    want = terms[0]
    for t in terms[1:]:
        want = want + t
    assert torch.equal(got, want) if n <= 8 else torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    w = torch.randn(*shape, generator=gen).cuda()
    (got * w).sum().backward()
    for t in terms:
        assert torch.equal(t.grad, w)


@pytest.mark.parametrize("case", ["one_consumer", "two_consumers", "reversed_roles", "stale_value"])
def test_tee_gradient_is_the_sum_in_every_usage(case):
    """glue.tee hands a tensor to two consumers whose backward kernels fold the other branch's
    gradient into their own store.  Whatever the usage -- the give alias consumed once (the
    models' case: no torch add may be needed), consumed TWICE (autograd sums both on the give
    alias, only one of them was folded), the aliases used the other way round (nothing can be
    folded), a value left over by a backward pass that pruned the Tee node -- the gradient of the
    teed tensor must equal the plain autograd sum."""
    from neural_lam_amd import glue
    from neural_lam_amd.interaction_net import InteractionNet
    from neural_lam_amd.utils import make_mlp

    torch.manual_seed(50)
    gen = torch.Generator().manual_seed(51)
    d, B, n_s, n_r, M = 64, 2, 90, 40, 300
    ei = torch.stack((torch.randint(0, n_s, (M,), generator=gen) + n_r,
                      torch.randint(0, n_r, (M,), generator=gen)))
    ei[0, 0], ei[0, 1], ei[1, 2], ei[1, 3] = n_r, n_r + n_s - 1, 0, n_r - 1
    net = InteractionNet(ei, d, update_edges=False).cuda()
    mlp = make_mlp([d, d, d]).cuda()
    x0 = torch.randn(B, n_s, d, generator=gen).cuda()
    rec = torch.randn(B, n_r, d, generator=gen).cuda()
    edge = torch.randn(B, M, d, generator=gen).cuda()
    w = torch.randn(B, n_s, d, generator=gen).cuda()
    c1 = torch.randn(B, n_r, d, generator=gen).cuda()
    c2 = torch.randn(B, n_s, d, generator=gen).cuda()

    def loss_of(a, b):
        if case == "reversed_roles":
            a, b = b, a
        out = (net(a, rec, edge) * c1).sum() + (mlp(b, res=b) * c2).sum()
        if case == "two_consumers":
            out = out + (b * w).sum()
        return out

    xr = x0.clone().requires_grad_(True)
    h = xr * 1.5
    loss_of(h, h).backward()
    want = xr.grad.clone()

    xt = x0.clone().requires_grad_(True)
    a, b = glue.tee(xt * 1.5)
    assert a is not b
    if case == "stale_value":
        # a pass that stops at the give alias leaves the encoding MLP's input gradient in the
        # slot; the next full pass must not fold that old tensor in a second time
        side = (mlp(b, res=b) * c2).sum()
        torch.autograd.grad(side, b, retain_graph=True)
        (side + (net(a, rec, edge) * c1).sum()).backward()
    else:
        loss_of(a, b).backward()
    err = float((xt.grad - want).abs().max() / want.abs().max())
    assert err < 1e-5, (case, err)
