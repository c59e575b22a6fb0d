"""Parity of the HIP InteractionNet (through the C ABI) with the golden vectors
captured from the reference's own source (tests/golden/op_*.pt), forward and
backward.  Tolerances (fp32, relative to max|ref|): forward 1e-4, grads 1e-3
(SURVEY.md section 8c)."""
import glob
import os

import pytest
import torch

from conftest import GOLDEN, elementwise_excess, fixture_files, l2_rel

pytestmark = pytest.mark.gpu
OP_FILES = fixture_files("op_*.pt")   # the bf16-autocast ones: tests/test_gpu_mfma_modes.py


def rel(a, b):
    return float((a.detach().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


def run_case(fx, force_generic):
    from neural_lam_amd import fused
    from neural_lam_amd.interaction_net import InteractionNet

    old = fused.FORCE_GENERIC
    fused.FORCE_GENERIC = force_generic
    try:
        kw, shared = fx["kwargs"], fx["shared"]
        net = InteractionNet(fx["edge_index"], fx["d"], **kw)
        missing = net.load_state_dict(fx["state_dict"], strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        net = net.cuda()
        s = fx["send"].cuda().requires_grad_(True)
        r = s if shared else fx["rec"].cuda().requires_grad_(True)
        e = fx["edge"].cuda().requires_grad_(True)
        out = net(s, r, e)
        if kw.get("update_edges", True):
            o_rec, o_edge = out
            loss = (o_rec * fx["cot_rec"].cuda()).sum() + (o_edge * fx["cot_edge"].cuda()).sum()
            assert rel(o_edge, fx["out_edge"]) < 1e-4
        else:
            o_rec = out
            loss = (o_rec * fx["cot_rec"].cuda()).sum()
        assert rel(o_rec, fx["out_rec"]) < 1e-4
        loss.backward()
        assert rel(s.grad, fx["grad_send"]) < 1e-3
        assert rel(e.grad, fx["grad_edge"]) < 1e-3
        if not shared:
            assert rel(r.grad, fx["grad_rec"]) < 1e-3
        for k, p in net.named_parameters():
            assert rel(p.grad, fx["grad_params"][k]) < 1e-3, k
        # the same bars in the 2-norm and element by element (|err| <= tol (|ref| + rms(ref))):
        # small-magnitude entries of a tensor are held too, not only its largest ones
        pairs = [("out_rec", o_rec, fx["out_rec"], 1e-4), ("grad_send", s.grad, fx["grad_send"], 1e-3),
                 ("grad_edge", e.grad, fx["grad_edge"], 1e-3)]
        pairs += [(k, p.grad, fx["grad_params"][k], 1e-3) for k, p in net.named_parameters()]
        for name, got, want, tol in pairs:
            assert l2_rel(got, want) < tol, name
            assert elementwise_excess(got, want, tol) < 1.0, name
    finally:
        fused.FORCE_GENERIC = old


@pytest.mark.parametrize("path", OP_FILES, ids=[os.path.basename(p)[:-3] for p in OP_FILES])
def test_generic_path_vs_reference_golden(path):
    run_case(torch.load(path, weights_only=False), force_generic=True)


@pytest.mark.parametrize("path", OP_FILES, ids=[os.path.basename(p)[:-3] for p in OP_FILES])
def test_default_path_vs_reference_golden(path):
    run_case(torch.load(path, weights_only=False), force_generic=False)


def test_stride0_batch_inputs_match_oracle():
    """g2m-style call: receiver and edge reps are stride-0 expands."""
    import nlam_oracle as orc
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(11)
    d, B, n_s, n_r, M = 32, 3, 60, 20, 150
    ei = torch.stack((torch.randint(0, n_s, (M,), generator=gen) + n_r, torch.randint(0, n_r, (M,), generator=gen)))
    ei[0, 0], ei[1, 0], ei[1, 1] = n_r, 0, n_r - 1
    torch.manual_seed(3)
    net = InteractionNet(ei, d, update_edges=False)
    sd = {f"n.{k}": v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    send = torch.randn(B, n_s, d, generator=gen)
    rec1 = torch.randn(n_r, d, generator=gen)
    edge1 = torch.randn(M, d, generator=gen)
    sc, rc, ec = (t.clone().requires_grad_(True) for t in (send, rec1, edge1))
    want = orc.interaction_net(sd, "n", ei, sc, rc.unsqueeze(0).expand(B, -1, -1),
                               ec.unsqueeze(0).expand(B, -1, -1), update_edges=False)
    (want ** 2).sum().backward()
    sg, rg, eg = (t.cuda().requires_grad_(True) for t in (send, rec1, edge1))
    got = net(sg, rg.unsqueeze(0).expand(B, -1, -1), eg.unsqueeze(0).expand(B, -1, -1))
    (got ** 2).sum().backward()
    assert rel(got, want) < 1e-4
    assert rel(sg.grad, sc.grad) < 1e-3 and rel(rg.grad, rc.grad) < 1e-3 and rel(eg.grad, ec.grad) < 1e-3


@pytest.mark.parametrize("d", [64, 16])
def test_leading_dims_2d_and_4d_match_3d(d):
    """interaction_net.py:86-115 takes any leading dims (node dim = -2): 2-D and
    (B, T, N, d) inputs give the 3-D result, on the fused (d=64) and generic (d=16) paths."""
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(5)
    n, M = 40, 300
    ei = torch.stack((torch.randint(0, n, (M,), generator=gen), torch.randint(0, n, (M,), generator=gen)))
    ei[0, 0], ei[1, 0], ei[1, 1] = 0, 0, n - 1
    torch.manual_seed(2)
    net = InteractionNet(ei, d).cuda()
    x = torch.randn(2, 3, n, d, generator=gen).cuda()
    e = torch.randn(2, 3, M, d, generator=gen).cuda()
    ox, oe = net(x, x, e)
    assert ox.shape == x.shape and oe.shape == e.shape
    rx, re_ = net(x.reshape(6, n, d), x.reshape(6, n, d), e.reshape(6, M, d))
    assert torch.equal(ox.reshape(6, n, d), rx) and torch.equal(oe.reshape(6, M, d), re_)
    ox2, oe2 = net(x[0, 0], x[0, 0], e[0, 0])
    assert ox2.shape == (n, d) and oe2.shape == (M, d)
    assert rel(ox2, rx[0].cpu()) < 1e-6 and rel(oe2, re_[0].cpu()) < 1e-6


def test_cpu_tensors_fail_loudly():
    from neural_lam_amd.interaction_net import InteractionNet

    ei = torch.tensor([[3, 4, 5, 3], [0, 1, 2, 2]])
    net = InteractionNet(ei, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.randn(1, 3, 8), torch.randn(1, 3, 8), torch.randn(1, 4, 8))


@pytest.mark.parametrize("d,deg_lo,deg_hi,upd", [(64, 1, 2, True), (64, 1, 1, False), (64, 3, 5, True),
                                                 (128, 1, 2, True)])
def test_low_in_degree_tiles_match_oracle(d, deg_lo, deg_hi, upd):
    """In-degree 1-2: a 32-edge receiver-aligned tile then owns up to 32 receivers.  The fused
    backward loads the tile's receiver rows once (16 prefetched, the rest on demand) and expands
    them to the edge slots from LDS; this is the shape that takes the on-demand branch."""
    import nlam_oracle as orc
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(d + deg_lo + deg_hi)
    B, n_s, n_r = 2, 90, 150
    deg = torch.randint(deg_lo, deg_hi + 1, (n_r,), generator=gen)
    rec = torch.repeat_interleave(torch.arange(n_r), deg)
    M = rec.numel()
    perm = torch.randperm(M, generator=gen)          # edges in arbitrary order
    rec = rec[perm]
    send = torch.randint(0, n_s, (M,), generator=gen) + n_r
    ei = torch.stack((send, rec))
    torch.manual_seed(4)
    net = InteractionNet(ei, d, update_edges=upd, aggr="mean")
    sd = {f"n.{k}": v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.cuda()
    s, r, e = (torch.randn(B, n, d, generator=gen) for n in (n_s, n_r, M))
    cr, ce = torch.randn(B, n_r, d, generator=gen), torch.randn(B, M, d, generator=gen)

    def run(fwd, s, r, e, cr, ce):
        out = fwd(s, r, e)
        o_r, o_e = out if upd else (out, None)
        ((o_r * cr).sum() + ((o_e * ce).sum() if upd else 0.0)).backward()
        return o_r, o_e

    sc, rc, ec = (t.clone().requires_grad_(True) for t in (s, r, e))
    w_r, w_e = run(lambda a, b, c: orc.interaction_net(sd, "n", ei, a, b, c, update_edges=upd,
                                                       aggr="mean"), sc, rc, ec, cr, ce)
    sg, rg, eg = (t.cuda().requires_grad_(True) for t in (s, r, e))
    g_r, g_e = run(net, sg, rg, eg, cr.cuda(), ce.cuda())
    assert rel(g_r, w_r) < 1e-4
    if upd:
        assert rel(g_e, w_e) < 1e-4
    assert rel(sg.grad, sc.grad) < 1e-3 and rel(rg.grad, rc.grad) < 1e-3 and rel(eg.grad, ec.grad) < 1e-3
    for k, p in net.named_parameters():
        assert rel(p.grad, sd[f"n.{k}"].grad) < 1e-3, k


@pytest.mark.parametrize("shared,upd", [(True, True), (False, True), (False, False)])
def test_tiles_of_empty_receivers_d64(shared, upd):
    """70 consecutive receivers without in-edges at hidden 64: whole receiver-aligned tiles with
    no edge (ne == 0); aggregates / gradients of those receivers are exactly the node-update
    of a zero aggregate, and nothing is read out of range.  Fused path, vs the CPU oracle."""
    import nlam_oracle as orc
    from neural_lam_amd import fused
    from neural_lam_amd.interaction_net import InteractionNet

    gen = torch.Generator().manual_seed(31)
    d, B = 64, 2
    n_s, n_r, M = (130, 130, 800) if shared else (90, 125, 700)
    span = 70
    rec = torch.randint(0, n_r, (M,), generator=gen)
    send = torch.randint(0, n_s, (M,), generator=gen)
    lo = torch.arange(M) % (n_r - span - 1) + span + 1
    rec = torch.where((rec >= 1) & (rec <= span), lo, rec)
    rec[0], rec[1], send[2] = 0, n_r - 1, 0
    ei = torch.stack((send + (0 if shared else n_r), rec))
    torch.manual_seed(4)
    net = InteractionNet(ei, d, update_edges=upd, aggr="mean")
    sd = {f"n.{k}": v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.cuda()
    s = torch.randn(B, n_s, d, generator=gen)
    r = s if shared else torch.randn(B, n_r, d, generator=gen)
    e = torch.randn(B, M, d, generator=gen)
    cr, ce = torch.randn(B, n_r, d, generator=gen), torch.randn(B, M, d, generator=gen)

    def loss_of(out, cr, ce):
        if upd:
            return (out[0] * cr).sum() + (out[1] * ce).sum(), out[0]
        return (out * cr).sum(), out

    sc = s.clone().requires_grad_(True)
    rc = sc if shared else r.clone().requires_grad_(True)
    ec = e.clone().requires_grad_(True)
    wl, w_rec = loss_of(orc.interaction_net(sd, "n", ei, sc, rc, ec, update_edges=upd, aggr="mean"), cr, ce)
    names = [k for k, _ in net.named_parameters()]
    want = torch.autograd.grad(wl, [sc, ec] + [sd[f"n.{k}"] for k in names])
    sg = s.cuda().requires_grad_(True)
    rg = sg if shared else r.cuda().requires_grad_(True)
    eg = e.cuda().requires_grad_(True)
    assert fused.inet_eligible(net, sg, rg, eg)
    gl, g_rec = loss_of(net(sg, rg, eg), cr.cuda(), ce.cuda())
    gl.backward()
    assert rel(g_rec, w_rec.detach()) < 1e-4
    assert rel(sg.grad, want[0]) < 1e-3 and rel(eg.grad, want[1]) < 1e-3
    for (k, p), w in zip(net.named_parameters(), want[2:]):
        assert rel(p.grad, w) < 1e-3, k
