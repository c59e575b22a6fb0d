"""nlam_inet_fwd / nlam_inet_bwd (csrc/inet_host.cpp: one host call per InteractionNet) against the
same layer issued launch by launch from Python: the two must run the same kernels on the same
operands, so every output and gradient is compared BITWISE; shared / separate nodes,
update_edges on / off, mean / sum, batch-invariant operands (the expand_to_batch views of the
reference, ar_model.py:204-209)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(net, s, r, e, cr, ce, upd, seq):
    from neural_lam_amd import inet_seq

    old = inet_seq.ENABLED
    inet_seq.ENABLED = seq
    try:
        leaves = [t.clone().requires_grad_(True) for t in (s, r, e)]
        sx = leaves[0] if s.dim() == 3 else leaves[0].unsqueeze(0).expand(cr.shape[0], -1, -1)
        rx = sx if r is s else (leaves[1] if r.dim() == 3 else leaves[1].unsqueeze(0).expand(cr.shape[0], -1, -1))
        ex = leaves[2] if e.dim() == 3 else leaves[2].unsqueeze(0).expand(cr.shape[0], -1, -1)
        for p in net.parameters():
            p.grad = None
        out = net(sx, rx, ex)
        if upd:
            loss = (out[0] * cr).sum() + (out[1] * ce).sum()
        else:
            loss = (out * cr).sum()
        loss.backward()
        outs = [o.detach().clone() for o in (out if upd else (out,))]
        grads = [l.grad.clone() if l.grad is not None else None for l in leaves]
        return outs, grads, [p.grad.clone() for p in net.parameters()]
    finally:
        inet_seq.ENABLED = old


CASES = [
    # shared, update_edges, aggr, B, batch-invariant (send, rec, edge)
    (True, True, "sum", 3, (False, False, False)),
    (True, True, "mean", 2, (False, False, True)),     # m2m first layer: edge embedding expanded
    (False, False, "sum", 4, (False, True, True)),     # g2m: mesh receivers + edges expanded
    (False, False, "mean", 2, (False, False, True)),   # m2g
    (False, True, "sum", 2, (False, False, False)),    # Hi-LAM up / down
    (False, True, "sum", 3, (True, False, True)),
    (True, False, "sum", 2, (False, False, False)),
]


@pytest.mark.parametrize("shared,upd,aggr,B,inv", CASES)
def test_sequencer_matches_launch_by_launch_bitwise(shared, upd, aggr, B, inv):
    from neural_lam_amd import inet_seq, ops
    from neural_lam_amd.interaction_net import InteractionNet

    if not (ops.lin_multi_supported() and ops.node_chain_supported()):
        pytest.skip("the sequencer covers the split-bf16 mode")
    gen = torch.Generator().manual_seed(17 + B)
    d = 64
    n_s, n_r, M = (77, 77, 600) if shared else (90, 53, 500)
    rec = torch.randint(0, n_r, (M,), generator=gen)
    send = torch.randint(0, n_s, (M,), generator=gen)
    rec[rec == 5] = 6          # an empty receiver
    send[send == 9] = 10       # a node that sends nothing
    rec[0], rec[1], send[2] = 0, n_r - 1, 0
    ei = torch.stack((send + (0 if shared else n_r), rec))
    torch.manual_seed(3)
    net = InteractionNet(ei, d, update_edges=upd, aggr=aggr).cuda()

    def rnd(rows, invariant):
        shape = (rows, d) if invariant else (B, rows, d)
        return torch.randn(*shape, generator=gen).cuda()

    s = rnd(n_s, inv[0] and not shared)
    r = s if shared else rnd(n_r, inv[1])
    e = rnd(M, inv[2])
    cr, ce = torch.randn(B, n_r, d, generator=gen).cuda(), torch.randn(B, M, d, generator=gen).cuda()
    a = _run(net, s, r, e, cr, ce, upd, seq=True)
    b = _run(net, s, r, e, cr, ce, upd, seq=False)
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    for x, y in zip(a[1], b[1]):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x, y)
    for x, y in zip(a[2], b[2]):
        assert torch.equal(x, y)


def test_sequencer_is_the_default_path():
    """One host call per direction: no per-launch ctypes calls are made (counted through the
    profiler hook, which the sequencer bypasses), and the per-launch path is used under it."""
    from neural_lam_amd import inet_seq, ops
    from neural_lam_amd.interaction_net import InteractionNet

    if not (ops.lin_multi_supported() and ops.node_chain_supported()):
        pytest.skip("the sequencer covers the split-bf16 mode")
    assert inet_seq.ENABLED
    gen = torch.Generator().manual_seed(1)
    ei = torch.stack((torch.randint(0, 40, (300,), generator=gen), torch.randint(0, 40, (300,), generator=gen)))
    ei[0, 0], ei[1, 0], ei[1, 1] = 0, 0, 39
    net = InteractionNet(ei, 64).cuda()
    x = torch.randn(2, 40, 64, device="cuda", requires_grad=True)
    e = torch.randn(2, 300, 64, device="cuda", requires_grad=True)
    calls = []
    orig = ops._launch

    def spy(name, fn, args, flops=0.0, nbytes=0.0):
        calls.append(name)
        return orig(name, fn, args, flops, nbytes)

    ops._launch = spy
    try:
        ox, oe = net(x, x, e)
        (ox.sum() + oe.sum()).backward()
    finally:
        ops._launch = orig
    assert calls == [], calls
