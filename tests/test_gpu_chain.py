"""The fused node-side chain of consecutive InteractionNets on shared nodes (the reference's
processor, models/graph_lam.py:51-57,88; csrc/fused16_node.hip) against (a) the same layers run
one at a time through the per-layer HIP path and (b) the CPU oracle, forward and every gradient;
plus the launch count the fusion is for (<= 6 launches per layer, forward + backward).
fp32-grade bars: forward 1e-4, gradients 1e-3 relative to max|ref|."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.detach().cpu() - b.detach().cpu()).abs().max() / (b.detach().abs().max().cpu() + 1e-30))


def make_chain(n_layers, N, M, d, seed, isolated=()):
    from neural_lam_amd.interaction_net import InteractionNet
    from neural_lam_amd.models.graph_lam import ProcessorSequential

    gen = torch.Generator().manual_seed(seed)
    rec = torch.randint(0, N, (M,), generator=gen)
    send = torch.randint(0, N, (M,), generator=gen)
    for n in isolated:            # nodes that send nothing (empty sender lists)
        send[send == n] = (n + 1) % N
    rec[0], rec[1], send[2] = 0, N - 1, 0
    ei = torch.stack((send, rec))
    torch.manual_seed(seed)
    nets = [InteractionNet(ei, d) for _ in range(n_layers)]
    with torch.no_grad():
        for net in nets:
            for p in net.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=gen))
    return ProcessorSequential(nets), ei, gen


def run(proc, x, e, cx, ce, chain):
    from neural_lam_amd._lib import lib

    # (the library's own default mask, whatever kernel families it names: a hard-coded one went
    # stale when the round-4 edge backward got its bits and left every later test of the process
    # on the older kernel)
    default_mask = lib.nlam_set_k16(-1)
    lib.nlam_set_k16(default_mask if chain else (default_mask & ~256))
    try:
        xs = x.clone().requires_grad_(True)
        es = e.clone().requires_grad_(True)
        for p in proc.parameters():
            p.grad = None
        ox, oe = proc(xs, es.expand(x.shape[0], -1, -1) if es.shape[0] == 1 else es)
        ((ox * cx).sum() + (oe * ce).sum()).backward()
        return ox.detach(), oe.detach(), xs.grad, es.grad, {k: p.grad.clone() for k, p in proc.named_parameters()}
    finally:
        lib.nlam_set_k16(default_mask)


@pytest.mark.parametrize("n_layers,N,M,B,e_batch", [(2, 45, 410, 2, True), (4, 333, 2900, 3, False),
                                                    (3, 16, 100, 1, True), (3, 97, 700, 4, True)])
def test_chain_matches_layer_by_layer_path(n_layers, N, M, B, e_batch):
    from neural_lam_amd import ops

    if not ops.node_chain_supported():
        pytest.skip("exact-fp32 mode has no chain kernels")
    proc, ei, gen = make_chain(n_layers, N, M, 64, 3 + n_layers, isolated=(5, N - 2))
    proc = proc.cuda()
    x = torch.randn(B, N, 64, generator=gen).cuda()
    e = torch.randn(B if e_batch else 1, M, 64, generator=gen).cuda()
    cx, ce = torch.randn(B, N, 64, generator=gen).cuda(), torch.randn(B, M, 64, generator=gen).cuda()
    a = run(proc, x, e, cx, ce, chain=True)
    b = run(proc, x, e, cx, ce, chain=False)
    assert rel(a[0], b[0]) < 1e-5 and rel(a[1], b[1]) < 1e-5
    assert rel(a[2], b[2]) < 1e-4 and rel(a[3], b[3]) < 1e-4
    for k in a[4]:
        assert rel(a[4][k], b[4][k]) < 1e-4, k
    # deterministic: no atomics
    a2 = run(proc, x, e, cx, ce, chain=True)
    assert torch.equal(a[0], a2[0]) and torch.equal(a[2], a2[2])
    assert all(torch.equal(a[4][k], a2[4][k]) for k in a[4])


def test_chain_vs_cpu_oracle():
    import nlam_oracle as orc
    from neural_lam_amd import ops

    if not ops.node_chain_supported():
        pytest.skip("exact-fp32 mode has no chain kernels")
    n_layers, N, M, B, d = 3, 61, 500, 2, 64
    proc, ei, gen = make_chain(n_layers, N, M, d, 21, isolated=(7,))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in proc.state_dict().items()}
    x = torch.randn(B, N, d, generator=gen)
    e = torch.randn(B, M, d, generator=gen)
    cx, ce = torch.randn(B, N, d, generator=gen), torch.randn(B, M, d, generator=gen)
    xc, ec = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
    hx, he = xc, ec
    for i in range(n_layers):
        hx, he = orc.interaction_net(sd, f"module_{i}", ei, hx, hx, he)
    names = [k for k, _ in proc.named_parameters()]
    want = torch.autograd.grad((hx * cx).sum() + (he * ce).sum(), [xc, ec] + [sd[k] for k in names])
    proc = proc.cuda()
    got = run(proc, x.cuda(), e.cuda(), cx.cuda(), ce.cuda(), chain=True)
    assert rel(got[0], hx) < 1e-4 and rel(got[1], he) < 1e-4
    assert rel(got[2], want[0]) < 1e-3 and rel(got[3], want[1]) < 1e-3
    for k, w in zip(names, want[2:]):
        assert rel(got[4][k], w) < 1e-3, k


def test_chain_launch_count():
    """<= 6 launches per layer (forward + backward) on the chain, and the per-layer projection /
    scatter entry points are gone from the inner layers."""
    from neural_lam_amd import ops

    if not ops.node_chain_supported():
        pytest.skip("exact-fp32 mode has no chain kernels")
    n_layers, N, M, B = 4, 200, 1700, 2
    proc, ei, gen = make_chain(n_layers, N, M, 64, 5)
    proc = proc.cuda()
    for net in proc:
        net.tables.tag = "m2m"
    x = torch.randn(B, N, 64, generator=gen).cuda()
    e = torch.randn(B, M, 64, generator=gen).cuda()
    cx, ce = torch.randn(B, N, 64, generator=gen).cuda(), torch.randn(B, M, 64, generator=gen).cuda()
    run(proc, x, e, cx, ce, chain=True)
    ops.PROFILER = ops.KernelProfiler()
    try:
        run(proc, x, e, cx, ce, chain=True)
        names = [n for (n, *_rest) in ops.PROFILER.pending]
        ops.PROFILER.collect()
    finally:
        ops.PROFILER = None
    assert len(names) <= 6 * n_layers, names
    assert names.count("nlam_lin_fwd@m2m") == 1 and "nlam_lin_bwd@m2m" not in names
    assert "nlam_segment_sum@m2m" not in names
    assert names.count("nlam_node_fwd@m2m") == n_layers - 1
    assert names.count("nlam_node_bwd@m2m") == n_layers
