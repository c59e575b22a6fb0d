"""GPU unit tests of the generic C-ABI kernels against plain torch fp32 (CPU)
references of the same op.  Tolerances are fp32 round-off scaled by the
reduction length."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.detach().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from neural_lam_amd import ops as o

    return o


def test_mfma_lane_maps(ops):
    """A(32x64) B(64x32) with exactly representable integers: any wrong operand
    or accumulator lane map of v_mfma_f32_32x32x2_f32 changes the result."""
    out = ops.mfma_probe().cpu()
    i = torch.arange(32).view(32, 1, 1)
    k = torch.arange(64).view(1, 64, 1)
    j = torch.arange(32).view(1, 1, 32)
    A = ((i + 1) + 100 * (k % 7)).to(torch.float64)
    Bm = ((j - 3) * ((k % 5) + 1)).to(torch.float64)
    want = (A * Bm).sum(1)
    assert torch.equal(out.to(torch.float64), want)


@pytest.mark.parametrize(
    "M,N,K", [(1, 1, 1), (7, 5, 3), (64, 64, 64), (130, 70, 33), (1000, 64, 192), (300, 17, 64),
              (1000, 128, 384), (513, 256, 100), (129, 97, 1030)]
)
def test_gemm_linear(ops, M, N, K):
    gen = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    x = torch.randn(M, K, generator=gen)
    W = torch.randn(N, K, generator=gen)
    b = torch.randn(N, generator=gen)
    xd, Wd, bd = x.cuda(), W.cuda(), b.cuda()
    y = torch.empty(M, N, device="cuda")
    ops.linear_fwd(ops.mat(xd), Wd, bd, ops.mat(y))
    want = x @ W.T + b
    assert rel(y, want) < 2e-6 * max(1, K ** 0.5)
    gy = torch.randn(M, N, generator=gen)
    gx = torch.empty(M, K, device="cuda")
    ops.linear_bwd_data(ops.mat(gy.cuda()), Wd, ops.mat(gx))
    assert rel(gx, gy @ W) < 2e-6 * max(1, N ** 0.5)
    dW = torch.empty(N, K, device="cuda")
    db = torch.empty(N, device="cuda")
    ops.linear_bwd_weight(ops.mat(gy.cuda()), ops.mat(xd), dW, db)
    assert rel(dW, gy.T @ x) < 2e-6 * max(1, M ** 0.5)
    assert rel(db, gy.sum(0)) < 2e-6 * max(1, M ** 0.5)


def test_gemm_splitk_and_strided(ops):
    gen = torch.Generator().manual_seed(3)
    rows = 70000
    cat = torch.randn(rows, 192, generator=gen)
    gy = torch.randn(rows, 64, generator=gen)
    cd, gd = cat.cuda(), gy.cuda()
    # column slice of a wider buffer as the operand
    dW = torch.empty(64, 64, device="cuda")
    db = torch.empty(64, device="cuda")
    ops.linear_bwd_weight(ops.mat(gd), ops.mat(cd, 64, 64), dW, db)
    want = gy.double().T @ cat[:, 64:128].double()
    assert rel(dW.double(), want) < 1e-5
    assert rel(db.double(), gy.double().sum(0)) < 1e-5


def test_silu_layernorm_colsum(ops):
    gen = torch.Generator().manual_seed(5)
    for rows, d in [(5, 4), (1000, 64), (333, 128), (70, 200)]:
        x = torch.randn(rows, d, generator=gen) * 3
        gy = torch.randn(rows, d, generator=gen)
        gam = 1 + 0.1 * torch.randn(d, generator=gen)
        bet = 0.1 * torch.randn(d, generator=gen)
        res = torch.randn(rows, d, generator=gen)
        xd, gyd = x.cuda(), gy.cuda()
        y = torch.empty_like(xd)
        ops.silu_fwd(xd, y)
        assert rel(y, torch.nn.functional.silu(x)) < 1e-6
        xr = x.clone().requires_grad_(True)
        torch.nn.functional.silu(xr).backward(gy)
        gx = torch.empty_like(xd)
        ops.silu_bwd(xd, gyd, gx)
        assert rel(gx, xr.grad) < 2e-6
        # layernorm with residual
        ops.layernorm_fwd(ops.mat(xd), gam.cuda(), bet.cuda(), ops.mat(res.cuda()), ops.mat(y))
        xr = x.clone().requires_grad_(True)
        gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
        ref = res + torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5)
        assert rel(y, ref) < 2e-6
        ref.backward(gy)
        gz = torch.empty_like(xd)
        dg, dbt = torch.empty(d, device="cuda"), torch.empty(d, device="cuda")
        ops.layernorm_bwd(ops.mat(xd), gam.cuda(), ops.mat(gyd), ops.mat(gz), dg, dbt)
        assert rel(gz, xr.grad) < 1e-5
        assert rel(dg, gr.grad) < 1e-5 and rel(dbt, br.grad) < 1e-5
        cs = torch.empty(d, device="cuda")
        ops.colsum(ops.mat(xd), cs)
        assert rel(cs, x.sum(0)) < 1e-5


@pytest.mark.parametrize("d", [4, 16, 64, 128, 256, 20])
def test_gather_segment(ops, d):
    from neural_lam_amd.graph import EdgeTables

    gen = torch.Generator().manual_seed(d)
    B, n_send, n_rec, M = 3, 40, 25, 300
    send = torch.randint(0, n_send, (M,), generator=gen)
    rec = torch.randint(0, n_rec, (M,), generator=gen)
    rec[rec == 7] = 8  # receiver 7 has no in-edges
    t = EdgeTables(send, rec, n_send, n_rec).cuda()
    x = torch.randn(B, n_send, d, generator=gen)
    out = torch.empty(B, M, 2 * d, device="cuda")
    ops.gather_rows(ops.mat(x.cuda()), t.send, ops.mat(out, d, d))
    assert torch.equal(out[:, :, d:].cpu(), x[:, send])
    msg = torch.randn(B, M, d, generator=gen)
    agg = torch.empty(B, n_rec, d, device="cuda")
    ops.segment_sum(ops.mat(msg.cuda()), t.csr_rowptr, t.csr_eid, ops.mat(agg))
    want = torch.zeros(B, n_rec, d).index_add_(1, rec, msg)
    assert rel(agg, want) < 1e-6
    ops.segment_sum(ops.mat(msg.cuda()), t.csr_rowptr, t.csr_eid, ops.mat(agg), scale=t.inv_deg)
    deg = torch.zeros(n_rec).index_add_(0, rec, torch.ones(M)).clamp(min=1)
    assert rel(agg, want / deg.view(1, -1, 1)) < 1e-6
    gs = torch.empty(B, n_send, d, device="cuda")
    ops.segment_sum(ops.mat(msg.cuda()), t.csc_colptr, t.csc_eid, ops.mat(gs))
    assert rel(gs, torch.zeros(B, n_send, d).index_add_(1, send, msg)) < 1e-6
    # batch-invariant (stride-0) source
    e1 = torch.randn(M, d, generator=gen).cuda()
    ops.segment_sum(ops.mat(e1.unsqueeze(0).expand(B, -1, -1)), t.csr_rowptr, t.csr_eid, ops.mat(agg))
    want1 = torch.zeros(n_rec, d).index_add_(0, rec, e1.cpu())
    assert rel(agg[2], want1) < 1e-6


@pytest.mark.parametrize("d,B", [(64, 4), (64, 5), (64, 9), (128, 2), (128, 5)])
def test_segment_sum_batch_folded(ops, d, B):
    """B >= 64/(d/4): one wavefront per output row serves all samples (mean + accumulate too);
    long and empty segments included."""
    from neural_lam_amd.graph import EdgeTables

    gen = torch.Generator().manual_seed(d + B)
    n_send, n_rec, M = 30, 50, 700
    send = torch.randint(0, n_send, (M,), generator=gen)
    rec = torch.randint(0, n_rec, (M,), generator=gen)
    rec[rec == 3] = 4
    rec[:60] = 11     # one long segment (> 8 rows in flight per trip)
    t = EdgeTables(send, rec, n_send, n_rec).cuda()
    msg = torch.randn(B, M, d, generator=gen)
    want = torch.zeros(B, n_rec, d).index_add_(1, rec, msg)
    agg = torch.full((B, n_rec, d), float("nan"), device="cuda")
    ops.segment_sum(ops.mat(msg.cuda()), t.csr_rowptr, t.csr_eid, ops.mat(agg))
    assert rel(agg, want) < 1e-6
    deg = torch.zeros(n_rec).index_add_(0, rec, torch.ones(M)).clamp(min=1)
    ops.segment_sum(ops.mat(msg.cuda()), t.csr_rowptr, t.csr_eid, ops.mat(agg), scale=t.inv_deg)
    assert rel(agg, want / deg.view(1, -1, 1)) < 1e-6
    gs = torch.full((B, n_send, d), float("nan"), device="cuda")
    ops.segment_sum(ops.mat(msg.cuda()), t.csc_colptr, t.csc_eid, ops.mat(gs))
    assert rel(gs, torch.zeros(B, n_send, d).index_add_(1, send, msg)) < 1e-6


@pytest.mark.parametrize("case", ["vec4_large", "scalar_small", "unaligned_cols", "column_block_dst"])
def test_slab_reduction_matches_a_plain_sum(case):
    """nlam_reduce_slabs_batch (every weight gradient of a layer goes through it): segments of
    per-workgroup slabs summed into their destinations.  Large float4-addressable layers take
    16-byte lanes, everything else the 4-byte form -- same sums either way, and the same order of
    summation run to run (bit-identical repeats)."""
    from neural_lam_amd import ops

    torch.manual_seed(0)
    dev = "cuda"
    if case == "vec4_large":        # a hidden-128 layer's seven 128 x 128 gradients + biases
        nslabs, d = 160, 128
        stride = d * d + d
        segs = lambda slab, dsts: [(0, d, d, d, dsts[0]), (d * d, 1, d, d, dsts[1])]
        shapes = [(d, d), (d,)]
        reps = 7
    elif case == "scalar_small":    # a hidden-64 layer: below the float4 threshold
        nslabs, d = 37, 64
        stride = d * d + d
        segs = lambda slab, dsts: [(0, d, d, d, dsts[0]), (d * d, 1, d, d, dsts[1])]
        shapes = [(d, d), (d,)]
        reps = 2
    elif case == "unaligned_cols":  # a 64 x 3 embedder gradient beside large aligned ones
        nslabs, d = 64, 256
        stride = d * d + 64 * 32 + 8
        segs = lambda slab, dsts: [(0, d, d, d, dsts[0]), (d * d, 64, 3, 32, dsts[1])]
        shapes = [(d, d), (64, 3)]
        reps = 1
    else:                           # destination = a column block of a wider matrix (dW1[:, d:2d])
        nslabs, d = 96, 128
        stride = d * d
        segs = None
        shapes = None
        reps = 6
    total = []
    with ops.slab_batch():
        for _ in range(reps):
            slab = torch.randn(nslabs * stride, device=dev)
            if case == "column_block_dst":
                wide = torch.zeros(d, 3 * d, device=dev)
                dst = wide[:, d : 2 * d]
                ops.reduce_segments(slab, nslabs, stride, [(0, d, d, d, dst)])
                total.append((slab, [(0, d, d, d, dst)], wide))
            else:
                dsts = [torch.empty(*s, device=dev) for s in shapes]
                ops.reduce_segments(slab, nslabs, stride, segs(slab, dsts))
                total.append((slab, segs(slab, dsts), None))
    torch.cuda.synchronize()
    for slab, sg, wide in total:
        S = slab.view(nslabs, stride).double()
        for off, r, c, ld, dst in sg:
            idx = (off + torch.arange(r, device=dev)[:, None] * ld + torch.arange(c, device=dev)[None, :])
            want = S[:, idx.reshape(-1)].sum(0).reshape(r, c).float()
            got = dst.reshape(r, c)
            assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()), case
        if wide is not None:        # nothing outside the column block was touched
            assert float(wide[:, :d].abs().max()) == 0.0 and float(wide[:, 2 * d :].abs().max()) == 0.0
    # deterministic: the same launch again gives the same bits
    slab, sg, _ = total[0]
    first = [s[4].clone() for s in sg]
    ops.reduce_segments(slab, nslabs, stride, sg)
    torch.cuda.synchronize()
    for a, s in zip(first, sg):
        assert torch.equal(a, s[4])


@pytest.mark.parametrize("d,B,n_out,M", [(128, 4, 300, 2500), (64, 4, 77, 900), (256, 3, 50, 640),
                                         (128, 5, 40, 333), (64, 16, 30, 200), (256, 1, 20, 100)])
def test_segment_sum_bsum_matches_segment_sum_and_a_batch_sum(d, B, n_out, M):
    """nlam_segment_sum_bsum: the per-sample segment sums of nlam_segment_sum AND the batch sum of
    every listed row from one pass (every row listed exactly once, as the out-edge lists of the
    senders list every edge once; segments of 0, 1 and > 8 entries)."""
    from neural_lam_amd import ops
    from neural_lam_amd.ops import mat

    gen = torch.Generator().manual_seed(d + B + M)
    owner = torch.randint(0, n_out, (M,), generator=gen)
    owner[:20] = 3                                   # a long list
    owner[owner == 5] = 6                            # an empty one
    order = torch.argsort(owner, stable=True)
    rowptr = torch.zeros(n_out + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(torch.bincount(owner, minlength=n_out), 0).int()
    src = torch.randn(B, M, d, generator=gen).cuda()
    pos = order.int().cuda()
    out = torch.full((B, n_out, d), float("nan"), device="cuda")
    bsum = torch.full((1, M, d), float("nan"), device="cuda")
    assert ops.segment_sum_bsum_ok(mat(src), mat(out), mat(bsum))
    ops.segment_sum_bsum(mat(src), rowptr.cuda(), pos, mat(out), mat(bsum))
    want = torch.zeros(B, n_out, d, dtype=torch.float64)
    want.index_add_(1, owner, src.cpu().double())
    assert torch.allclose(out.cpu().double(), want, rtol=1e-5, atol=1e-5)
    assert torch.allclose(bsum[0].cpu().double(), src.cpu().double().sum(0), rtol=1e-6, atol=1e-6)
    ref = torch.empty_like(out)
    ops.segment_sum(mat(src), rowptr.cuda(), pos, mat(ref))
    assert torch.equal(out, ref)                     # (same list order per sample)
