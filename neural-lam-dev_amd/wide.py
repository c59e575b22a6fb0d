"""hidden_dim 128 / 256 path of the fused gfx950 kernels (csrc/fused_wide.hip, fused_fs.hip).

Same operators as fused.py -- the make_mlp blocks (reference utils.py:191-214) and one
InteractionNet layer (interaction_net.py:86-131) -- but cut at the Linear boundaries: at d = 128
two split-bf16 weight images (2 x 67.6 KB) plus row tiles exceed the 160 KB LDS of a CU and a
d x d weight-gradient accumulator fills a wave's registers, so every kernel holds ONE weight
matrix and every weight gradient is a streaming pass of its own:

  forward : nlam_lin_fwd   x3   Pe = W1e e, Ps = W1s x_s, Pr = W1r x_r + b1
            nlam_tail_fwd       h = Pe + Ps[send] + Pr[rec]; m = LN(W2 silu(h) + b2);
                                agg = segment-sum(m) over receiver-aligned tiles; e' = e + m
            nlam_lin_fwd   x2 + nlam_tail_fwd   node update  x_r + LN(V2 silu(V1 [x_r | agg] + c1) + c2)
  backward: nlam_tail_bwd (node), nlam_tail_bwd (edges, + receiver-side sums of gh),
            nlam_segment_sum (sender side), nlam_lin_bwd_data (data gradients),
            nlam_wide_outer (every weight / bias gradient), one slab reduction per layer.
The pre-activations h are kept by the forward (the backward neither repeats the first GEMM nor
gathers); memory is sized for 288 GB of HBM.  Arithmetic: NLAM_MFMA=bf16x3 (default, fp32-grade)
or bf16 (plain bf16 products, fp32 accumulate); exact-fp32 mode takes the generic kernels.

hidden_dim 256 runs the same host sequence on the feature-split kernels of csrc/fused_fs.hip
(weights register-stationary, one 64-row tile per 512-thread workgroup) and exists in bf16
arithmetic only -- the reference's `--precision bf16-mixed` (BASELINE configs[4]); the multi-problem
launches of the 128 path become single launches there.
"""
import torch

from . import ops
from ._lib import lib
from .ops import _launch, _p, is_bf16, mat, mat16, stream

WIDE_HIDDEN = (128, 256)


def enabled(hid=128):
    """Both widths run in both MFMA modes: split-bf16 (the default, fp32-grade) and bf16."""
    return int(lib.nlam_mfma_mode()) != 0


def bf16_rows(d):
    """Hidden 256 under NLAM_MFMA=bf16 stores its Linear outputs (and their gradients) as bf16
    rows; every other combination stores fp32 rows."""
    return d == 256 and int(lib.nlam_mfma_mode()) == 2


import os

# Layers with at most this many edge rows (B * M) issue their weight-gradient passes on a second
# stream (ops.WeightGradLane).  Measured on Hi-LAM (MI355X, HIP-graph replay): hidden 256 with one
# launch per weight gradient 48.96 -> 46.09 ms / step, hidden 128 (whose weight gradients are ONE
# multi-problem launch already) 22.64 -> 23.21 ms: splitting that launch costs more than the
# overlap returns.  With multi-problem launches at both widths the lane is off by default.
SIDE_LANE_ROWS = int(os.environ.get("NLAM_SIDE_ROWS", "0"))


# Slab reductions of the hidden-128 layers at the end of an AR step's backward instead of one
# launch per layer (ops.slab_batch(defer=True); NLAM_DEFER_REDUCE=0: per layer, =2: at hidden 256
# as well -- measured there it LOSES, 30.27 -> 30.48 ms: a layer's slabs, up to 67 MB, are read
# back out of the 256 MB Infinity Cache when the reduction follows the launch that wrote them, and
# out of HBM when it comes at the end of the step).  PROXY:
# id(parameter) -> its glue.DeferGrad output for the predict_step in flight; a parameter is handed
# out ONCE per predict_step (a second use of a layer gets the parameters themselves and the
# undeferred path: two gradients into one proxy would be summed by the engine before the flush).
PROXY = {}
# widths whose LARGE layers defer their reductions as well (hidden 256: measured slower, see above;
# hidden 64, through the Python-level MLP / chain functions: measured neutral, not wired)
DEFER_WIDTHS = (128, 256) if os.environ.get("NLAM_DEFER_REDUCE", "1") == "2" else (128,)


def defer_begin(model):
    """Called at the start of predict_step (device path, gradients on, training)."""
    PROXY.clear()
    if os.environ.get("NLAM_DEFER_REDUCE", "1") == "0" or not ops.deferral_allowed():
        return
    widths = WIDE_HIDDEN
    mods = getattr(model, "_wide_defer_mods", None)
    if mods is not None and mods[0] != widths:
        mods = None
    if mods is None:
        from .interaction_net import InteractionNet, SplitMLPs
        from .utils import HipMLP

        mods = []
        for m in model.modules():
            if isinstance(m, InteractionNet):
                if (m.hidden_dim in widths and m.hidden_layers == 1
                        and not isinstance(m.edge_mlp, SplitMLPs) and not isinstance(m.aggr_mlp, SplitMLPs)):
                    mods.append(m)
            elif isinstance(m, HipMLP):
                lin = [l for l in m if isinstance(l, torch.nn.Linear)]
                if len(lin) == 2 and lin[0].weight.shape[0] in widths:
                    mods.append(m)
        mods = (widths, mods)
        model._wide_defer_mods = mods
    seen, params = set(), []
    for m in mods[1]:
        for p in m.parameters():
            if p.requires_grad and p.is_cuda and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
    if not params:
        return
    from . import glue

    for p, q in zip(params, glue.DeferGrad.apply(*params)):
        PROXY[id(p)] = q


def defer_end():
    PROXY.clear()


def _proxies(params):
    """The DeferGrad outputs of `params` (None entries pass through) if every parameter has an
    unused one, else None."""
    if not PROXY or not torch.is_grad_enabled() or any(p is not None and id(p) not in PROXY for p in params):
        return None
    return [PROXY.pop(id(p)) if p is not None else None for p in params]


def _empty(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


def _inter(d, *shape, device):
    """An edge- / node-sized intermediate that is the output of a Linear (projection, pre-activation)
    or the gradient of one (gz): bf16 at hidden 256 in the bf16 mode -- the dtype the reference's
    autocast gives them, and half the vector-memory instructions of kernels that are bound by
    those -- fp32 otherwise."""
    dt = torch.bfloat16 if bf16_rows(d) else torch.float32
    return torch.empty(*shape, dtype=dt, device=device)


def _m(t):
    return mat16(t) if t.dtype == torch.bfloat16 else mat(t)


def _esz(x):
    """bytes per element of a row operand (Mat or tensor) in the algorithmic-byte accounting"""
    if x is None:
        return 0
    dt = x.keep.dtype if hasattr(x, "keep") else x.dtype
    return 2 if dt == torch.bfloat16 else 4


def _aligned(m, width=None):
    width = m.cols if width is None else width
    return (m.ptr % 16 == 0 and m.ld % 4 == 0 and m.bstride % 4 == 0 and width % 4 == 0)


# ------------------------------------------------------------ C-ABI wrappers
class Tiling:
    """Row mode (tiles of 32 consecutive rows) or edge mode (receiver-aligned tiles of an
    EdgeTables over its receiver-sorted positions)."""

    def __init__(self, rows, tables=None):
        self.rows = int(rows)
        self.g = tables
        if tables is None:
            self.args = (None, (self.rows + 31) // 32, self.rows, None, None)
            self.ntiles = (self.rows + 31) // 32
        else:
            # graphs with in-degree > 32: the tiles, segment ids and row pointers of the VIRTUAL
            # receivers (graph.VirtualReceivers); per-receiver outputs then have one row per virtual
            # receiver and are folded back by the caller (fold_virtual)
            t = tables.virtual if tables.virtual is not None else tables
            self.args = (t.tiles.data_ptr(), t.ntiles, self.rows,
                         t.csr_rec.data_ptr(), t.csr_rowptr.data_ptr())
            self.ntiles = t.ntiles


def recv_rows(tables):
    """Rows of a per-receiver output of an edge-mode kernel (virtual receivers if the graph has any)."""
    return tables.virtual.n_rec if tables.virtual is not None else tables.n_rec


def fold_virtual(tables, per_virtual, out, scale=None):
    """Per-virtual-receiver rows -> per-receiver rows (fixed order; scale = the 1/deg of "mean")."""
    ops.segment_sum(mat(per_virtual), tables.virtual.rowptr2, None, out, scale=scale)


def _src(m, idx=None):
    if m is None:
        return (None, 0, 0, None)
    return (m.ptr, m.bstride, m.ld, idx.data_ptr() if idx is not None else None)


def keep_z(B, rows, d, gamma, device):
    """(B, rows, d) buffer for the pre-LayerNorm rows the hidden-256 backward reads back: bf16
    in the bf16 mode, fp32 in the split-bf16 mode."""
    if d != 256 or gamma is None:
        return None
    return torch.empty(B, rows, d, dtype=torch.bfloat16 if bf16_rows(d) else torch.float32,
                       device=device)


def tail_fwd(tl, a, idx_a, b, idx_b, c, idx_c, W2, b2, gamma, beta, h_out, y, idx_y, res, agg,
             inv_deg, B, d, z_keep=None):
    n_out = W2.shape[0]
    if is_bf16(a):   # all-or-nothing: every source and the kept h are bf16 rows
        assert all(is_bf16(t) for t in (b, c) if t is not None)
        assert h_out is None or h_out.dtype == torch.bfloat16
    _launch(
        "nlam_tail_fwd", lib.nlam_tail_fwd,
        tl.args + _src(a, idx_a) + _src(b, idx_b) + _src(c, idx_c)
        + (W2.data_ptr(), W2.stride(0), _p(b2), _p(gamma), _p(beta), n_out,
           h_out.data_ptr() if h_out is not None else None,
           h_out.stride(0) if h_out is not None else 0,
           z_keep.data_ptr() if z_keep is not None else None,
           z_keep.stride(0) if z_keep is not None else 0)
        + ((y.ptr, y.bstride, y.ld) if y is not None else (None, 0, 0))
        + (idx_y.data_ptr() if idx_y is not None else None,)
        + ((res.ptr, res.bstride, res.ld) if res is not None else (None, 0, 0))
        + ((agg.ptr, agg.bstride, agg.ld) if agg is not None else (None, 0, 0))
        + (inv_deg.data_ptr() if inv_deg is not None else None, B, d, int(is_bf16(a)), stream()),
        flops=2.0 * B * tl.rows * d * n_out,
        nbytes=1.0 * B * tl.rows * (d * (_esz(a) + _esz(h_out) + _esz(z_keep))
                                    + 4 * n_out * (y is not None) * (1 + (res is not None)))
        + 4.0 * d * B * (agg.rows if agg is not None else 0),
    )


def tail_fwd_pre_ok(d, B, rows, *mats):
    """The node update's aggregate projection inside its tail launch (nlam_tail_fwd_pre):
    hidden 128, one row tile per wave, fp32 rows; NLAM_TAIL_PRE=0 keeps the two launches."""
    return (os.environ.get("NLAM_TAIL_PRE", "1") != "0" and not bf16_rows(d)
            and bool(lib.nlam_tail_fwd_pre_supported(d, B, rows))
            and all(m is not None and not is_bf16(m) and _aligned(m) for m in mats))


def tail_fwd_pre(rows, a, pre, preW, W2, b2, gamma, beta, h_out, y, res, B, d):
    """h = a + pre preW^T;  y = res + LN(W2 silu(h) + b2);  h kept."""
    _launch(
        "nlam_tail_fwd_pre", lib.nlam_tail_fwd_pre,
        (rows, a.ptr, a.bstride, a.ld, pre.ptr, pre.bstride, pre.ld, preW.data_ptr(), preW.stride(0),
         W2.data_ptr(), W2.stride(0), _p(b2), _p(gamma), _p(beta), h_out.data_ptr(), h_out.stride(0),
         y.ptr, y.bstride, y.ld, res.ptr if res is not None else None,
         res.bstride if res is not None else 0, res.ld if res is not None else 0, B, d, stream()),
        flops=4.0 * B * rows * d * d,
        nbytes=4.0 * B * rows * d * (4 + (res is not None)),
    )


def tail_bwd(tl, h, g1, idx_g1, scale1, g2, idx_g2, W2, b2, gamma, gz_out, gh, idx_gh, gpr, B, d,
             dgamma, dbeta, z_keep=None):
    n_out = W2.shape[0]
    dev = W2.device
    # (bf16 rows: h and, in the LayerNorm form, gz; the narrow head's 32-wide gz stays fp32)
    assert (gz_out.dtype == torch.bfloat16) == (h.dtype == torch.bfloat16 and gamma is not None)
    stride = int(lib.nlam_tail_bwd_slab_stride(n_out))
    nslabs = int(lib.nlam_bwd_grid(B * tl.ntiles))
    slab = torch.empty(nslabs * stride, dtype=torch.float32, device=dev) if gamma is not None else None
    _launch(
        "nlam_tail_bwd", lib.nlam_tail_bwd,
        tl.args + (h.data_ptr(), h.stride(0),
                   z_keep.data_ptr() if z_keep is not None else None,
                   z_keep.stride(0) if z_keep is not None else 0)
        + (g1.ptr, g1.bstride, g1.ld, idx_g1.data_ptr() if idx_g1 is not None else None,
           scale1.data_ptr() if scale1 is not None else None)
        + ((g2.ptr, g2.bstride, g2.ld, idx_g2.data_ptr() if idx_g2 is not None else None)
           if g2 is not None else (None, 0, 0, None))
        + (W2.data_ptr(), W2.stride(0), _p(b2), _p(gamma), n_out, gz_out.data_ptr(),
           gz_out.stride(0), gh.ptr, gh.bstride, gh.ld,
           idx_gh.data_ptr() if idx_gh is not None else None)
        + ((gpr.ptr, gpr.bstride, gpr.ld) if gpr is not None else (None, 0, 0))
        + (slab.data_ptr() if slab is not None else None, stride, B, d,
           int(h.dtype == torch.bfloat16), stream()),
        flops=2.0 * B * tl.rows * d * n_out * (2 if gamma is not None else 1),
        nbytes=1.0 * B * tl.rows * (d * (_esz(h) + 4 + _esz(z_keep)) + n_out * (4 + _esz(gz_out))
                                    + (4 * n_out if g2 is not None else 0)),
    )
    if gamma is not None:
        no = (n_out + 31) // 32 * 32
        ops.reduce_segments(slab, nslabs, stride,
                            [(0, 1, n_out, n_out, dgamma), (no, 1, n_out, n_out, dbeta)])


def lin_bwd_data(gy, W, gx, gx_add=None):
    """gx = gy W (+ gx_add); W: (d, d) view (any row pitch)."""
    n_out, k_in = W.shape
    _launch(
        "nlam_lin_bwd_data", lib.nlam_lin_bwd_data,
        (gy.ptr, gy.bstride, gy.ld, n_out, W.data_ptr(), W.stride(0), k_in, gx.ptr, gx.bstride,
         gx.ld) + ((gx_add.ptr, gx_add.bstride, gx_add.ld) if gx_add is not None else (None, 0, 0))
        + (gx.B, gx.rows, stream()),
        flops=2.0 * gx.B * gx.rows * n_out * k_in,
        nbytes=4.0 * gx.B * gx.rows * (n_out + k_in * (2 if gx_add is not None else 1)),
    )


def outer(g, x, dW, db, silu_x=False, rows_out=None):
    """dW (rows_out x d view) = sum_rows g^T f(x); db = colsum(g).  g: (B, rows, d | 32)."""
    ng, nx = g.cols, x.cols
    nxp = (nx + 31) // 32 * 32
    B, rows = g.B, g.rows
    stride = ng * nxp + ng
    nslabs = int(lib.nlam_bwd_grid(B * ((rows + 31) // 32)))
    slab = torch.empty(nslabs * stride, dtype=torch.float32, device=dW.device)
    _launch(
        "nlam_wide_outer", lib.nlam_wide_outer,
        (g.ptr, g.bstride, g.ld, ng, x.ptr, x.bstride, x.ld, nx, int(silu_x), slab.data_ptr(),
         stride, B, rows, int(is_bf16(g)) | 2 * int(is_bf16(x)), stream()),
        flops=2.0 * B * rows * ng * nx, nbytes=1.0 * B * rows * (ng * _esz(g) + nx * _esz(x)),
    )
    r = ng if rows_out is None else rows_out
    ops.reduce_segments(slab, nslabs, stride,
                        [(0, r, nx, nxp, dW), (ng * nxp, 1, r, r, db)])


def _arr(ctype, vals):
    return (ctype * len(vals))(*vals)


def _parr(vals):
    import ctypes
    return (ctypes.c_void_p * len(vals))(*vals)


def lin_fwd_multi(problems):
    """[(x Mat, W (d, d) view, bias or None, out Mat)] -> one launch (aligned d-wide rows)."""
    import ctypes
    d = problems[0][1].shape[0]
    I64 = ctypes.c_int64
    n = len(problems)
    _launch(
        "nlam_lin_fwd_multi", lib.nlam_lin_fwd_multi,
        (n, d, _parr([x.ptr for x, _, _, _ in problems]), _arr(I64, [x.bstride for x, _, _, _ in problems]),
         _arr(I64, [x.ld for x, _, _, _ in problems]), _parr([W.data_ptr() for _, W, _, _ in problems]),
         _arr(I64, [W.stride(0) for _, W, _, _ in problems]),
         _parr([_p(b) for _, _, b, _ in problems]), _parr([o.ptr for _, _, _, o in problems]),
         _arr(I64, [o.bstride for _, _, _, o in problems]), _arr(I64, [o.ld for _, _, _, o in problems]),
         _arr(I64, [o.B for _, _, _, o in problems]), _arr(I64, [o.rows for _, _, _, o in problems]),
         sum(int(is_bf16(o)) << k for k, (_, _, _, o) in enumerate(problems)), stream()),
        flops=sum(2.0 * o.B * o.rows * d * d for _, _, _, o in problems),
        nbytes=sum(1.0 * o.B * o.rows * d * (4 + _esz(o)) for _, _, _, o in problems),
    )


def lin_bwd_data_multi(problems):
    """[(gy Mat, W (d, d) view, gx Mat, gx_add Mat or None)] -> one launch."""
    import ctypes
    d = problems[0][1].shape[0]
    I64 = ctypes.c_int64
    n = len(problems)
    _launch(
        "nlam_lin_bwd_data_multi", lib.nlam_lin_bwd_data_multi,
        (n, d, _parr([g.ptr for g, _, _, _ in problems]), _arr(I64, [g.bstride for g, _, _, _ in problems]),
         _arr(I64, [g.ld for g, _, _, _ in problems]), _parr([W.data_ptr() for _, W, _, _ in problems]),
         _arr(I64, [W.stride(0) for _, W, _, _ in problems]),
         _parr([x.ptr for _, _, x, _ in problems]), _arr(I64, [x.bstride for _, _, x, _ in problems]),
         _arr(I64, [x.ld for _, _, x, _ in problems]),
         _parr([a.ptr if a is not None else None for _, _, _, a in problems]),
         _arr(I64, [a.bstride if a is not None else 0 for _, _, _, a in problems]),
         _arr(I64, [a.ld if a is not None else 0 for _, _, _, a in problems]),
         _arr(I64, [x.B for _, _, x, _ in problems]), _arr(I64, [x.rows for _, _, x, _ in problems]),
         stream()),
        flops=sum(2.0 * x.B * x.rows * d * d for _, _, x, _ in problems),
        nbytes=sum(4.0 * x.B * x.rows * d * (2 + (a is not None)) for _, _, x, a in problems),
    )


# slabs of a weight-gradient launch: in all / average floor per problem (tunables; see outer_multi)
_OUTER_BUDGET = int(os.environ.get("NLAM_OUTER_BUDGET", "256"))
_OUTER_FLOOR = int(os.environ.get("NLAM_OUTER_FLOOR", "64"))
_OUTER_FLOOR_MERGED = int(os.environ.get("NLAM_OUTER_FLOOR_MERGED", "16"))
_OUTER_MIN_TILES = int(os.environ.get("NLAM_OUTER_MIN_TILES", "8"))   # 32-row tiles per slab, at least


OUTER_MAXP = 24   # csrc/fused_common.h NLAM_WIDE_MAXP_OUTER

# Weight-gradient problems handed on to the end of the AR step's backward (glue.DeferGrad ->
# flush_deferred_outers): the products feed nothing before the optimizer; a small layer's launch is
# all latency (20 - 60 us for a few MB), and merged 24 to a launch -- workgroups shared out by work --
# the problems of a whole step run side by side; each merged launch is followed by the reduction of
# ITS slabs while the Infinity Cache still has them.  Measured (same box, layers of at most 60 k /
# 250 k / any number of edge rows deferred): Hi-LAM-128 19.20 -> 18.65 / 18.14 / 18.06 ms, Hi-LAM-256
# (bf16) 30.36 -> 29.89 / 29.66 / 29.64 ms.  NLAM_DEFER_OUTER_ROWS caps the size of a layer that
# defers (0: none); the problems hold their operands alive until the flush (a few GB at B = 4).
DEFER_OUTER_ROWS = int(os.environ.get("NLAM_DEFER_OUTER_ROWS", str(1 << 40)))
SMALL_LAYER_ROWS = 60000   # (a hidden-256 layer up to this size defers its remaining slab reductions too)
_DEFERRED_OUTERS = []


def defer_outers(problems):
    _DEFERRED_OUTERS.extend(problems)
    ops.on_flush(flush_deferred_outers)


def flush_deferred_outers():
    if not _DEFERRED_OUTERS:
        return
    problems = list(_DEFERRED_OUTERS)
    del _DEFERRED_OUTERS[:]
    by_kind = {}   # (d, 0) = d x d problems; (d, 32 | 64) = narrow x, by slab row width
    for pr in problems:
        d, nx = pr[0].cols, pr[1].cols
        by_kind.setdefault((d, 0 if nx == d else (32 if nx <= 32 else 64)), []).append(pr)
    with ops.tag("deferred"):
        for (d, nxp), prs in by_kind.items():
            for i in range(0, len(prs), OUTER_MAXP):
                # (each merged launch's slabs are reduced at once, while the Infinity Cache has them)
                with ops.slab_batch():
                    (outer_multi if nxp == 0 else outer_multi_nx)(prs[i : i + OUTER_MAXP])


def outer_multi_nx(problems):
    """[(g Mat (.., d), x Mat (.., nx <= 64), dW (d, nx) view, db, _)] -> one launch + its slab
    reduction (narrow first Linears; slab rows of 32 ceil(max nx / 32) columns)."""
    import ctypes
    d = problems[0][0].cols
    I64, I32 = ctypes.c_int64, ctypes.c_int32
    n = len(problems)
    dev = problems[0][2].device
    nxp = 32 if max(x.cols for _, x, _, _, _ in problems) <= 32 else 64
    stride = d * nxp + d
    ns = [int(lib.nlam_bwd_grid(g.B * ((g.rows + 31) // 32))) for g, _, _, _, _ in problems]
    slabs = [torch.empty(nsl * stride, dtype=torch.float32, device=dev) for nsl in ns]
    _launch(
        "nlam_wide_outer_multi", lib.nlam_wide_outer_multi_nx,
        (n, d, _parr([g.ptr for g, _, _, _, _ in problems]),
         _arr(I64, [g.bstride for g, _, _, _, _ in problems]),
         _arr(I64, [g.ld for g, _, _, _, _ in problems]),
         _parr([x.ptr for _, x, _, _, _ in problems]),
         _arr(I64, [x.bstride for _, x, _, _, _ in problems]),
         _arr(I64, [x.ld for _, x, _, _, _ in problems]),
         _arr(I32, [x.cols for _, x, _, _, _ in problems]),
         _parr([sl.data_ptr() for sl in slabs]), _arr(I64, [stride] * n),
         _arr(I64, [g.B for g, _, _, _, _ in problems]),
         _arr(I64, [g.rows for g, _, _, _, _ in problems]), _arr(I32, ns), stream()),
        flops=sum(2.0 * g.B * g.rows * d * x.cols for g, x, _, _, _ in problems),
        nbytes=sum(1.0 * g.B * g.rows * (d * _esz(g) + x.cols * _esz(x)) for g, x, _, _, _ in problems),
    )
    for (g, x, dW, db, _), sl, nsl in zip(problems, slabs, ns):
        ops.reduce_segments(sl, nsl, stride, [(0, d, x.cols, nxp, dW), (d * nxp, 1, d, d, db)])


def outer_multi(problems):
    """[(g Mat (.., d), x Mat (.., d), dW view, db, silu_x)] -> one launch + the layer's slab
    reduction (all d x d; at most OUTER_MAXP problems)."""
    import ctypes
    d = problems[0][0].cols
    I64, I32 = ctypes.c_int64, ctypes.c_int32
    n = len(problems)
    dev = problems[0][2].device
    stride = d * d + d
    # slabs (= workgroups) per problem: proportional to the row counts, one round of the device
    # (256; one workgroup is resident per CU) or 64 per problem in all -- every slab is d*d floats
    # written and read back.  Measured on Hi-LAM-128 / -256: 512 in all 20.22 / 32.97 ms, 256:
    # 19.89 / 32.19; an average below 64 per problem 20.7 / 33.8
    tiles = [g.B * ((g.rows + 31) // 32) for g, _, _, _, _ in problems]
    total = max(1, sum(tiles))
    # (merged end-of-step launches, n up to 24: an average of 16 slabs per problem -- measured on
    # Hi-LAM-256 / -128 for 64 / 32 / 16 / 8: 29.37 / 29.28 / 29.19 / 30.5 ms and 17.77 / 17.70 / 17.71 / 18.1)
    budget = max(_OUTER_BUDGET, (_OUTER_FLOOR if n <= 8 else _OUTER_FLOOR_MERGED) * n)
    slabs, ns = [], []
    for t in tiles:
        nsl = max(1, min(int(lib.nlam_bwd_grid(t)), -(-budget * t // total), -(-t // _OUTER_MIN_TILES)))
        ns.append(nsl)
        slabs.append(torch.empty(nsl * stride, dtype=torch.float32, device=dev))
    _launch(
        "nlam_wide_outer_multi", lib.nlam_wide_outer_multi,
        (n, d, _parr([g.ptr for g, _, _, _, _ in problems]),
         _arr(I64, [g.bstride for g, _, _, _, _ in problems]),
         _arr(I64, [g.ld for g, _, _, _, _ in problems]),
         _parr([x.ptr for _, x, _, _, _ in problems]),
         _arr(I64, [x.bstride for _, x, _, _, _ in problems]),
         _arr(I64, [x.ld for _, x, _, _, _ in problems]),
         _arr(I32, [int(sx) for _, _, _, _, sx in problems]),
         _parr([sl.data_ptr() for sl in slabs]), _arr(I64, [stride] * n),
         _arr(I64, [g.B for g, _, _, _, _ in problems]),
         _arr(I64, [g.rows for g, _, _, _, _ in problems]), _arr(I32, ns),
         _arr(I32, [int(is_bf16(g)) | 2 * int(is_bf16(x)) for g, x, _, _, _ in problems]),
         stream()),
        flops=sum(2.0 * g.B * g.rows * d * d for g, _, _, _, _ in problems),
        nbytes=sum(1.0 * g.B * g.rows * d * (_esz(g) + _esz(x)) for g, x, _, _, _ in problems),
    )
    for (g, x, dW, db, sx), sl, nsl in zip(problems, slabs, ns):
        ops.reduce_segments(sl, nsl, stride, [(0, d, d, d, dW), (d * d, 1, d, d, db)])


def _first_linear(x, W, b, out):
    """out = x W^T + b: the split-bf16 128 -> 128 projection, or the generic-K kernel."""
    ops.fused_lin_fwd(x, W, b, None, None, out)


def _first_linear_bwd(x, ga, W, need_gx, gx_add, dW, db, dev, deferred=False):
    """Backward of out = x W^T + b given ga = dL/dout: returns gx (or None), fills dW / db
    (deferred: the d x d weight-gradient product joins the step's merged launches)."""
    k_in = W.shape[1]
    gx = None
    wide_ok = k_in == ga.cols and _aligned(x)
    if need_gx:
        gx = _empty(ga.B, ga.rows, k_in, device=dev)
        if wide_ok:
            lin_bwd_data(ga, W, mat(gx), gx_add)
        else:
            ops.linear_bwd_data(ga, W, mat(gx))
            if gx_add is not None:
                ops.add_rows(mat(gx), gx_add, mat(gx))
    if deferred and (wide_ok or k_in <= 64) and not is_bf16(ga) and not is_bf16(x) \
            and ga.B * ga.rows <= DEFER_OUTER_ROWS:
        defer_outers([(ga, x, dW, db, False)])   # (d x d, or a narrow first Linear: its own merged launches)
    elif wide_ok or k_in <= 64:
        outer(ga, x, dW, db)    # (narrow / unaligned static features: scalar staging, K padded)
    else:
        ops.linear_bwd_weight(ga, x, dW, db)   # generic split-K GEMM
    return gx


# ----------------------------------------------------------------------- MLP
def mlp_eligible(seq, x, res=None):
    from .fused import FORCE_GENERIC, _mlp_parts

    if FORCE_GENERIC or not x.is_cuda or x.dtype != torch.float32:
        return False
    if res is not None and res is not x and res.shape[:-1] != x.shape[:-1]:
        return False
    lin, ln = _mlp_parts(seq)
    if len(lin) != 2:
        return False
    hid, k_in = lin[0].weight.shape
    n_out = lin[1].weight.shape[0]
    if hid not in WIDE_HIDDEN or not enabled(hid) or lin[1].weight.shape[1] != hid or k_in > hid:
        return False
    if ln is not None:
        return n_out == hid
    return n_out <= 32


class WideMLPFunction(torch.autograd.Function):
    """y = [res +] [LN](W2 silu(W1 x + b1) + b2), x: (..., rows, k_in), hidden 128 / 256."""

    @staticmethod
    def forward(ctx, x, res, W1, b1, W2, b2, gamma, beta, give=None, deferred=False):
        ctx.tag = ops._TAG[-1] if ops._TAG else "mlp"
        ctx.give = give   # glue.GradSlot: leave the input gradient there as well (glue.Tee)
        ctx.deferred = deferred   # the parameters are glue.DeferGrad outputs (ops.slab_batch)
        dev = x.device
        hid, n_out = W1.shape[0], W2.shape[0]
        xm = mat(x.detach())
        res_is_x = res is x
        rm = xm if res_is_x else (mat(res.detach()) if res is not None else None)
        B, rows = xm.B, xm.rows
        h = _inter(hid, B, rows, hid, device=dev)
        _first_linear(xm, W1, b1, _m(h))
        out = _empty(B, rows, n_out, device=dev)
        zk = keep_z(B, rows, hid, gamma, dev)
        tail_fwd(Tiling(rows), _m(h), None, None, None, None, None, W2, b2, gamma, beta, None,
                 mat(out), None, rm, None, None, B, hid, zk)
        ctx.save_for_backward(W1, b1, W2, b2, gamma, h, zk)
        ctx.xm, ctx.x_shape = xm, x.shape
        ctx.res_mode = 0 if res is None else (1 if res_is_x else 2)
        ctx.B = B
        return out.reshape(*x.shape[:-1], n_out)

    @staticmethod
    def backward(ctx, gy):
        W1, b1, W2, b2, gamma, h, zk = ctx.saved_tensors
        hid, k_in = W1.shape
        n_out = W2.shape[0]
        xm, B = ctx.xm, ctx.B
        dev = W1.device
        rows = xm.rows
        gy = gy.contiguous().reshape(B, rows, n_out)
        gym = mat(gy)
        no = (n_out + 31) // 32 * 32
        has_ln = gamma is not None
        dW1, db1 = torch.empty_like(W1), _empty(hid, device=dev)
        dW2, db2 = torch.empty_like(W2), _empty(n_out, device=dev)
        dg = _empty(n_out, device=dev) if has_ln else None
        dbt = _empty(n_out, device=dev) if has_ln else None
        need_gx = ctx.needs_input_grad[0]
        defer_red = ctx.deferred and (hid in DEFER_WIDTHS or B * rows <= SMALL_LAYER_ROWS)
        with ops.tag(ctx.tag), ops.slab_batch(defer=defer_red):
            gz = torch.empty(B, rows, no, device=dev,
                             dtype=h.dtype if has_ln else torch.float32)   # (narrow heads: fp32 gz)
            ga = _empty(B, rows, hid, device=dev)
            tail_bwd(Tiling(rows), h, gym, None, None, None, None, W2, b2, gamma, gz, mat(ga),
                     None, None, B, hid, dg, dbt, zk)
            if ctx.deferred and has_ln and n_out == hid and B * rows <= DEFER_OUTER_ROWS:
                defer_outers([(_m(gz), _m(h), dW2, db2, True)])   # (a small embedder: merged later)
            else:
                outer(_m(gz), _m(h), dW2, db2, silu_x=True, rows_out=n_out)
            gx_add = gym if (ctx.res_mode == 1 and need_gx) else None
            gx = _first_linear_bwd(xm, mat(ga), W1, need_gx, gx_add, dW1, db1, dev, ctx.deferred)
        gres = gy if ctx.res_mode == 2 else None
        if gx is not None:
            gx = gx.reshape(ctx.x_shape)
            if ctx.give is not None:
                ctx.give.put(gx)
        return (gx, gres, dW1, db1, dW2, db2, dg, dbt, None, None)


# ---- the static-feature embedders of a model as multi-problem launches (hidden 128)
TAIL_MAXP = 8   # csrc/fused_wide.hip


def embedder_multi_ok(seq, x):
    """A static-feature embedder the multi-problem tails take: (rows, k) input, Linear(k, 128),
    SiLU, Linear(128, 128), LayerNorm; fp32 rows."""
    from .fused import FORCE_GENERIC, _mlp_parts

    if FORCE_GENERIC or os.environ.get("NLAM_EMBED_MULTI", "1") == "0" or not x.is_cuda \
            or x.dtype != torch.float32 or x.dim() != 2 or x.requires_grad:
        return False
    lin, ln = _mlp_parts(seq)
    if len(lin) != 2 or ln is None:
        return False
    hid = lin[0].weight.shape[0]
    return (hid == 128 and enabled(hid) and not bf16_rows(hid) and tuple(lin[1].weight.shape) == (hid, hid)
            and lin[0].weight.shape[1] <= hid and x.shape[0] >= 1)


def _shares(rows):
    import ctypes
    n = len(rows)
    out = (ctypes.c_int32 * n)()
    ops.check(lib.nlam_mlp_tail_multi_shares(n, _arr(ctypes.c_int64, [1] * n),
                                             _arr(ctypes.c_int64, rows), out),
              "nlam_mlp_tail_multi_shares")
    return [int(v) for v in out]


class WideMultiMLPFunction(torch.autograd.Function):
    """n static-feature embedders y_k = LN(W2_k silu(W1_k x_k + b1_k) + b2_k): the first Linears one
    launch each (K <= 4 mostly), the tails in launches of up to TAIL_MAXP problems, both ways."""

    @staticmethod
    def forward(ctx, n, deferred, *args):
        import ctypes
        xs, params = args[:n], args[n:]
        P = [params[6 * k : 6 * k + 6] for k in range(n)]
        dev = xs[0].device
        d = P[0][2].shape[0]
        xms = [mat(x.detach()) for x in xs]
        hs = [_empty(1, xm.rows, d, device=dev) for xm in xms]
        outs = [_empty(xm.rows, d, device=dev) for xm in xms]
        I64 = ctypes.c_int64
        with ops.tag("static_embedders"):
            for k in range(n):
                _first_linear(xms[k], P[k][0], P[k][1], mat(hs[k]))
            for i in range(0, n, TAIL_MAXP):
                ks = range(i, min(n, i + TAIL_MAXP))
                _launch(
                    "nlam_tail_fwd_multi", lib.nlam_mlp_tail_fwd_multi,
                    (len(ks), d, _parr([hs[k].data_ptr() for k in ks]),
                     _parr([P[k][2].data_ptr() for k in ks]), _arr(I64, [P[k][2].stride(0) for k in ks]),
                     _parr([P[k][3].data_ptr() for k in ks]), _parr([P[k][4].data_ptr() for k in ks]),
                     _parr([P[k][5].data_ptr() for k in ks]), _parr([outs[k].data_ptr() for k in ks]),
                     _arr(I64, [1] * len(ks)), _arr(I64, [xms[k].rows for k in ks]), stream()),
                    flops=sum(2.0 * xms[k].rows * d * d for k in ks),
                    nbytes=sum(8.0 * xms[k].rows * d for k in ks))
        ctx.save_for_backward(*params, *hs)
        ctx.set_materialize_grads(False)
        ctx.n, ctx.xms, ctx.deferred = n, xms, deferred
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        import ctypes
        n = ctx.n
        saved = ctx.saved_tensors
        params, hs = saved[: 6 * n], saved[6 * n :]
        P = [params[6 * k : 6 * k + 6] for k in range(n)]
        dev = params[0].device
        d = P[0][2].shape[0]
        I64, I32 = ctypes.c_int64, ctypes.c_int32
        stride = int(lib.nlam_tail_bwd_slab_stride(d))
        grads = []
        with ops.tag("static_embedders"), ops.slab_batch(defer=ctx.deferred):
            gy_c, gzs, gas, slabs, shares = [], [], [], [], []
            for k in range(n):
                rows = ctx.xms[k].rows
                gy = gys[k]
                gy = torch.zeros(rows, d, dtype=torch.float32, device=dev) if gy is None else gy.contiguous()
                gy_c.append(gy)
                gzs.append(_empty(1, rows, d, device=dev))
                gas.append(_empty(1, rows, d, device=dev))
            for i in range(0, n, TAIL_MAXP):
                ks = list(range(i, min(n, i + TAIL_MAXP)))
                sh = _shares([ctx.xms[k].rows for k in ks])
                sl = [torch.empty(s_ * stride, dtype=torch.float32, device=dev) for s_ in sh]
                shares += sh
                slabs += sl
                _launch(
                    "nlam_tail_bwd_multi", lib.nlam_mlp_tail_bwd_multi,
                    (len(ks), d, _parr([hs[k].data_ptr() for k in ks]), _parr([gy_c[k].data_ptr() for k in ks]),
                     _parr([P[k][2].data_ptr() for k in ks]), _arr(I64, [P[k][2].stride(0) for k in ks]),
                     _parr([P[k][3].data_ptr() for k in ks]), _parr([P[k][4].data_ptr() for k in ks]),
                     _parr([gzs[k].data_ptr() for k in ks]), _parr([gas[k].data_ptr() for k in ks]),
                     _parr([t.data_ptr() for t in sl]), _arr(I32, sh), _arr(I64, [1] * len(ks)),
                     _arr(I64, [ctx.xms[k].rows for k in ks]), stream()),
                    flops=sum(4.0 * ctx.xms[k].rows * d * d for k in ks),
                    nbytes=sum(16.0 * ctx.xms[k].rows * d for k in ks))
            for k in range(n):
                W1, b1, W2, b2, gam, bet = P[k]
                dW1, db1 = torch.empty_like(W1), _empty(W1.shape[0], device=dev)
                dW2, db2 = torch.empty_like(W2), _empty(d, device=dev)
                dg, dbt = _empty(d, device=dev), _empty(d, device=dev)
                ops.reduce_segments(slabs[k], shares[k], stride, [(0, 1, d, d, dg), (d, 1, d, d, dbt)])
                if ctx.deferred:
                    defer_outers([(mat(gzs[k]), mat(hs[k]), dW2, db2, True)])
                else:
                    outer(mat(gzs[k]), mat(hs[k]), dW2, db2, silu_x=True, rows_out=d)
                _first_linear_bwd(ctx.xms[k], mat(gas[k]), W1, False, None, dW1, db1, dev, ctx.deferred)
                grads += [dW1, db1, dW2, db2, dg, dbt]
        return (None, None, *([None] * n), *grads)


def embed_many(items):
    """items: [(key, HipMLP, x)] all embedder_multi_ok -> {key: embedding}."""
    from .fused import _mlp_parts

    params = []
    for _, m, _x in items:
        lin, ln = _mlp_parts(m)
        params += [lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, ln.weight, ln.bias]
    prox = _proxies(params)
    res = WideMultiMLPFunction.apply(len(items), prox is not None, *[x for _, _, x in items],
                                     *(prox if prox is not None else params))
    return {k: r for (k, _, _), r in zip(items, res)}


def apply_mlp(seq, x, res=None):
    from .fused import _mlp_parts

    from .fused import _sink

    lin, ln = _mlp_parts(seq)
    params = [lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias,
              ln.weight if ln is not None else None, ln.bias if ln is not None else None]
    prox = _proxies(params)
    return WideMLPFunction.apply(
        x, res, *(prox if prox is not None else params),
        _sink(x, "give") if (res is None or res is x) else None, prox is not None)


# ---------------------------------------------------------- InteractionNet
def inet_eligible(net, send_rep, rec_rep, edge_rep):
    from .fused import FORCE_GENERIC
    from .interaction_net import SplitMLPs

    if FORCE_GENERIC or not edge_rep.is_cuda or edge_rep.dtype != torch.float32:
        return False
    if isinstance(net.edge_mlp, SplitMLPs) or isinstance(net.aggr_mlp, SplitMLPs):
        return False
    if net.hidden_layers != 1 or net.input_dim != net.hidden_dim:
        return False
    if net.hidden_dim not in WIDE_HIDDEN or not enabled(net.hidden_dim):
        return False
    if send_rep.dim() != 3 or rec_rep.dim() != 3 or edge_rep.dim() != 3:
        return False
    for t in (send_rep, rec_rep, edge_rep):
        if not _aligned(mat(t.detach())):
            return False
    return net.tables.ntiles > 0 or net.tables.virtual is not None


class WideInteractionNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, send_rep, rec_rep, edge_rep, same, g, update_edges, mean,
                W1, b1, W2, b2, gam, bet, V1, c1, V2, c2, gam2, bet2, take=None, give=None,
                deferred=False):
        # glue.GradSlot side channels of a glue.Tee on send_rep / rec_rep (see fused.py)
        ctx.take, ctx.give = take, give
        ctx.deferred = deferred   # the parameters are glue.DeferGrad outputs (ops.slab_batch)
        with ops.tag(g.tag):
            dev = edge_rep.device
            d = W2.shape[0]
            B = max(send_rep.shape[0], rec_rep.shape[0], edge_rep.shape[0])
            N_s, N_r, M = send_rep.shape[1], rec_rep.shape[1], edge_rep.shape[1]
            sm, rm, em = mat(send_rep.detach()), mat(rec_rep.detach()), mat(edge_rep.detach())
            W1e, W1s, W1r = W1[:, :d], W1[:, d : 2 * d], W1[:, 2 * d :]
            Ps = _inter(d, sm.B, N_s, d, device=dev)
            Pr = _inter(d, rm.B, N_r, d, device=dev)
            Pe = _inter(d, em.B, M, d, device=dev)
            hn1 = _inter(d, rm.B, N_r, d, device=dev)
            # the four projections that only need the layer inputs: ONE launch
            lin_fwd_multi([(sm, W1s, None, _m(Ps)), (rm, W1r, b1, _m(Pr)),
                           (em, W1e, None, _m(Pe)), (rm, V1[:, :d], c1, _m(hn1))])
            h_e = _inter(d, B, M, d, device=dev)
            agg = _empty(B, N_r, d, device=dev)
            e_out = _empty(B, M, d, device=dev) if update_edges else None
            tl = Tiling(M, g)
            z_e = keep_z(B, M, d, gam, dev)
            virt = g.virtual is not None
            agg_k = _empty(B, recv_rows(g), d, device=dev) if virt else agg
            tail_fwd(tl, _m(Pe), g.csr_eid, _m(Ps), g.csr_send, _m(Pr), g.csr_rec, W2, b2, gam,
                     bet, h_e, mat(e_out) if update_edges else None,
                     g.csr_eid if update_edges else None, em if update_edges else None,
                     mat(agg_k), (g.inv_deg if mean else None) if not virt else None, B, d, z_e)
            if virt:
                fold_virtual(g, agg_k, mat(agg), g.inv_deg if mean else None)
            del Pe, Ps, Pr
            # node update x_r + LN(V2 silu(V1 [x_r | agg] + c1) + c2)
            h_n = _inter(d, B, N_r, d, device=dev)
            rec_out = _empty(B, N_r, d, device=dev)
            z_n = keep_z(B, N_r, d, gam2, dev)
            if z_n is None and tail_fwd_pre_ok(d, B, N_r, _m(hn1), mat(agg), mat(rec_out), rm):
                # (mesh-sized receiver sets: the aggregate's projection rides inside the tail)
                tail_fwd_pre(N_r, _m(hn1), mat(agg), V1[:, d:], V2, c2, gam2, bet2, h_n,
                             mat(rec_out), rm, B, d)
            else:
                hn2 = _inter(d, B, N_r, d, device=dev)
                _first_linear(mat(agg), V1[:, d:], None, _m(hn2))
                tail_fwd(Tiling(N_r), _m(hn1), None, _m(hn2), None, None, None, V2, c2, gam2, bet2,
                         h_n, mat(rec_out), None, rm, None, None, B, d, z_n)
            ctx.save_for_backward(W1, b1, W2, b2, gam, V1, c1, V2, c2, gam2, h_e, h_n, agg, z_e, z_n)
            ctx.set_materialize_grads(False)
            ctx.g, ctx.same, ctx.update_edges, ctx.mean = g, same, update_edges, mean
            ctx.mats = (sm, rm, em)
            ctx.dims = (B, N_s, N_r, M, d)
        if update_edges:
            return rec_out, e_out
        return rec_out

    @staticmethod
    def backward(ctx, g_rec_out, g_edge_out=None):
        # (reductions at the end of the step: hidden 128 always, hidden 256 for small layers only)
        defer_red = ctx.deferred and (ctx.dims[4] in DEFER_WIDTHS
                                      or ctx.dims[0] * ctx.dims[3] <= SMALL_LAYER_ROWS)
        with ops.tag(ctx.g.tag), ops.slab_batch(defer=defer_red):
            W1, b1, W2, b2, gam, V1, c1, V2, c2, gam2, h_e, h_n, agg, z_e, z_n = ctx.saved_tensors
            g = ctx.g
            sm, rm, em = ctx.mats
            B, N_s, N_r, M, d = ctx.dims
            dev = W1.device
            W1e, W1s, W1r = W1[:, :d], W1[:, d : 2 * d], W1[:, 2 * d :]
            same = ctx.same
            if g_rec_out is None:
                g_rec_out = torch.zeros(B, N_r, d, dtype=torch.float32, device=dev)
            g_rec_out = g_rec_out.contiguous()
            dW1 = _empty(d, 3 * d, device=dev)
            db1 = _empty(d, device=dev)
            dW2, db2 = torch.empty_like(W2), _empty(d, device=dev)
            dgam, dbet = _empty(d, device=dev), _empty(d, device=dev)
            dV1, dc1 = torch.empty_like(V1), _empty(d, device=dev)
            dV2, dc2 = torch.empty_like(V2), _empty(d, device=dev)
            dg2, db2n = _empty(d, device=dev), _empty(d, device=dev)
            lane = ops.WeightGradLane(B * M <= SIDE_LANE_ROWS, dev)   # (default: off, see above)
            # 1. node update backward
            gz_n = _inter(d, B, N_r, d, device=dev)
            ga_n = _empty(B, N_r, d, device=dev)
            tail_bwd(Tiling(N_r), h_n, mat(g_rec_out), None, None, None, None, V2, c2, gam2, gz_n,
                     mat(ga_n), None, None, B, d, dg2, db2n, z_n)
            g_rec = _empty(B, N_r, d, device=dev)       # node-update part + residual
            g_agg = _empty(B, N_r, d, device=dev)
            lin_bwd_data_multi([(mat(ga_n), V1[:, :d], mat(g_rec), mat(g_rec_out)),
                                (mat(ga_n), V1[:, d:], mat(g_agg), None)])
            dummy = [_empty(d, device=dev) for _ in range(4)]
            outers = [(_m(gz_n), _m(h_n), dV2, dc2, True), (mat(ga_n), rm, dV1[:, :d], dc1, False),
                      (mat(ga_n), mat(agg), dV1[:, d:], dummy[0], False)]
            if lane.enabled:
                with lane:
                    outer_multi(outers)
                outers = []
            if rm.B == 1 and B > 1:
                t3 = _empty(1, N_r, d, device=dev)
                ops.sum_batch(g_rec, t3)
                g_rec = t3
            # 2. edge backward
            gz_e = _inter(d, B, M, d, device=dev)
            gh = _empty(B, M, d, device=dev)
            gPr = _empty(B, N_r, d, device=dev)
            geo = None
            if ctx.update_edges and g_edge_out is not None:
                geo = mat(g_edge_out.contiguous())
            virt = g.virtual is not None
            gPr_k = _empty(B, recv_rows(g), d, device=dev) if virt else gPr
            tail_bwd(Tiling(M, g), h_e, mat(g_agg), g.csr_rec, g.inv_deg if ctx.mean else None,
                     geo, g.csr_eid if geo is not None else None, W2, b2, gam, gz_e, mat(gh),
                     g.csr_eid, mat(gPr_k), B, d, dgam, dbet, z_e)
            if virt:
                fold_virtual(g, gPr_k, mat(gPr))
            outers.append((_m(gz_e), _m(h_e), dW2, db2, True))
            # 3. sender-side reduction of gh (edge order; sender lists of edge ids)
            gPs = _empty(B, N_s, d, device=dev)
            if N_s > g.n_send:   # (rows of nodes past the last sender: no edge, zero gradient)
                gPs[:, g.n_send :].zero_()
            # (batch-invariant edge term: its gradient -- the sum of gh over the batch -- comes out of
            # the same walk over the sender lists, every edge being in exactly one of them)
            t6 = None
            if not ctx.update_edges and em.B == 1 and B > 1:
                t6 = _empty(1, M, d, device=dev)
                if not ops.segment_sum_bsum_ok(mat(gh), mat(gPs[:, : g.n_send]), mat(t6)):
                    t6 = None
            if t6 is not None:
                ops.segment_sum_bsum(mat(gh), g.csc_colptr, g.csc_eid, mat(gPs[:, : g.n_send]), mat(t6))
            else:
                ops.segment_sum(mat(gh), g.csc_colptr, g.csc_eid, mat(gPs[:, : g.n_send]))
            # 4. projections backward (batch-invariant operands: gradients summed over B first)
            gps_m, gpr_m = mat(gPs), mat(gPr)
            if sm.B == 1 and B > 1:
                t1 = _empty(1, N_s, d, device=dev)
                ops.sum_batch(gPs, t1)
                gps_m = mat(t1)
            if rm.B == 1 and B > 1:
                t2 = _empty(1, N_r, d, device=dev)
                ops.sum_batch(gPr, t2)
                gpr_m = mat(t2)
            # batch-invariant edge operand (g2m / m2g: Pe = W1e e with e the same for every sample):
            # its gradient is summed over the batch ONCE and that sum serves the data gradient and
            # the weight gradient dW1e = (sum_b gh_b)^T e -- the weight-gradient pass read all B
            # slices of gh (522 MB of the 3.2 GB of wide_outer_multi@m2g at hidden 128) for a
            # product that is linear in them
            dPe = mat(gh)
            if t6 is not None:
                dPe = mat(t6)
            elif not ctx.update_edges and em.B == 1 and B > 1:
                t6 = _empty(1, M, d, device=dev)
                ops.sum_batch(gh, t6)
                dPe = mat(t6)
            ge_w = dPe if em.B == 1 else mat(gh)
            outers += [(gps_m, sm, dW1[:, d : 2 * d], dummy[1], False),
                       (gpr_m, rm, dW1[:, 2 * d :], db1, False),
                       (ge_w, em, dW1[:, :d], dummy[2], False)]
            if lane.enabled:   # the remaining weight gradients run next to step 5
                with lane:
                    outer_multi(outers)
                outers = []
            # 5. data gradients of the three projections (e' = e + m adds g_e' to the edge one)
            g_e = _empty(dPe.B, M, d, device=dev)
            g_send = _empty(sm.B, N_s, d, device=dev)
            if same:
                t4 = _empty(sm.B, N_s, d, device=dev)
                lin_bwd_data_multi([(gps_m, W1s, mat(t4), mat(g_rec)), (dPe, W1e, mat(g_e), geo)])
                lin_bwd_data(gpr_m, W1r, mat(g_send), mat(t4))
                g_rec_total = None
            else:
                from .fused import _give_rec, _take_addend

                g_rec_total = _empty(rm.B, N_r, d, device=dev)
                add_s = _take_addend(ctx, g_send)   # (another consumer's gradient on send_rep)
                lin_bwd_data_multi([(gps_m, W1s, mat(g_send), mat(add_s) if add_s is not None else None),
                                    (gpr_m, W1r, mat(g_rec_total), mat(g_rec)),
                                    (dPe, W1e, mat(g_e), geo)])
                _give_rec(ctx, g_rec_total, g_send)
            g_edge = g_e
            if ctx.update_edges and em.B == 1 and B > 1:
                t5 = _empty(1, M, d, device=dev)
                ops.sum_batch(g_e, t5)
                g_edge = t5
            # 6. every weight / bias gradient of the layer: one streaming launch (small layers
            #    have issued theirs on the weight-gradient lane already: only the join is left)
            if outers and ctx.deferred and not lane.enabled and B * M <= DEFER_OUTER_ROWS:
                defer_outers(outers)   # (merged with the other small layers' at the end of the step)
            elif outers:
                with lane:
                    outer_multi(outers)
            lane.finish()
        return (g_send, g_rec_total, g_edge, None, None, None, None,
                dW1, db1, dW2, db2, dgam, dbet, dV1, dc1, dV2, dc2, dg2, db2n, None, None, None)


def apply_inet(net, send_rep, rec_rep, edge_rep):
    from .fused import _base, _mlp_parts

    same = send_rep is rec_rep
    s, e = _base(send_rep), _base(edge_rep)
    r = s if same else _base(rec_rep)
    el, al = _mlp_parts(net.edge_mlp), _mlp_parts(net.aggr_mlp)
    from .fused import _sink

    params = [el[0][0].weight, el[0][0].bias, el[0][1].weight, el[0][1].bias, el[1].weight, el[1].bias,
              al[0][0].weight, al[0][0].bias, al[0][1].weight, al[0][1].bias, al[1].weight, al[1].bias]
    prox = _proxies(params)
    return WideInteractionNetFunction.apply(
        s, r, e, same, net.tables, net.update_edges, net.aggr == "mean",
        *(prox if prox is not None else params),
        None if same else _sink(send_rep, "take"),
        None if same else (_sink(rec_rep, "give"), _sink(send_rep, "give")), prox is not None)


# ------------------------------------------- InteractionNet with SplitMLPs (hidden 128 / 256)
# HiLAMParallel (reference hi_lam_parallel.py:26-53) runs ONE InteractionNet over the union of
# every level's edges with a separate edge MLP per edge set and a separate node MLP per mesh level
# (SplitMLPs, interaction_net.py:134-163).  As at hidden 64 (fused.apply_inet_split) the layer is
# composed of an edge pass per edge chunk on that chunk's sub-graph (its own receiver-aligned
# tiles over ALL receivers; aggregates of the chunks are summed) and a node update per row range.
def inet_split_eligible(net, send_rep, rec_rep, edge_rep):
    from .fused import FORCE_GENERIC
    from .interaction_net import SplitMLPs

    if FORCE_GENERIC or not edge_rep.is_cuda or edge_rep.dtype != torch.float32:
        return False
    e_split = isinstance(net.edge_mlp, SplitMLPs)
    a_split = isinstance(net.aggr_mlp, SplitMLPs)
    if not (e_split or a_split):
        return False
    if net.hidden_layers != 1 or net.input_dim != net.hidden_dim:
        return False
    if net.hidden_dim not in WIDE_HIDDEN or not enabled(net.hidden_dim):
        return False
    if send_rep.dim() != 3 or rec_rep.dim() != 3 or edge_rep.dim() != 3:
        return False
    for t in (send_rep, rec_rep, edge_rep):
        if not _aligned(mat(t.detach())):
            return False
    tabs = list(net.chunk_tables) if e_split else [net.tables]
    return all(t.ntiles > 0 or t.virtual is not None for t in tabs)


class WideEdgePassFunction(torch.autograd.Function):
    """(send_rep, rec_rep, the edge rows of one chunk) -> (UNSCALED sum-aggregate of the chunk's
    messages for every receiver[, e + m for the chunk's edges]): nlam_lin_fwd_multi + nlam_tail_fwd;
    backward nlam_tail_bwd + nlam_segment_sum + nlam_lin_bwd_data_multi + one weight-gradient launch."""

    @staticmethod
    def forward(ctx, send_rep, rec_rep, edge_rep, same, g, update_edges, W1, b1, W2, b2, gam, bet):
        with ops.tag(g.tag):
            dev = edge_rep.device
            d = W2.shape[0]
            B = max(send_rep.shape[0], rec_rep.shape[0], edge_rep.shape[0])
            N_s, N_r, M = send_rep.shape[1], rec_rep.shape[1], edge_rep.shape[1]
            sm, rm, em = mat(send_rep.detach()), mat(rec_rep.detach()), mat(edge_rep.detach())
            W1e, W1s, W1r = W1[:, :d], W1[:, d : 2 * d], W1[:, 2 * d :]
            Ps = _inter(d, sm.B, N_s, d, device=dev)
            Pr = _inter(d, rm.B, N_r, d, device=dev)
            Pe = _inter(d, em.B, M, d, device=dev)
            lin_fwd_multi([(sm, W1s, None, _m(Ps)), (rm, W1r, b1, _m(Pr)), (em, W1e, None, _m(Pe))])
            h_e = _inter(d, B, M, d, device=dev)
            agg = _empty(B, N_r, d, device=dev)
            e_out = _empty(B, M, d, device=dev) if update_edges else None
            z_e = keep_z(B, M, d, gam, dev)
            virt = g.virtual is not None
            agg_k = _empty(B, recv_rows(g), d, device=dev) if virt else agg
            tail_fwd(Tiling(M, g), _m(Pe), g.csr_eid, _m(Ps), g.csr_send, _m(Pr), g.csr_rec, W2, b2,
                     gam, bet, h_e, mat(e_out) if update_edges else None,
                     g.csr_eid if update_edges else None, em if update_edges else None,
                     mat(agg_k), None, B, d, z_e)
            if virt:
                fold_virtual(g, agg_k, mat(agg))
            ctx.save_for_backward(W1, b1, W2, b2, gam, h_e, z_e)
            ctx.set_materialize_grads(False)
            ctx.g, ctx.same, ctx.update_edges = g, same, update_edges
            ctx.mats = (sm, rm, em)
            ctx.dims = (B, N_s, N_r, M, d)
        if update_edges:
            return agg, e_out
        return agg

    @staticmethod
    def backward(ctx, g_agg, g_edge_out=None):
        with ops.tag(ctx.g.tag), ops.slab_batch():
            W1, b1, W2, b2, gam, h_e, z_e = ctx.saved_tensors
            g = ctx.g
            sm, rm, em = ctx.mats
            B, N_s, N_r, M, d = ctx.dims
            dev = W1.device
            W1e, W1s, W1r = W1[:, :d], W1[:, d : 2 * d], W1[:, 2 * d :]
            same = ctx.same
            if g_agg is None:
                g_agg = torch.zeros(B, N_r, d, dtype=torch.float32, device=dev)
            g_agg = g_agg.contiguous()
            dW1, db1 = _empty(d, 3 * d, device=dev), _empty(d, device=dev)
            dW2, db2 = torch.empty_like(W2), _empty(d, device=dev)
            dgam, dbet = _empty(d, device=dev), _empty(d, device=dev)
            gz_e = _inter(d, B, M, d, device=dev)
            gh = _empty(B, M, d, device=dev)
            gPr = _empty(B, N_r, d, device=dev)
            geo = None
            if ctx.update_edges and g_edge_out is not None:
                geo = mat(g_edge_out.contiguous())
            virt = g.virtual is not None
            gPr_k = _empty(B, recv_rows(g), d, device=dev) if virt else gPr
            tail_bwd(Tiling(M, g), h_e, mat(g_agg), g.csr_rec, None, geo,
                     g.csr_eid if geo is not None else None, W2, b2, gam, gz_e, mat(gh), g.csr_eid,
                     mat(gPr_k), B, d, dgam, dbet, z_e)
            if virt:
                fold_virtual(g, gPr_k, mat(gPr))
            gPs = (torch.zeros if N_s > g.n_send else torch.empty)(
                B, N_s, d, dtype=torch.float32, device=dev)
            ops.segment_sum(mat(gh), g.csc_colptr, g.csc_eid, mat(gPs[:, : g.n_send]))
            gps_m, gpr_m = mat(gPs), mat(gPr)
            if sm.B == 1 and B > 1:
                t1 = _empty(1, N_s, d, device=dev)
                ops.sum_batch(gPs, t1)
                gps_m = mat(t1)
            if rm.B == 1 and B > 1:
                t2 = _empty(1, N_r, d, device=dev)
                ops.sum_batch(gPr, t2)
                gpr_m = mat(t2)
            dummy = [_empty(d, device=dev) for _ in range(2)]
            outers = [(_m(gz_e), _m(h_e), dW2, db2, True),
                      (gps_m, sm, dW1[:, d : 2 * d], dummy[0], False),
                      (gpr_m, rm, dW1[:, 2 * d :], db1, False),
                      (mat(gh), em, dW1[:, :d], dummy[1], False)]
            dPe = mat(gh)
            if not ctx.update_edges and em.B == 1 and B > 1:
                t6 = _empty(1, M, d, device=dev)
                ops.sum_batch(gh, t6)
                dPe = mat(t6)
            g_e = _empty(dPe.B, M, d, device=dev)
            g_send = _empty(sm.B, N_s, d, device=dev)
            if same:
                t4 = _empty(sm.B, N_s, d, device=dev)
                lin_bwd_data_multi([(gps_m, W1s, mat(t4), None), (dPe, W1e, mat(g_e), geo)])
                lin_bwd_data(gpr_m, W1r, mat(g_send), mat(t4))
                g_rec_total = None
            else:
                g_rec_total = _empty(rm.B, N_r, d, device=dev)
                lin_bwd_data_multi([(gps_m, W1s, mat(g_send), None),
                                    (gpr_m, W1r, mat(g_rec_total), None),
                                    (dPe, W1e, mat(g_e), geo)])
            g_edge = g_e
            if ctx.update_edges and em.B == 1 and B > 1:
                t5 = _empty(1, M, d, device=dev)
                ops.sum_batch(g_e, t5)
                g_edge = t5
            outer_multi(outers)
        return (g_send, g_rec_total, g_edge, None, None, None, dW1, db1, dW2, db2, dgam, dbet)


class WideNodeUpdateFunction(torch.autograd.Function):
    """x_r + LN(V2 silu(V1 [x_r | agg] + c1) + c2) on a range of receiver rows."""

    @staticmethod
    def forward(ctx, x_r, agg, tag, V1, c1, V2, c2, gam2, bet2):
        with ops.tag(tag):
            dev = x_r.device
            d = V2.shape[0]
            B, N = x_r.shape[0], x_r.shape[1]
            xm, am = mat(x_r.detach()), mat(agg.detach())
            hn1 = _inter(d, B, N, d, device=dev)
            hn2 = _inter(d, B, N, d, device=dev)
            lin_fwd_multi([(xm, V1[:, :d], c1, _m(hn1)), (am, V1[:, d:], None, _m(hn2))])
            h_n = _inter(d, B, N, d, device=dev)
            out = _empty(B, N, d, device=dev)
            z_n = keep_z(B, N, d, gam2, dev)
            tail_fwd(Tiling(N), _m(hn1), None, _m(hn2), None, None, None, V2, c2, gam2, bet2, h_n,
                     mat(out), None, xm, None, None, B, d, z_n)
            ctx.save_for_backward(V1, c1, V2, c2, gam2, h_n, z_n)
            ctx.tag, ctx.mats, ctx.dims = tag, (xm, am), (B, N, d)
        return out

    @staticmethod
    def backward(ctx, gy):
        with ops.tag(ctx.tag), ops.slab_batch():
            V1, c1, V2, c2, gam2, h_n, z_n = ctx.saved_tensors
            xm, am = ctx.mats
            B, N, d = ctx.dims
            dev = V1.device
            gy = gy.contiguous()
            dV1, dc1 = torch.empty_like(V1), _empty(d, device=dev)
            dV2, dc2 = torch.empty_like(V2), _empty(d, device=dev)
            dg2, db2n = _empty(d, device=dev), _empty(d, device=dev)
            gz_n = _inter(d, B, N, d, device=dev)
            ga_n = _empty(B, N, d, device=dev)
            tail_bwd(Tiling(N), h_n, mat(gy), None, None, None, None, V2, c2, gam2, gz_n, mat(ga_n),
                     None, None, B, d, dg2, db2n, z_n)
            g_x = _empty(B, N, d, device=dev)
            g_agg = _empty(B, N, d, device=dev)
            lin_bwd_data_multi([(mat(ga_n), V1[:, :d], mat(g_x), mat(gy)),
                                (mat(ga_n), V1[:, d:], mat(g_agg), None)])
            dummy = _empty(d, device=dev)
            outer_multi([(_m(gz_n), _m(h_n), dV2, dc2, True), (mat(ga_n), xm, dV1[:, :d], dc1, False),
                         (mat(ga_n), am, dV1[:, d:], dummy, False)])
        return (g_x, g_agg, None, dV1, dc1, dV2, dc2, dg2, db2n)


def apply_inet_split(net, send_rep, rec_rep, edge_rep):
    from .fused import _base, _mlp_parts
    from .interaction_net import SplitMLPs

    same = send_rep is rec_rep
    s = _base(send_rep)
    r = s if same else _base(rec_rep)
    e = _base(edge_rep)
    B = max(s.shape[0], r.shape[0], e.shape[0])
    if isinstance(net.edge_mlp, SplitMLPs):
        e_mlps, e_sizes, tabs = (list(net.edge_mlp.mlps), list(net.edge_mlp.chunk_sizes),
                                 list(net.chunk_tables))
    else:
        e_mlps, e_sizes, tabs = [net.edge_mlp], [net.tables.M], [net.tables]
    # torch.split, as the reference's SplitMLPs.forward does (interaction_net.py:159-163): its
    # backward is ONE concatenation of the chunk gradients; indexing e[:, o : o + m] per chunk gave
    # a zero-filled full-size gradient + a copy per chunk and a chain of full-size adds
    aggs, e_outs = [], []
    e_chunks = torch.split(e, e_sizes, dim=1) if len(e_sizes) > 1 else (e,)
    for mlp, e_c, tab in zip(e_mlps, e_chunks, tabs):
        tab.tag = net.tables.tag
        lin, ln = _mlp_parts(mlp)
        out = WideEdgePassFunction.apply(
            s, r, e_c, same, tab, net.update_edges,
            lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, ln.weight, ln.bias)
        if net.update_edges:
            a_c, eo_c = out
            e_outs.append(eo_c)
        else:
            a_c = out
        aggs.append(a_c)
    from . import glue

    agg = glue.sum_many(aggs)   # (one pass over the chunk aggregates, not a chain of adds)
    if net.aggr == "mean":
        agg = agg * net.tables.inv_deg.view(1, -1, 1)
    if isinstance(net.aggr_mlp, SplitMLPs):
        a_mlps, a_sizes = list(net.aggr_mlp.mlps), list(net.aggr_mlp.chunk_sizes)
    else:
        a_mlps, a_sizes = [net.aggr_mlp], [net.num_rec]
    if r.shape[0] == 1 and B > 1:
        r = r.expand(B, -1, -1)
    outs = []
    r_chunks = torch.split(r, a_sizes, dim=1) if len(a_sizes) > 1 else (r,)
    g_chunks = torch.split(agg, a_sizes, dim=1) if len(a_sizes) > 1 else (agg,)
    for mlp, r_c, g_c in zip(a_mlps, r_chunks, g_chunks):
        lin, ln = _mlp_parts(mlp)
        outs.append(WideNodeUpdateFunction.apply(
            r_c, g_c, net.tables.tag,
            lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, ln.weight, ln.bias))
    rec_out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
    if net.update_edges:
        return rec_out, (e_outs[0] if len(e_outs) == 1 else torch.cat(e_outs, dim=1))
    return rec_out
