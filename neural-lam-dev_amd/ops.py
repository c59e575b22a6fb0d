"""Thin tensor-level wrappers over the C ABI (include/nlam_hip.h).

torch is used for device memory, streams and autograd bookkeeping only; every
arithmetic operation below is a kernel of libnlam_hip.so launched on torch's
current stream.  All functions require fp32 HIP-device tensors and raise
otherwise (no CPU fallback).
"""
import ctypes
import os
import sys
from collections import namedtuple

import torch

from ._lib import NlamError, check, lib

# A (B, rows, cols) fp32 matrix stack in device memory with unit column stride.
# bstride may be 0 (batch-invariant).  `keep` pins the owning tensor.
Mat = namedtuple("Mat", "ptr B rows cols bstride ld keep")


def stream():
    return torch.cuda.current_stream().cuda_stream


class KernelProfiler:
    """Brackets every C-ABI launch with HIP events on the launch stream (torch's
    current stream is the stream the kernels are launched on) and accumulates
    per-entry-point time, algorithmic flops and algorithmic bytes.  Used by
    bench.py for the live `roofline` numbers; off (None) otherwise."""

    def __init__(self):
        self.pending = []  # (name, start, end, flops, bytes)
        self.stats = {}

    def add(self, name, start, end, flops, nbytes):
        self.pending.append((name, start, end, flops, nbytes))

    def collect(self):
        torch.cuda.synchronize()
        for name, s, e, flops, nbytes in self.pending:
            st = self.stats.setdefault(name, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            st["calls"] += 1
            st["ms"] += s.elapsed_time(e)
            st["flops"] += flops
            st["bytes"] += nbytes
        self.pending = []
        return self.stats


PROFILER = None
_TAG = []


class tag:
    """with ops.tag("m2m"): launches inside are accounted as "<entry>@m2m"."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _TAG.append(self.name)

    def __exit__(self, *a):
        _TAG.pop()


_DEBUG_SYNC = os.environ.get("NLAM_DEBUG_SYNC", "0") == "1"


def _launch(name, fn, args, flops=0.0, nbytes=0.0):
    if _DEBUG_SYNC:  # print-before-launch + sync-after: the last line names a faulting kernel
        print(f"[nlam] {name} {args}", file=sys.stderr, flush=True)
        check(fn(*args), name)
        torch.cuda.synchronize()
        return
    prof = PROFILER
    if prof is None:
        check(fn(*args), name)
        return
    if _TAG:
        name = f"{name}@{_TAG[-1]}"
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    check(fn(*args), name)
    e.record()
    prof.add(name, s, e, flops, nbytes)


def _require_dev(t, what="tensor"):
    if not t.is_cuda:
        raise RuntimeError(
            f"neural_lam_amd: {what} is on {t.device}; the MI355X HIP path needs device tensors "
            "(there is no CPU fallback)"
        )
    if t.dtype != torch.float32:
        raise RuntimeError(f"neural_lam_amd: {what} has dtype {t.dtype}, expected float32")


def mat(t, col_off=0, cols=None):
    """Describe `t` ((rows, c) or (B, rows, c), or more leading dims that
    flatten) as a Mat, optionally restricted to columns [col_off, col_off+cols)."""
    _require_dev(t)
    if t.dim() == 1:
        t = t.unsqueeze(0)
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() > 3:
        t = t.reshape(-1, t.shape[-2], t.shape[-1])
    if t.shape[-1] > 1 and t.stride(-1) != 1:
        t = t.contiguous()
    B, rows, c = t.shape
    bs, ld = t.stride(0), t.stride(1)
    if rows == 1:
        ld = max(ld, c)
    if B == 1:
        bs = 0
    if ld < c or (bs != 0 and bs < rows * ld):
        t = t.contiguous()
        bs, ld = t.stride(0) if B > 1 else 0, t.stride(1)
    cols = c - col_off if cols is None else cols
    assert 0 <= col_off and col_off + cols <= c
    return Mat(t.data_ptr() + 4 * col_off, B, rows, cols, bs, ld, t)


def mat16(t):
    """Mat of a contiguous bf16 (B, rows, c) tensor: a row operand stored in bf16 (hidden 256;
    pitches in bf16 elements; see include/nlam_hip.h 'bf16 STORAGE')."""
    assert t.dtype == torch.bfloat16 and t.is_cuda and t.dim() == 3 and t.is_contiguous()
    B, rows, c = t.shape
    return Mat(t.data_ptr(), B, rows, c, rows * c if B > 1 else 0, c, t)


def is_bf16(m):
    return m is not None and m.keep is not None and m.keep.dtype == torch.bfloat16


def flat(m):
    """(B, rows) -> one (B*rows)-row matrix if the batch pitch allows it."""
    if m.B == 1:
        return m
    if m.bstride == m.rows * m.ld:
        return Mat(m.ptr, 1, m.B * m.rows, m.cols, 0, m.ld, m.keep)
    return None


def batch_item(m, b):
    return Mat(m.ptr + 4 * b * m.bstride, 1, m.rows, m.cols, 0, m.ld, m.keep)


def _each_flat(*ms):
    """Yield tuples of 2-D Mats covering the batch (one tuple if all flatten)."""
    fl = [flat(m) for m in ms]
    if all(f is not None for f in fl):
        yield tuple(fl)
        return
    B = max(m.B for m in ms)
    for b in range(B):
        yield tuple(batch_item(m, b if m.B > 1 else 0) for m in ms)


def _ws(n, device):
    return torch.empty(max(int(n), 1), dtype=torch.float32, device=device)


# ------------------------------------------------------------------ linear
def _splitk_for(rows):
    return int(max(1, min(256, rows // 512)))


def linear_fwd(x, W, b, out, accumulate=False):
    """out = x @ W.T (+ b) (+ out).  x: Mat (rows x in), W: (out_f, in) tensor."""
    of, inf = W.shape
    assert x.cols == inf and out.cols == of
    for xm, om in _each_flat(x, out):
        _launch(
            "nlam_gemm", lib.nlam_gemm,
            (xm.rows, of, inf, xm.ptr, xm.ld, 1, W.data_ptr(), 1, W.stride(0),
             b.data_ptr() if b is not None else None, om.ptr, om.ld, int(accumulate), 1,
             None, stream()),
            flops=2.0 * xm.rows * of * inf, nbytes=4.0 * (xm.rows * (of + inf) + of * inf),
        )


def linear_bwd_data(gy, W, gx, accumulate=False):
    """gx = gy @ W  (rows x in)."""
    of, inf = W.shape
    assert gy.cols == of and gx.cols == inf
    for gm, xm in _each_flat(gy, gx):
        _launch(
            "nlam_gemm", lib.nlam_gemm,
            (gm.rows, inf, of, gm.ptr, gm.ld, 1, W.data_ptr(), W.stride(0), 1, None, xm.ptr,
             xm.ld, int(accumulate), 1, None, stream()),
            flops=2.0 * gm.rows * of * inf, nbytes=4.0 * (gm.rows * (of + inf) + of * inf),
        )


def linear_bwd_weight(gy, x, dW, db):
    """dW = gy.T @ x (out_f x in), db = column sums of gy; deterministic split-K."""
    of, inf = dW.shape
    dev = dW.device
    first = True
    for gm, xm in _each_flat(gy, x):
        sk = _splitk_for(gm.rows)
        ws = _ws(sk * of * inf, dev) if sk > 1 else None
        _launch(
            "nlam_gemm", lib.nlam_gemm,
            (of, inf, gm.rows, gm.ptr, 1, gm.ld, xm.ptr, xm.ld, 1, None, dW.data_ptr(),
             dW.stride(0), int(not first), sk, ws.data_ptr() if ws is not None else None,
             stream()),
            flops=2.0 * gm.rows * of * inf, nbytes=4.0 * (gm.rows * (of + inf) + of * inf),
        )
        if db is not None:
            colsum(gm, db, accumulate=not first)
        first = False


def colsum(x, out, accumulate=False):
    first = not accumulate
    for (xm,) in _each_flat(x):
        nb = lib.nlam_colsum_blocks(xm.rows)
        part = _ws(nb * xm.cols, out.device)
        _launch(
            "nlam_colsum", lib.nlam_colsum,
            (xm.ptr, xm.ld, out.data_ptr(), int(not first), part.data_ptr(), xm.rows, xm.cols,
             stream()),
            nbytes=4.0 * xm.rows * xm.cols,
        )
        first = False


# -------------------------------------------------------------- elementwise
def silu_fwd(x, y):
    """x, y: contiguous tensors of equal numel."""
    _launch("nlam_silu_fwd", lib.nlam_silu_fwd, (x.data_ptr(), y.data_ptr(), x.numel(), stream()),
            nbytes=8.0 * x.numel())


def silu_bwd(x, gy, gx):
    _launch("nlam_silu_bwd", lib.nlam_silu_bwd,
            (x.data_ptr(), gy.data_ptr(), gx.data_ptr(), x.numel(), stream()),
            nbytes=12.0 * x.numel())


def layernorm_fwd(z, gamma, beta, res, y):
    for parts in _each_flat(*([z, y] + ([res] if res is not None else []))):
        zm, ym = parts[0], parts[1]
        rm = parts[2] if res is not None else None
        _launch(
            "nlam_layernorm_fwd", lib.nlam_layernorm_fwd,
            (zm.ptr, zm.ld, gamma.data_ptr(), beta.data_ptr(),
             rm.ptr if rm is not None else None, rm.ld if rm is not None else 0, ym.ptr, ym.ld,
             zm.rows, zm.cols, stream()),
            nbytes=4.0 * zm.rows * zm.cols * (3 if rm is not None else 2),
        )


def layernorm_bwd(z, gamma, gy, gz, dgamma, dbeta):
    first = True
    for zm, gm, om in _each_flat(z, gy, gz):
        nb = lib.nlam_layernorm_bwd_blocks(zm.rows)
        part = _ws(2 * nb * zm.cols, dgamma.device)
        _launch(
            "nlam_layernorm_bwd", lib.nlam_layernorm_bwd,
            (zm.ptr, zm.ld, gamma.data_ptr(), gm.ptr, gm.ld, om.ptr, om.ld, dgamma.data_ptr(),
             dbeta.data_ptr(), int(not first), part.data_ptr(), zm.rows, zm.cols, stream()),
            nbytes=12.0 * zm.rows * zm.cols,
        )
        first = False


def add_rows(a, b, out):
    for am, bm, om in _each_flat(a, b, out):
        _launch(
            "nlam_add_rows", lib.nlam_add_rows,
            (am.ptr, am.ld, bm.ptr, bm.ld, om.ptr, om.ld, am.rows, am.cols, stream()),
            nbytes=12.0 * am.rows * am.cols,
        )


def copy_rows(x, out):
    """out[b] = x[b]; x may be batch-invariant (bstride 0) and is then broadcast."""
    B = out.B
    _launch(
        "nlam_copy_rows", lib.nlam_copy_rows,
        (x.ptr, x.bstride, x.ld, out.ptr, out.bstride, out.ld, B, out.rows, out.cols, stream()),
        nbytes=8.0 * B * out.rows * out.cols,
    )


def sum_batch(x, out):
    """out (n,) = sum over batch of contiguous x (B, n)."""
    B = x.shape[0]
    n = x.numel() // B
    _launch("nlam_sum_batch", lib.nlam_sum_batch, (x.data_ptr(), n, out.data_ptr(), B, n, stream()),
            nbytes=4.0 * n * (B + 1))


# ---------------------------------------------------------- gather / segment
def gather_rows(x, idx, out, row_scale=None):
    """out[b][k] = x[b][idx[k]] (* row_scale[idx[k]])."""
    if idx.numel() != out.rows or idx.dtype != torch.int32:
        raise ValueError(f"gather_rows: idx has {idx.numel()} entries for {out.rows} output rows")
    _launch(
        "nlam_gather_rows", lib.nlam_gather_rows,
        (x.ptr, x.bstride, x.ld, idx.data_ptr(),
         row_scale.data_ptr() if row_scale is not None else None, out.ptr, out.bstride, out.ld,
         out.B, out.rows, out.cols, stream()),
        nbytes=out.B * (4.0 * out.cols * (out.rows + x.rows) + 4.0 * out.rows),
    )


def segment_sum(src, rowptr, pos, out, scale=None, accumulate=False):
    """out[b][i] = scale[i] * sum_{p in segment i} src[b][pos[p]]."""
    if rowptr.numel() != out.rows + 1 or rowptr.dtype != torch.int32:
        raise ValueError(
            f"segment_sum: rowptr has {rowptr.numel()} entries for {out.rows} output rows"
        )
    if scale is not None and scale.numel() != out.rows:
        raise ValueError("segment_sum: scale length does not match the output rows")
    # algorithmic bytes: read src rows 4dM + write 4dN + int32
    # indices 4M + 4(N+1), per batch item
    _launch(
        "nlam_segment_sum", lib.nlam_segment_sum,
        (src.ptr, src.bstride, src.ld, rowptr.data_ptr(),
         pos.data_ptr() if pos is not None else None,
         scale.data_ptr() if scale is not None else None, out.ptr, out.bstride, out.ld,
         int(accumulate), out.B, out.rows, out.cols, stream()),
        nbytes=out.B * (4.0 * out.cols * (src.rows + out.rows) + 4.0 * src.rows
                        + 4.0 * (out.rows + 1)),
    )


def segment_sum_bsum_ok(src, out, bsum):
    d = out.cols
    return (d in (64, 128, 256) and out.B * d <= 1024 and src.B == out.B and bsum.B == 1
            and all(m.ptr % 16 == 0 and m.ld % 4 == 0 and m.bstride % 4 == 0 for m in (src, out, bsum))
            and os.environ.get("NLAM_SEGSUM_BSUM", "1") != "0")


def segment_sum_bsum(src, rowptr, pos, out, bsum):
    """segment_sum AND bsum[pos[p]] = sum_b src[b][pos[p]] for every listed row, one pass."""
    if rowptr.numel() != out.rows + 1 or rowptr.dtype != torch.int32:
        raise ValueError(f"segment_sum_bsum: rowptr has {rowptr.numel()} entries for {out.rows} output rows")
    _launch(
        "nlam_segment_sum", lib.nlam_segment_sum_bsum,
        (src.ptr, src.bstride, src.ld, rowptr.data_ptr(), pos.data_ptr() if pos is not None else None,
         out.ptr, out.bstride, out.ld, bsum.ptr, bsum.ld, out.B, out.rows, out.cols, stream()),
        nbytes=out.B * (4.0 * out.cols * (src.rows + out.rows) + 4.0 * src.rows + 4.0 * (out.rows + 1))
        + 4.0 * out.cols * src.rows,
    )


def mfma_probe():
    out = torch.empty(32, 32, dtype=torch.float32, device="cuda")
    check(lib.nlam_mfma_probe(out.data_ptr(), stream()), "nlam_mfma_probe")
    return out


# ------------------------------------------------------------------- fused
def _p(t):
    return t.data_ptr() if t is not None else None


def fused_mlp_fwd(xa, xb, W1, b1, W2, b2, gamma, beta, res, out, hid, n_out):
    """xa/xb/res/out: Mat (xb, res may be None); W1 (hid, k_in) and W2 (n_out, hid)
    tensors (any row pitch, unit column stride)."""
    B = out.B
    k_in = xa.cols + (xb.cols if xb is not None else 0)
    _launch(
        "nlam_mlp_fwd", lib.nlam_mlp_fwd,
        (xa.ptr, xa.bstride, xa.ld, xa.cols,
         xb.ptr if xb is not None else None, xb.bstride if xb is not None else 0,
         xb.ld if xb is not None else 0, xb.cols if xb is not None else 0,
         W1.data_ptr(), W1.stride(0), _p(b1), W2.data_ptr(), W2.stride(0), _p(b2),
         _p(gamma), _p(beta),
         res.ptr if res is not None else None, res.bstride if res is not None else 0,
         res.ld if res is not None else 0,
         out.ptr, out.bstride, out.ld, B, out.rows, hid, n_out, stream()),
        flops=2.0 * B * out.rows * hid * (k_in + n_out),
        nbytes=4.0 * B * out.rows * (k_in + n_out * (2 if res is not None else 1)),
    )


def grid_encode_supported():
    return bool(lib.nlam_grid_encode_supported())


def grid_encode_fwd(srcs, emb_w, Ws, enc_w, Wr, br, feat, emb, ps, rep, pr):
    """One pass over the grid rows (csrc/fused16_grid.hip): srcs = Mats of the grid features'
    sources; emb_w / enc_w = (W1, b1, W2, b2, gamma, beta) of grid_embedder / encoding_grid_mlp;
    Ws / (Wr, br) = the sender third of g2m's and the receiver third of m2g's first edge-MLP
    Linear; feat (or None), emb, ps, rep, pr: contiguous (B, rows, .) output tensors."""
    n = len(srcs)
    B, rows = emb.shape[0], emb.shape[1]
    k_in = sum(m.cols for m in srcs)
    P, I64, I32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
    W1, b1, W2, b2, gam, bet = emb_w
    E1, e1, E2, e2, egam, ebet = enc_w
    _launch(
        "nlam_grid_encode_fwd", lib.nlam_grid_encode_fwd,
        (n, P(*[m.ptr for m in srcs]), I64(*[m.bstride for m in srcs]), I64(*[m.ld for m in srcs]),
         I32(*[m.cols for m in srcs]),
         W1.data_ptr(), W1.stride(0), b1.data_ptr(), W2.data_ptr(), W2.stride(0), b2.data_ptr(),
         gam.data_ptr(), bet.data_ptr(), Ws.data_ptr(), Ws.stride(0),
         E1.data_ptr(), E1.stride(0), e1.data_ptr(), E2.data_ptr(), E2.stride(0), e2.data_ptr(),
         egam.data_ptr(), ebet.data_ptr(), Wr.data_ptr(), Wr.stride(0), _p(br),
         _p(feat), emb.data_ptr(), ps.data_ptr(), rep.data_ptr(), pr.data_ptr(), B, rows, stream()),
        flops=2.0 * B * rows * 64 * (k_in + 5 * 64),
        # algorithmic bytes: the sources once, the five outputs once
        nbytes=4.0 * rows * (sum((m.B if m.bstride else 1) * m.cols for m in srcs)
                             + B * ((k_in if feat is not None else 0) + 4 * 64)),
    )


def fused_lin_fwd(x, WA, bA, WB, bB, out):
    """out[:, :nA] = x WA^T + bA ; out[:, nA:] = x WB^T + bB.  WA/WB: 2-D weight
    views (any row pitch)."""
    nA = WA.shape[0]
    nB = WB.shape[0] if WB is not None else 0
    _launch(
        "nlam_lin_fwd", lib.nlam_lin_fwd,
        (x.ptr, x.bstride, x.ld, x.cols, WA.data_ptr(), WA.stride(0), _p(bA), nA,
         _p(WB), WB.stride(0) if WB is not None else 0, _p(bB), nB,
         out.ptr, out.bstride, out.ld, out.B, out.rows, int(is_bf16(out)), stream()),
        flops=2.0 * out.B * out.rows * x.cols * (nA + nB),
        nbytes=out.B * out.rows * (4.0 * x.cols + (2.0 if is_bf16(out) else 4.0) * (nA + nB)),
    )


def fused_edge_fwd(g, e, has_egemm, ps, pr, W1e, W2, b2, gamma, beta, agg, e_out, mean, d):
    """g: EdgeTables (with tiles); e/ps/pr/agg/e_out: Mat."""
    B = agg.B
    M = g.M
    units = 2 if has_egemm else 1
    _launch(
        "nlam_edge_fwd", lib.nlam_edge_fwd,
        (g.tiles.data_ptr(), g.ntiles, g.csr_rowptr.data_ptr(), g.csr_eid.data_ptr(),
         g.csr_send.data_ptr(), g.csr_rec.data_ptr(), g.inv_deg.data_ptr() if mean else None,
         e.ptr, e.bstride, e.ld, int(has_egemm), ps.ptr, ps.bstride, ps.ld, pr.ptr, pr.bstride,
         pr.ld, _p(W1e), W1e.stride(0) if W1e is not None else 0, W2.data_ptr(), W2.stride(0),
         b2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), agg.ptr, agg.bstride, agg.ld,
         e_out.ptr if e_out is not None else None, e_out.bstride if e_out is not None else 0,
         e_out.ld if e_out is not None else 0, B, d, stream()),
        flops=2.0 * B * M * d * d * units,
        # algorithmic bytes: read e (per batch item if it varies), write e',
        # read the projected node rows once, write agg, int32 indices
        nbytes=4.0 * d * ((B if e.bstride else 1) * M + (B * M if has_egemm else 0)
                          + B * (ps.rows + pr.rows) + B * agg.rows) + 16.0 * M,
    )


def _ntiles(B, rows):
    return B * ((rows + 31) // 32)


def reduce_slabs(slab, nslabs, stride, n, out, accumulate=False):
    _launch("nlam_reduce_slabs", lib.nlam_reduce_slabs,
            (slab.data_ptr(), nslabs, stride, n, out.data_ptr(), int(accumulate), stream()),
            nbytes=4.0 * nslabs * n)


_SLAB_BATCH = []   # stack of pending (slab, nslabs, stride, src_off, rows, cols, src_ld, dst)
MAX_SEGS = 64


# Slab reductions handed on to the end of an AR step's backward (slab_batch(defer=True)): the
# weight gradients of a layer feed nothing inside the backward, so the reductions of ALL layers of
# one predict_step can run as a few large launches instead of one small launch per layer.  Safe
# only when nothing reads the destinations before flush_deferred(): the layers that defer take
# their parameters through glue.DeferGrad, an identity node between the parameters and the layers,
# created at the START of predict_step: every gradient of a deferring layer reaches its parameter
# (input-buffer sums, AccumulateGrad, hooks) only THROUGH that node, whose backward -- the flush --
# cannot run before every layer that took parameters from it has run (the engine's dependency
# count), and, being the earliest-created node of the step, is not run before the step's other
# ready nodes either.  An engine callback flushes at the end of the pass as well (a node that
# never ran: nobody asked for parameter gradients).
_DEFERRED = []
_DEFER_TASK = [None]
# gradient reducers that overlap their bucket all-reduces with the backward through per-parameter
# hooks (parallel.GradAllReduce): while their hooks are live the deferral stays off -- it would make
# every gradient of the deferring layers "ready" only at the end of the AR step's backward
OVERLAP_REDUCERS = []


def deferral_allowed():
    live = [r() for r in OVERLAP_REDUCERS]
    return not any(r is not None and r.hooks_enabled and r.active and r.overlap for r in live)


_FLUSH_HOOKS = []   # work that must run before the deferred reductions (wide.flush_deferred_outers)


def on_flush(fn):
    """Run fn() at the next flush_deferred() (and make sure one comes at the end of this pass)."""
    if fn not in _FLUSH_HOOKS:
        _FLUSH_HOOKS.append(fn)
    task = torch._C._current_graph_task_id()
    if task != -1 and _DEFER_TASK[0] != task:
        _DEFER_TASK[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_pass_flush)


def flush_deferred():
    hooks = list(_FLUSH_HOOKS)
    del _FLUSH_HOOKS[:]
    for fn in hooks:
        fn()
    if _DEFERRED:
        entries = list(_DEFERRED)
        del _DEFERRED[:]
        with tag("deferred"):
            _flush_segments(entries)


def _end_of_pass_flush():
    _DEFER_TASK[0] = None
    flush_deferred()


class slab_batch:
    """Collect the slab reductions issued inside the block and finish them with ONE
    launch at exit (the destinations must not be read before that); defer=True: hand them on to
    flush_deferred() instead (see above)."""

    def __init__(self, defer=False):
        self.defer = bool(defer)

    def __enter__(self):
        _SLAB_BATCH.append([])
        return self

    def __exit__(self, et, ev, tb):
        pending = _SLAB_BATCH.pop()
        if et is not None:
            return False
        task = torch._C._current_graph_task_id() if self.defer else -1
        if task == -1:
            _flush_segments(pending)
            return False
        _DEFERRED.extend(pending)
        if _DEFER_TASK[0] != task:
            _DEFER_TASK[0] = task
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_pass_flush)
        return False


class WeightGradLane:
    """Second HIP stream for the weight-gradient passes of a small layer's backward.

    On the small mesh levels of Hi-LAM (reference hi_lam.py:82-207: 81 ... 6,561-node levels) every
    kernel of an InteractionNet backward is a handful of workgroups on a 256-CU device, so the
    backward is a chain of launch / memory latencies.  The weight gradients (nlam_wide_outer and
    their slab reductions) feed nothing inside that chain: they are issued on this lane, after the
    lane has waited for the main stream at the point where their operands exist, and the main
    stream waits for the lane once, before the backward returns.  Inside a HIP-graph capture the
    fork / join become graph edges.  `with lane:` = fork + launches on the lane (slab reductions
    collected); `lane.finish()` = flush the reductions on the lane and join.  Disabled lane: the
    block runs on the current stream and finish() only flushes."""

    _streams = {}

    def __init__(self, enabled, device):
        self.enabled = bool(enabled)
        self.pending = []
        self.ctx = None
        if self.enabled:
            key = torch.device(device).index
            if key not in WeightGradLane._streams:
                WeightGradLane._streams[key] = torch.cuda.Stream(device=device)
            self.side = WeightGradLane._streams[key]
            self.main = torch.cuda.current_stream(device)

    def __enter__(self):
        if self.enabled:
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        # a disabled lane inside a slab_batch(): its reductions join the enclosing block's single
        # launch (they used to be flushed by finish() as a second reduction launch per layer:
        # 46 launches of ~8 us per Hi-LAM step at hidden 128 / 256)
        self._own = self.enabled or not _SLAB_BATCH
        if self._own:
            _SLAB_BATCH.append(self.pending)
        return self

    def __exit__(self, et, ev, tb):
        if self._own:
            _SLAB_BATCH.pop()
        if self.enabled:
            self.ctx.__exit__(et, ev, tb)
            self.ctx = None
        return False

    def finish(self):
        if self.enabled:
            with torch.cuda.stream(self.side):
                _flush_segments(self.pending)
            self.main.wait_stream(self.side)
        else:
            _flush_segments(self.pending)
        self.pending = []


def _flush_segments(entries):
    for i in range(0, len(entries), MAX_SEGS):
        part = entries[i : i + MAX_SEGS]
        n = len(part)
        I64, I32, P = ctypes.c_int64 * n, ctypes.c_int32 * n, ctypes.c_void_p * n
        dst_ld = [e[7].stride(0) if e[7].dim() == 2 else e[7].numel() for e in part]
        _launch(
            "nlam_reduce_slabs_multi", lib.nlam_reduce_slabs_batch,
            (n, P(*[e[0].data_ptr() for e in part]), I64(*[e[1] for e in part]),
             I64(*[e[2] for e in part]), I64(*[e[3] for e in part]), I32(*[e[4] for e in part]),
             I32(*[e[5] for e in part]), I64(*[e[6] for e in part]),
             P(*[e[7].data_ptr() for e in part]), I64(*dst_ld), stream()),
            nbytes=4.0 * sum(e[1] * e[4] * e[5] for e in part),
        )


def reduce_segments(slab, nslabs, stride, segs):
    """For every (src_off, rows, cols, src_ld, dst) sum that matrix segment of the
    per-workgroup slabs into dst (a 1-D or 2-D fp32 view with unit column stride).
    One launch -- or, inside `slab_batch()`, deferred to the block's single launch.
    Deterministic."""
    entries = []
    for (off, r, c, ld, d) in segs:
        if d is None:
            continue
        assert d.dtype == torch.float32 and (d.dim() == 1 or d.stride(1) == 1)
        assert d.numel() == r * c, (tuple(d.shape), r, c)
        entries.append((slab, nslabs, stride, off, r, c, ld, d))
    if _SLAB_BATCH:
        _SLAB_BATCH[-1].extend(entries)
    else:
        _flush_segments(entries)


def fused_mlp_bwd(xa, xb, W1, b1, W2, b2, gamma, gy, gxa, gxb, add_gy_to_gxa, hid, n_out, dst,
                  outer_jobs=None):
    """dst: dict with the gradient tensors to fill: dW1 (hid, k_in), db1 (hid,),
    dW2 (n_out, hid), db2 (n_out,), dgamma / dbeta (n_out,) (views allowed).
    outer_jobs: a list that receives the deferred first-layer weight gradient of the K = 128 node
    update as a problem of fused_lin_bwd_multi (instead of its own nlam_outer_bwd launch)."""
    B, rows = gy.B, gy.rows
    k_in = xa.cols + (xb.cols if xb is not None else 0)
    stride = lib.nlam_mlp_bwd_slab_stride(k_in, hid, n_out)
    nslabs = lib.nlam_bwd_grid(_ntiles(B, rows))
    dev = W1.device
    slab = torch.empty(nslabs * stride, dtype=torch.float32, device=dev)
    # K = 128 (node update [x_r | agg]): the first layer's weight gradient is formed by a
    # lean second pass (nlam_outer_bwd) from the stored hidden-pre-activation gradient
    defer = hid == 64 and k_in == 128 and gamma is not None and xb is not None
    ga = torch.empty(B, rows, hid, dtype=torch.float32, device=dev) if defer else None
    _launch(
        "nlam_mlp_bwd", lib.nlam_mlp_bwd,
        (xa.ptr, xa.bstride, xa.ld, xa.cols,
         xb.ptr if xb is not None else None, xb.bstride if xb is not None else 0,
         xb.ld if xb is not None else 0, xb.cols if xb is not None else 0,
         W1.data_ptr(), W1.stride(0), _p(b1), W2.data_ptr(), W2.stride(0), _p(b2), _p(gamma),
         gy.ptr, gy.bstride, gy.ld,
         gxa.ptr if gxa is not None else None, gxa.bstride if gxa is not None else 0,
         gxa.ld if gxa is not None else 0,
         gxb.ptr if gxb is not None else None, gxb.bstride if gxb is not None else 0,
         gxb.ld if gxb is not None else 0, int(add_gy_to_gxa),
         slab.data_ptr(), stride, _p(ga), B, rows, hid, n_out, stream()),
        flops=2.0 * B * rows * hid * ((2 if (gxa is not None or gxb is not None) else 1) * k_in
                                       + (0 if defer else k_in) + 4 * n_out),
        nbytes=4.0 * B * rows * (2 * k_in + n_out),
    )
    kp32 = (k_in + 31) // 32 * 32
    no32 = (n_out + 31) // 32 * 32
    o1 = hid * kp32
    o2 = o1 + hid
    ov = o2 + no32 * hid
    segs = [(o2, n_out, hid, hid, dst["dW2"]), (ov, 1, n_out, n_out, dst["db2"])]
    if gamma is not None:
        segs += [(ov + no32, 1, n_out, n_out, dst["dgamma"]),
                 (ov + 2 * no32, 1, n_out, n_out, dst["dbeta"])]
    if defer and outer_jobs is not None:
        outer_jobs.append({"x": xa, "xb": xb, "gy": mat(ga), "dW": dst["dW1"], "db": dst["db1"]})
    elif defer:
        fused_outer_bwd(mat(ga), xa, xb, None, dst["dW1"], dst["db1"])
    else:
        segs += [(0, hid, k_in, kp32, dst["dW1"]), (o1, 1, hid, hid, dst["db1"])]
    reduce_segments(slab, nslabs, stride, segs)


def fused_outer_bwd(g, xa, xb, x_index, dW_dst, db_dst):
    """dW_dst (ng, kx) = sum_rows g^T (x) [xa | xb] (x rows optionally gathered by
    x_index), db_dst (ng,) = colsum(g) (db_dst may be None)."""
    B, rows, ng = g.B, g.rows, g.cols
    kx = xa.cols + (xb.cols if xb is not None else 0)
    stride = lib.nlam_outer_bwd_slab_stride(ng, kx)
    nslabs = lib.nlam_bwd_grid(_ntiles(B, rows))
    dev = dW_dst.device
    slab = torch.empty(nslabs * stride, dtype=torch.float32, device=dev)
    _launch(
        "nlam_outer_bwd", lib.nlam_outer_bwd,
        (g.ptr, g.bstride, g.ld, ng, xa.ptr, xa.bstride, xa.ld, xa.cols,
         xb.ptr if xb is not None else None, xb.bstride if xb is not None else 0,
         xb.ld if xb is not None else 0, xb.cols if xb is not None else 0,
         _p(x_index), slab.data_ptr(), stride, B, rows, stream()),
        flops=2.0 * B * rows * ng * kx, nbytes=4.0 * B * rows * (ng + kx),
    )
    kx32 = (kx + 31) // 32 * 32
    reduce_segments(slab, nslabs, stride,
                    [(0, ng, kx, kx32, dW_dst), (ng * kx32, 1, ng, ng, db_dst)])


def lin_bwd_can_sum(x, gy):
    """True when nlam_lin_bwd may fold the batch sum of gy (B, rows, n) into its load
    (batch-invariant x, 16-byte aligned rows)."""
    if (x.cols + 31) // 32 != 2 or gy.cols != 64:
        return False
    return (x.B == 1 and gy.B > 1 and x.ptr % 16 == 0 and x.ld % 4 == 0 and x.cols % 4 == 0
            and gy.ptr % 16 == 0 and gy.ld % 4 == 0 and gy.cols % 4 == 0 and gy.bstride % 4 == 0)


def fused_lin_bwd(x, gy, WA, WB, gx, dWA, dbA, dWB, dbB, gx_add=None, sum_gy_batch=False):
    """gx = gy [WA; WB] [+ gx_add] (optional); dWA (nA, k), dbA (nA,), dWB, dbB: destination
    views (any of them None = not needed).  sum_gy_batch: x is (1, rows, k) and gy
    (B, rows, n) is summed over B while it is loaded."""
    nsum, sum_stride = 1, 0
    if sum_gy_batch:
        if not lin_bwd_can_sum(x, gy):
            raise NlamError("fused_lin_bwd: sum_gy_batch needs batch-invariant x and aligned rows")
        nsum, sum_stride = gy.B, gy.bstride
        gy = gy._replace(B=1, bstride=0)
    if gx_add is not None:
        if gx is None or (gx_add.B, gx_add.rows, gx_add.cols) != (gx.B, gx.rows, gx.cols):
            raise NlamError("fused_lin_bwd: gx_add must match gx")
    nA = WA.shape[0]
    nB = WB.shape[0] if WB is not None else 0
    B, rows, k_in = gy.B, gy.rows, x.cols
    stride = lib.nlam_lin_bwd_slab_stride(k_in, nA + nB)
    nslabs = lib.nlam_bwd_grid(_ntiles(B, rows))
    dev = WA.device
    slab = torch.empty(nslabs * stride, dtype=torch.float32, device=dev)
    _launch(
        "nlam_lin_bwd", lib.nlam_lin_bwd,
        (x.ptr, x.bstride, x.ld, k_in, gy.ptr, gy.bstride, gy.ld, WA.data_ptr(), WA.stride(0), nA,
         _p(WB), WB.stride(0) if WB is not None else 0, nB,
         gx.ptr if gx is not None else None, gx.bstride if gx is not None else 0,
         gx.ld if gx is not None else 0,
         gx_add.ptr if gx_add is not None else None,
         gx_add.bstride if gx_add is not None else 0, gx_add.ld if gx_add is not None else 0,
         nsum, sum_stride, slab.data_ptr(), stride, B, rows, stream()),
        flops=2.0 * B * rows * k_in * (nA + nB) * (2 if gx is not None else 1),
        nbytes=4.0 * B * rows * (k_in * (2 if gx is not None else 1) + nA + nB),
    )
    kp32 = (k_in + 31) // 32 * 32
    n = nA + nB
    segs = [(0, nA, k_in, kp32, dWA), (n * kp32, 1, nA, nA, dbA)]
    if nB:
        segs += [(nA * kp32, nB, k_in, kp32, dWB), (n * kp32 + nA, 1, nB, nB, dbB)]
    reduce_segments(slab, nslabs, stride, segs)


def mlp_multi_supported():
    """Multi-problem embedder MLP launches exist in this process's arithmetic mode."""
    return bool(lib.nlam_mlp_multi_supported())


def fused_mlp_fwd_multi(problems):
    """[(x Mat (k <= 32 wide), W1, b1, W2, b2, gamma, beta, out Mat)] -> ONE launch
    (make_mlp([k, 64, 64]) + LayerNorm each)."""
    n = len(problems)
    I64, I32, P = ctypes.c_int64 * n, ctypes.c_int32 * n, ctypes.c_void_p * n
    xs = [pr[0] for pr in problems]
    outs = [pr[7] for pr in problems]
    _launch(
        "nlam_mlp_fwd_multi", lib.nlam_mlp_fwd_multi,
        (n, P(*[x.ptr for x in xs]), I64(*[x.bstride for x in xs]), I64(*[x.ld for x in xs]),
         I32(*[x.cols for x in xs]), P(*[pr[1].data_ptr() for pr in problems]),
         I64(*[pr[1].stride(0) for pr in problems]), P(*[_p(pr[2]) for pr in problems]),
         P(*[pr[3].data_ptr() for pr in problems]), I64(*[pr[3].stride(0) for pr in problems]),
         P(*[_p(pr[4]) for pr in problems]), P(*[pr[5].data_ptr() for pr in problems]),
         P(*[pr[6].data_ptr() for pr in problems]), P(*[o.ptr for o in outs]),
         I64(*[o.bstride for o in outs]), I64(*[o.ld for o in outs]), I64(*[o.B for o in outs]),
         I64(*[o.rows for o in outs]), 64, 64, stream()),
        flops=sum(2.0 * o.B * o.rows * 64 * (x.cols + 64) for x, o in zip(xs, outs)),
        nbytes=sum(4.0 * o.B * o.rows * (x.cols + 64) for x, o in zip(xs, outs)),
    )


def fused_mlp_bwd_multi(problems):
    """[dict(x, W1, b1, W2, b2, gamma, gy (Mat), gx (Mat or None), dst=dict(dW1, db1, dW2, db2,
    dgamma, dbeta))] -> ONE launch + the slab reductions (batched by the caller's slab_batch)."""
    n = len(problems)
    I64, I32, P = ctypes.c_int64 * n, ctypes.c_int32 * n, ctypes.c_void_p * n
    slabs, nsl, strides = [], [], []
    for pr in problems:
        x, gy = pr["x"], pr["gy"]
        st = lib.nlam_mlp_bwd_slab_stride(x.cols, 64, 64)
        ns = lib.nlam_bwd_grid(_ntiles(gy.B, gy.rows))
        strides.append(st)
        nsl.append(ns)
        slabs.append(torch.empty(ns * st, dtype=torch.float32, device=pr["W1"].device))
    gxs = [pr.get("gx") for pr in problems]
    _launch(
        "nlam_mlp_bwd_multi", lib.nlam_mlp_bwd_multi,
        (n, P(*[pr["x"].ptr for pr in problems]), I64(*[pr["x"].bstride for pr in problems]),
         I64(*[pr["x"].ld for pr in problems]), I32(*[pr["x"].cols for pr in problems]),
         P(*[pr["W1"].data_ptr() for pr in problems]), I64(*[pr["W1"].stride(0) for pr in problems]),
         P(*[_p(pr["b1"]) for pr in problems]), P(*[pr["W2"].data_ptr() for pr in problems]),
         I64(*[pr["W2"].stride(0) for pr in problems]), P(*[_p(pr["b2"]) for pr in problems]),
         P(*[pr["gamma"].data_ptr() for pr in problems]),
         P(*[pr["gy"].ptr for pr in problems]), I64(*[pr["gy"].bstride for pr in problems]),
         I64(*[pr["gy"].ld for pr in problems]),
         P(*[g.ptr if g is not None else None for g in gxs]),
         I64(*[g.bstride if g is not None else 0 for g in gxs]),
         I64(*[g.ld if g is not None else 0 for g in gxs]),
         P(*[s_.data_ptr() for s_ in slabs]), I64(*strides),
         I64(*[pr["gy"].B for pr in problems]), I64(*[pr["gy"].rows for pr in problems]), 64, 64,
         stream()),
        flops=sum(4.0 * pr["gy"].B * pr["gy"].rows * 64 * (pr["x"].cols + 2 * 64) for pr in problems),
        nbytes=sum(4.0 * pr["gy"].B * pr["gy"].rows * (2 * pr["x"].cols + 64) for pr in problems),
    )
    for pr, slab, ns, st in zip(problems, slabs, nsl, strides):
        k_in = pr["x"].cols
        kp32 = (k_in + 31) // 32 * 32
        o1 = 64 * kp32
        o2 = o1 + 64
        ov = o2 + 64 * 64
        dst = pr["dst"]
        reduce_segments(slab, ns, st, [
            (0, 64, k_in, kp32, dst["dW1"]), (o1, 1, 64, 64, dst["db1"]),
            (o2, 64, 64, 64, dst["dW2"]), (ov, 1, 64, 64, dst["db2"]),
            (ov + 64, 1, 64, 64, dst["dgamma"]), (ov + 128, 1, 64, 64, dst["dbeta"])])


def lin_multi_supported():
    """Multi-problem projection launches exist at hidden 64 in this process's arithmetic mode."""
    return bool(lib.nlam_lin_multi_supported())


def fused_lin_fwd_multi(problems):
    """[(x Mat, W (64, 64) view, bias or None, out Mat)]: out_k = x_k W_k^T + bias_k, ONE launch."""
    n = len(problems)
    I64, P = ctypes.c_int64 * n, ctypes.c_void_p * n
    d = problems[0][1].shape[0]
    _launch(
        "nlam_lin_fwd_multi", lib.nlam_lin_fwd_multi,
        (n, d, P(*[x.ptr for x, _, _, _ in problems]), I64(*[x.bstride for x, _, _, _ in problems]),
         I64(*[x.ld for x, _, _, _ in problems]), P(*[W.data_ptr() for _, W, _, _ in problems]),
         I64(*[W.stride(0) for _, W, _, _ in problems]), P(*[_p(b) for _, _, b, _ in problems]),
         P(*[o.ptr for _, _, _, o in problems]), I64(*[o.bstride for _, _, _, o in problems]),
         I64(*[o.ld for _, _, _, o in problems]), I64(*[o.B for _, _, _, o in problems]),
         I64(*[o.rows for _, _, _, o in problems]), 0, stream()),
        flops=sum(2.0 * o.B * o.rows * d * d for _, _, _, o in problems),
        nbytes=sum(8.0 * o.B * o.rows * d for _, _, _, o in problems),
    )


def fused_lin_bwd_multi(problems):
    """Several projection backward passes (hidden 64) in ONE launch.  Each problem is a dict:
    x (Mat), W ((64, 64) view), dW (destination view), db (destination or None), gx (Mat or
    None), gx_add (Mat or None), and EITHER gy (Mat) OR gather = (gh Mat (B, M, 64), csc_colptr,
    csc_eid, n_send): gy rows formed as sums of gh rows over the sender lists.  nsum > 1: x is
    batch-invariant and the gradient is summed over nsum batch slices while it is read."""
    n = len(problems)
    I64, P = ctypes.c_int64 * n, ctypes.c_void_p * n
    d = 64
    rowsB, slabs, nslabs, strides = [], [], [], []
    for pr in problems:
        x = pr["x"]
        nsum = pr.get("nsum", 1)
        if nsum > 1:
            B = 1
        else:
            B = pr["gather"][0].B if pr.get("gather") is not None else pr["gy"].B
        rowsB.append((B, x.rows))
        ns = lib.nlam_bwd_grid(_ntiles(B, x.rows))
        nslabs.append(ns)
        # "xb" present: a deferred 64 x 128 first-layer weight gradient (no data gradient)
        stride = d * 2 * d + d if pr.get("xb") is not None else d * d + d
        strides.append(stride)
        slabs.append(torch.empty(ns * stride, dtype=torch.float32, device=pr["dW"].device))

    def opt(m, f):
        return [f(pr[m]) if pr.get(m) is not None else None for pr in problems]

    def opti(m, f):
        return [f(pr[m]) if pr.get(m) is not None else 0 for pr in problems]

    gath = [pr.get("gather") for pr in problems]
    sum_stride = []
    for pr, g in zip(problems, gath):
        if pr.get("nsum", 1) > 1:
            sum_stride.append(g[0].bstride if g is not None else pr["gy"].bstride)
        else:
            sum_stride.append(0)
    flops = sum(2.0 * max(B, pr.get("nsum", 1)) * r * d * d * (2 if pr.get("gx") is not None else 1)
                for (B, r), pr in zip(rowsB, problems))
    nbytes = sum(4.0 * max(B, pr.get("nsum", 1)) * r * d * 2 for (B, r), pr in zip(rowsB, problems))
    # (gathered rows: one per list entry -- edges, or the (tile, sender) pairs of the sender partials)
    nbytes += sum(4.0 * g[0].B * min(g[0].rows, g[2].numel()) * d for g in gath if g is not None)
    _launch(
        "nlam_lin_bwd_multi", lib.nlam_lin_bwd_multi,
        (n, d, P(*[pr["x"].ptr for pr in problems]), I64(*[pr["x"].bstride for pr in problems]),
         I64(*[pr["x"].ld for pr in problems]),
         P(*opt("xb", lambda m: m.ptr)), I64(*opti("xb", lambda m: m.bstride)),
         I64(*opti("xb", lambda m: m.ld)),
         P(*opt("gy", lambda m: m.ptr)),
         I64(*[0 if pr.get("nsum", 1) > 1 or pr.get("gy") is None else pr["gy"].bstride for pr in problems]),
         I64(*opti("gy", lambda m: m.ld)),
         P(*opt("W", lambda w: w.data_ptr())), I64(*opti("W", lambda w: w.stride(0))),
         P(*opt("gx", lambda m: m.ptr)), I64(*opti("gx", lambda m: m.bstride)),
         I64(*opti("gx", lambda m: m.ld)),
         P(*opt("gx_add", lambda m: m.ptr)), I64(*opti("gx_add", lambda m: m.bstride)),
         I64(*opti("gx_add", lambda m: m.ld)),
         I64(*[pr.get("nsum", 1) for pr in problems]), I64(*sum_stride),
         P(*[g[0].ptr if g is not None else None for g in gath]),
         I64(*[g[0].bstride if g is not None else 0 for g in gath]),
         P(*[g[1].data_ptr() if g is not None else None for g in gath]),
         P(*[g[2].data_ptr() if g is not None else None for g in gath]),
         I64(*[g[3] if g is not None else 0 for g in gath]),
         P(*[s_.data_ptr() for s_ in slabs]), I64(*strides),
         I64(*[B for B, _ in rowsB]), I64(*[r for _, r in rowsB]), stream()),
        flops=flops, nbytes=nbytes,
    )
    for pr, slab, ns, stride in zip(problems, slabs, nslabs, strides):
        k = 2 * d if pr.get("xb") is not None else d
        reduce_segments(slab, ns, stride, [(0, d, k, k, pr["dW"]), (d * k, 1, d, d, pr.get("db"))])


def node_chain_supported():
    """The fused node-side kernels of an InteractionNet chain exist in this process's arithmetic
    mode (split-bf16 products, hidden width 64)."""
    return bool(lib.nlam_node_chain_supported())


def node_fwd(x, agg, V1, c1, V2, c2, gamma, beta, xout, WA, bA, WB, bB, P):
    """xout = x + LN(V2 silu(V1 [x | agg] + c1) + c2) and (P not None) the next layer's
    projections P = [xout WA^T + bA | xout WB^T + bB] in one launch."""
    B, rows, d = xout.B, xout.rows, xout.cols
    _launch(
        "nlam_node_fwd", lib.nlam_node_fwd,
        (x.ptr, x.bstride, x.ld, agg.ptr, agg.bstride, agg.ld, V1.data_ptr(), V1.stride(0),
         c1.data_ptr(), V2.data_ptr(), V2.stride(0), c2.data_ptr(), gamma.data_ptr(),
         beta.data_ptr(), xout.ptr, xout.bstride, xout.ld,
         _p(WA), WA.stride(0) if WA is not None else 0, _p(bA),
         _p(WB), WB.stride(0) if WB is not None else 0, _p(bB),
         P.ptr if P is not None else None, P.bstride if P is not None else 0,
         P.ld if P is not None else 0, B, rows, stream()),
        flops=2.0 * B * rows * d * (3 * d + (2 * d if P is not None else 0)),
        nbytes=4.0 * B * rows * d * (3 + (2 if P is not None else 0)),
    )


def node_bwd(gh, csc_colptr, csc_eid, n_send, gP, g_res, WA, WB, upd, gx_out):
    """Backward data pass of one chain link (include/nlam_hip.h, nlam_node_bwd).  upd: None, or
    the node update below as a dict(x, agg, V1, c1, V2, c2, gamma, gagg_out, ga_out (tensor),
    dst=dict(dW2, db2, dgamma, dbeta) destination views)."""
    B, rows, d = gx_out.B, gx_out.rows, gx_out.cols
    M = gh.rows
    if upd is None:
        args_u = (None, 0, 0, None, 0, 0, None, 0, None, None, 0, None, None)
        out_u = (None, 0, 0, None, None, 0)
        slab = None
    else:
        x, agg = upd["x"], upd["agg"]
        V1, V2 = upd["V1"], upd["V2"]
        stride = lib.nlam_node_bwd_slab_stride()
        nslabs = lib.nlam_node_bwd_grid(B, rows)
        slab = torch.empty(nslabs * stride, dtype=torch.float32, device=V1.device)
        args_u = (x.ptr, x.bstride, x.ld, agg.ptr, agg.bstride, agg.ld, V1.data_ptr(), V1.stride(0),
                  upd["c1"].data_ptr(), V2.data_ptr(), V2.stride(0), upd["c2"].data_ptr(),
                  upd["gamma"].data_ptr())
        ga, go = upd["ga_out"], upd["gagg_out"]
        out_u = (go.ptr, go.bstride, go.ld, ga.data_ptr(), slab.data_ptr(), stride)
    _launch(
        "nlam_node_bwd", lib.nlam_node_bwd,
        (gh.ptr, gh.bstride, csc_colptr.data_ptr(), csc_eid.data_ptr(), n_send,
         gP.ptr, gP.bstride, gP.ld, g_res.ptr, g_res.bstride, g_res.ld,
         WA.data_ptr(), WA.stride(0), WB.data_ptr(), WB.stride(0)) + args_u
        + (gx_out.ptr, gx_out.bstride, gx_out.ld) + out_u + (B, rows, stream()),
        flops=2.0 * B * rows * d * (2 * d + (8 * d if upd is not None else 0)),
        nbytes=4.0 * B * (M * d + rows * d * (5 + (5 if upd is not None else 0))) + 4.0 * (M + rows),
    )
    if upd is not None:
        dst = upd["dst"]
        reduce_segments(slab, nslabs, stride, [
            (0, d, d, d, dst["dW2"]), (d * d, 1, d, d, dst["db2"]),
            (d * d + d, 1, d, d, dst["dgamma"]), (d * d + 2 * d, 1, d, d, dst["dbeta"])])


def node_outer(ga, xa, xb, gP, xl, dV1, dc1, dWA, dWB, dbB):
    """Weight gradients of one chain link in one launch: dV1 (d, 2d) = ga^T [xa | xb], dc1 =
    colsum ga (ga None: skipped); dWA / dWB (d, d) = gP[:, :d]^T xl / gP[:, d:]^T xl, dbB =
    colsum gP[:, d:] (the bias rides on the receiver projection)."""
    B, rows, d = xl.B, xl.rows, xl.cols
    stride = lib.nlam_node_outer_slab_stride()
    nslabs = lib.nlam_node_outer_grid(B, rows)
    slab = torch.empty(nslabs * stride, dtype=torch.float32, device=dWA.device)
    has_a = ga is not None
    _launch(
        "nlam_node_outer", lib.nlam_node_outer,
        ((ga.ptr if hasattr(ga, "ptr") else ga.data_ptr()) if has_a else None,
         xa.ptr if has_a else None, xa.bstride if has_a else 0, xa.ld if has_a else 0,
         xb.ptr if has_a else None, xb.bstride if has_a else 0, xb.ld if has_a else 0,
         gP.ptr, gP.bstride, gP.ld, xl.ptr, xl.bstride, xl.ld, slab.data_ptr(), stride, B, rows,
         stream()),
        flops=2.0 * B * rows * 2 * d * d * (2 if has_a else 1),
        nbytes=4.0 * B * rows * d * (3 + (3 if has_a else 0)),
    )
    o = 2 * d * d + d
    segs = [(o, d, d, d, dWA), (o + d * d, d, d, d, dWB), (o + 2 * d * d + d, 1, d, d, dbB)]
    if has_a:
        segs += [(0, d, 2 * d, 2 * d, dV1), (2 * d * d, 1, d, d, dc1)]
    reduce_segments(slab, nslabs, stride, segs)


def fused_edge_bwd_parts(g, pe, ps, pr, W2, b2, gamma, g_agg, gpart, gpr, dpe, mean, d, dW2, db2,
                         dgamma, dbeta):
    """nlam_edge_bwd_parts: no edge update, batch-invariant edge term pe (1, M, d); gpart
    (B, 16 ntiles, d) receives the per-tile sender partial sums of gh, dpe (1, M, d) its batch sum."""
    B = g_agg.B
    stride = lib.nlam_edge_bwd_slab_stride(d)
    nslabs = lib.nlam_bwd_grid(B * g.ntiles)
    slab = torch.empty(nslabs * stride, dtype=torch.float32, device=W2.device)
    _launch(
        "nlam_edge_bwd", lib.nlam_edge_bwd_parts,
        (g.tiles.data_ptr(), g.ntiles, g.csr_rowptr.data_ptr(), g.csr_eid.data_ptr(),
         g.csr_send.data_ptr(), g.csr_rec.data_ptr(), g.inv_deg.data_ptr() if mean else None,
         pe.ptr, pe.ld, ps.ptr, ps.bstride, ps.ld, pr.ptr, pr.bstride, pr.ld, W2.data_ptr(),
         W2.stride(0), b2.data_ptr(), gamma.data_ptr(), g_agg.ptr, g_agg.bstride, g_agg.ld,
         g.part_slot.data_ptr(), gpart.ptr, gpart.bstride, gpr.ptr, gpr.bstride, gpr.ld,
         dpe.ptr, dpe.ld, slab.data_ptr(), stride, B, d, stream()),
        flops=2.0 * B * g.M * d * d * 3,
        # algorithmic bytes: pe once, the sender partials (one row per (tile, sender) pair and
        # sample) and dpe out; node-side rows (ps, pr, g_agg, gpr) once; indices
        nbytes=4.0 * d * (2 * g.M + B * g.n_sender_parts + B * (ps.rows + 3 * pr.rows)) + 20.0 * g.M,
    )
    dd = d * d
    reduce_segments(slab, nslabs, stride, [(dd, d, d, d, dW2), (2 * dd, 1, d, d, db2),
                                           (2 * dd + d, 1, d, d, dgamma), (2 * dd + 2 * d, 1, d, d, dbeta)])


def fused_edge_bwd(g, e, has_egemm, ps, pr, W1e, W2, b2, gamma, g_agg, g_eout, gh_out, gpr, g_e,
                   mean, d, dW1e, dW2, db2, dgamma, dbeta):
    """dW1e (d, d) (has_egemm), dW2, db2, dgamma, dbeta: destination views."""
    B = g_agg.B
    stride = lib.nlam_edge_bwd_slab_stride(d)
    nslabs = lib.nlam_bwd_grid(B * g.ntiles)
    dev = W2.device
    slab = torch.empty(nslabs * stride, dtype=torch.float32, device=dev)
    units = 6 if has_egemm else 3
    _launch(
        "nlam_edge_bwd", lib.nlam_edge_bwd,
        (g.tiles.data_ptr(), g.ntiles, g.csr_rowptr.data_ptr(), g.csr_eid.data_ptr(),
         g.csr_send.data_ptr(), g.csr_rec.data_ptr(), g.inv_deg.data_ptr() if mean else None,
         e.ptr, e.bstride, e.ld, int(has_egemm), ps.ptr, ps.bstride, ps.ld, pr.ptr, pr.bstride,
         pr.ld, _p(W1e), W1e.stride(0) if W1e is not None else 0, W2.data_ptr(), W2.stride(0),
         b2.data_ptr(), gamma.data_ptr(), g_agg.ptr, g_agg.bstride, g_agg.ld,
         g_eout.ptr if g_eout is not None else None,
         g_eout.bstride if g_eout is not None else 0, g_eout.ld if g_eout is not None else 0,
         gh_out.ptr, gh_out.bstride, gpr.ptr, gpr.bstride, gpr.ld,
         g_e.ptr if g_e is not None else None, g_e.bstride if g_e is not None else 0,
         g_e.ld if g_e is not None else 0, slab.data_ptr(), stride, B, d, stream()),
        flops=2.0 * B * g.M * d * d * units,
        # algorithmic bytes: read e, g_e', write g_e (has_egemm) / gh;
        # node-side rows (ps, pr, g_agg, gpr) once; indices
        nbytes=4.0 * d * ((B if e.bstride else 1) * g.M
                          + (2 * B * g.M if has_egemm else 0) + B * g.M
                          + B * (ps.rows + 3 * pr.rows)) + 16.0 * g.M,
    )
    dd = d * d
    segs = [(dd, d, d, d, dW2), (2 * dd, 1, d, d, db2), (2 * dd + d, 1, d, d, dgamma),
            (2 * dd + 2 * d, 1, d, d, dbeta)]
    if has_egemm:
        segs.append((0, d, d, d, dW1e))
    reduce_segments(slab, nslabs, stride, segs)
