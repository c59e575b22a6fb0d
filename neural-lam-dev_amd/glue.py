"""Device-side rollout glue of the training path as HIP kernels with autograd:
state residual (base_graph_model.py:174-177), boundary overwrite
(ar_model.py:244-247) and the masked wmse/mse training loss (metrics.py:21-108,
ar_model.py:294-298).  They replace ~15 eager elementwise / indexing launches
per AR step, including a boolean-mask gather that synchronises the host."""
import os

import torch

from . import ops
from ._lib import lib


_TEE = os.environ.get("NLAM_TEE", "1") != "0"   # 0: plain autograd sums (A/B switch)


def _graph_task():
    """Id of the backward pass that is running (-1 outside one): a value left in a slot by an
    EARLIER pass (a pass that pruned the Tee node: retain_graph + autograd.grad on a sub-graph)
    must never be folded into this one."""
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


class GradSlot:
    """Side channel of a Tee: the first consumer's backward leaves its input gradient here
    (`put`), the second consumer's backward kernel adds it to its own output (`gx_add`) and
    records WHICH tensor it folded in (`take`)."""

    __slots__ = ("value", "task", "folded")

    def __init__(self):
        self.value, self.task, self.folded = None, -1, None

    def put(self, v):
        self.value, self.task = v, _graph_task()

    def take(self, shape):
        """The addend for a gradient of `shape`, or None; the slot remembers what was handed out."""
        v = self.value
        if v is None or self.task != _graph_task():
            return None
        if v.shape != shape or not v.is_contiguous() or v.dtype != torch.float32:
            return None
        self.value, self.folded = None, v
        return v


class Tee(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers whose backward kernels can fold the other
    branch's gradient into their own store (fused.py reads `_nlam_grad_sink` off the inputs).
    grid_emb feeds the g2m InteractionNet (sender side) and the grid's own encoding MLP
    (base_graph_model.py:134-141): autograd summed the two gradients with a full-size
    elementwise add, the one torch kernel left inside a GraphLAM step (31.6 us of 2.6 ms).
    The result is the sum whatever the execution order and however many consumers the `give`
    alias has: `ga` contains exactly the tensor `slot.folded` (if any); whatever else arrived on
    the give alias -- `gb` is autograd's sum over ALL its consumers -- is added here."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        slot = ctx.slot
        folded = slot.folded
        slot.value, slot.folded, slot.task = None, None, -1
        if ga is None:
            return gb, None
        if gb is None or gb is folded or (
                folded is not None and gb.data_ptr() == folded.data_ptr()
                and gb.shape == folded.shape and gb.stride() == folded.stride()):
            return ga, None   # the give alias's only gradient is already inside ga
        if folded is None:
            return ga + gb, None
        return ga + (gb - folded), None   # a second consumer of the give alias


def tee(x):
    """Two aliases of x (first: the consumer that runs its backward LAST and takes the other's
    gradient as an addend; second: the consumer whose backward runs first)."""
    if not (_TEE and x.is_cuda and x.requires_grad and torch.is_grad_enabled()):
        return x, x
    slot = GradSlot()
    a, b = Tee.apply(x, slot)
    a._nlam_grad_sink, b._nlam_grad_sink = ("take", slot), ("give", slot)
    return a, b


class StateResidual(torch.autograd.Function):
    """prev_state + net_out * diff_std + diff_mean."""

    @staticmethod
    def forward(ctx, prev_state, net_out, diff_std, diff_mean):
        prev_state, net_out = prev_state.contiguous(), net_out.contiguous()
        ops._require_dev(net_out, "net_out")
        y = torch.empty_like(net_out)
        F = net_out.shape[-1]
        ops._launch("nlam_affine_residual", lib.nlam_affine_residual,
                    (prev_state.data_ptr(), net_out.data_ptr(), diff_std.data_ptr(),
                     diff_mean.data_ptr(), y.data_ptr(), net_out.numel() // F, F, ops.stream()),
                    nbytes=12.0 * net_out.numel())
        ctx.save_for_backward(diff_std)
        return y

    @staticmethod
    def backward(ctx, gy):
        (diff_std,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[1]:
            gx = torch.empty_like(gy)
            F = gy.shape[-1]
            ops._launch("nlam_scale_cols", lib.nlam_scale_cols,
                        (gy.data_ptr(), diff_std.data_ptr(), gx.data_ptr(), gy.numel() // F, F,
                         ops.stream()), nbytes=8.0 * gy.numel())
        return (gy if ctx.needs_input_grad[0] else None), gx, None, None


class BoundaryMix(torch.autograd.Function):
    """boundary_mask * true_state + interior_mask * pred_state (mask (N,1) in {0,1})."""

    @staticmethod
    def forward(ctx, pred, truth, mask):
        pred, truth = pred.contiguous(), truth.contiguous()
        ops._require_dev(pred, "pred")
        B, N, F = pred.shape
        out = torch.empty_like(pred)
        ops._launch("nlam_boundary_mix", lib.nlam_boundary_mix,
                    (pred.data_ptr(), truth.data_ptr(), mask.data_ptr(), out.data_ptr(), B, N, F,
                     ops.stream()), nbytes=12.0 * pred.numel())
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        g = g.contiguous()
        B, N, F = g.shape
        gp = torch.empty_like(g)
        ops._launch("nlam_boundary_mix", lib.nlam_boundary_mix,
                    (g.data_ptr(), None, mask.data_ptr(), gp.data_ptr(), B, N, F, ops.stream()),
                    nbytes=8.0 * g.numel())
        return gp, None, None


def _batch_view(t):
    """(pointer, batch pitch) of a (B, N, F) tensor whose (N, F) items are contiguous -- e.g. a
    slice init_states[:, 1] of the batch tensor -- or of a contiguous copy otherwise."""
    B, N, F = t.shape
    if not (t.stride(2) == 1 and t.stride(1) == F and (B == 1 or t.stride(0) >= N * F)):
        t = t.contiguous()
    return t, (t.stride(0) if B > 1 else N * F)


class StateStep(torch.autograd.Function):
    """StateResidual followed by BoundaryMix in one pass (one launch each way instead of two, and
    no contiguous copies of the sliced batch tensors):
    boundary_mask * truth + interior_mask * (prev_state + net_out * diff_std + diff_mean)."""

    @staticmethod
    def forward(ctx, prev_state, net_out, truth, mask, diff_std, diff_mean):
        net_out = net_out.contiguous()
        ops._require_dev(net_out, "net_out")
        B, N, F = net_out.shape
        prev_state, pb = _batch_view(prev_state)
        truth, tb = _batch_view(truth)
        out = torch.empty_like(net_out)
        ops._launch("nlam_state_step", lib.nlam_state_step,
                    (prev_state.data_ptr(), pb, net_out.data_ptr(), truth.data_ptr(), tb,
                     mask.data_ptr(), diff_std.data_ptr(), diff_mean.data_ptr(), out.data_ptr(),
                     B, N, F, ops.stream()), nbytes=16.0 * net_out.numel())
        ctx.save_for_backward(mask, diff_std)
        return out

    @staticmethod
    def backward(ctx, g):
        mask, diff_std = ctx.saved_tensors
        g = g.contiguous()
        B, N, F = g.shape
        gx = torch.empty_like(g)
        gprev = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        ops._launch("nlam_state_step_bwd", lib.nlam_state_step_bwd,
                    (g.data_ptr(), mask.data_ptr(), diff_std.data_ptr(), gx.data_ptr(),
                     gprev.data_ptr() if gprev is not None else None, B, N, F, ops.stream()),
                    nbytes=(8.0 if gprev is None else 12.0) * g.numel())
        return gprev, gx, None, None, None, None


class DeferGrad(torch.autograd.Function):
    """Identity on a set of parameters whose backward runs ops.flush_deferred() before handing the
    gradients on (see ops.slab_batch).  Created once per predict_step, before any layer runs."""

    @staticmethod
    def forward(ctx, *ws):
        ctx.set_materialize_grads(False)   # (a parameter no layer used keeps a None gradient)
        return tuple(w.view_as(w) for w in ws)

    @staticmethod
    def backward(ctx, *gs):
        ops.flush_deferred()
        return gs


class StateStepLoss(torch.autograd.Function):
    """StateStep AND the training-loss term of the same AR step in one pass each way (the loss
    target of AR step t is the boundary truth of step t: ar_model.py:244-247 with 294-298):
    returns (new_state, lscale * sum keep w (new_state - truth)^2).  Forward reads what StateStep
    reads and nothing more; backward folds MaskedWMSE's and StateStep's into one kernel."""

    @staticmethod
    def forward(ctx, prev_state, net_out, truth, mask, diff_std, diff_mean, keep, w, lscale):
        net_out = net_out.contiguous()
        ops._require_dev(net_out, "net_out")
        B, N, F = net_out.shape
        prev_state, pb = _batch_view(prev_state)
        truth, tb = _batch_view(truth)
        out = torch.empty_like(net_out)
        partial = torch.empty(lib.nlam_state_step_wmse_blocks(), dtype=torch.float32, device=out.device)
        loss = torch.empty(1, dtype=torch.float32, device=out.device)
        ops._launch("nlam_state_step_wmse_fwd", lib.nlam_state_step_wmse_fwd,
                    (prev_state.data_ptr(), pb, net_out.data_ptr(), truth.data_ptr(), tb,
                     mask.data_ptr(), diff_std.data_ptr(), diff_mean.data_ptr(), keep.data_ptr(),
                     w.data_ptr(), out.data_ptr(), partial.data_ptr(), loss.data_ptr(), lscale,
                     B, N, F, ops.stream()), nbytes=16.0 * net_out.numel())
        ctx.save_for_backward(out, truth, mask, diff_std, keep, w)
        ctx.tb, ctx.lscale = tb, lscale
        ctx.set_materialize_grads(False)
        return out, loss.reshape(())

    @staticmethod
    def backward(ctx, g_state, g_loss):
        out, truth, mask, diff_std, keep, w = ctx.saved_tensors
        B, N, F = out.shape
        g_state = g_state.contiguous() if g_state is not None else None
        gl = g_loss.reshape(1).contiguous() if g_loss is not None else None
        gx = torch.empty_like(out)
        gprev = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        ops._launch("nlam_state_step_wmse_bwd", lib.nlam_state_step_wmse_bwd,
                    (out.data_ptr(), truth.data_ptr(), ctx.tb, mask.data_ptr(), diff_std.data_ptr(),
                     keep.data_ptr(), w.data_ptr(), gl.data_ptr() if gl is not None else None,
                     ctx.lscale, g_state.data_ptr() if g_state is not None else None, gx.data_ptr(),
                     gprev.data_ptr() if gprev is not None else None, B, N, F, ops.stream()),
                    nbytes=(12.0 + 4.0 * (g_state is not None) + 4.0 * (gprev is not None)) * out.numel())
        return gprev, gx, None, None, None, None, None, None, None


class MaskedWMSE(torch.autograd.Function):
    """mean over the leading dims of sum_f mean_{kept n} (pred - target)^2 * w_f."""

    @staticmethod
    def forward(ctx, pred, target, keep, w, scale):
        pred, target = pred.contiguous(), target.contiguous()
        ops._require_dev(pred, "pred")
        N, F = pred.shape[-2], pred.shape[-1]
        rows = pred.numel() // F
        partial = torch.empty(lib.nlam_wmse_blocks(), dtype=torch.float32, device=pred.device)
        out = torch.empty(1, dtype=torch.float32, device=pred.device)
        ops._launch("nlam_wmse_fwd", lib.nlam_wmse_fwd,
                    (pred.data_ptr(), target.data_ptr(), keep.data_ptr(), w.data_ptr(),
                     partial.data_ptr(), out.data_ptr(), rows, N, F, scale, ops.stream()),
                    nbytes=8.0 * pred.numel())
        ctx.save_for_backward(pred, target, keep, w)
        ctx.scale = scale
        return out.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        pred, target, keep, w = ctx.saved_tensors
        N, F = pred.shape[-2], pred.shape[-1]
        g = torch.empty_like(pred)
        gl = gloss.reshape(1).contiguous()
        ops._launch("nlam_wmse_bwd", lib.nlam_wmse_bwd,
                    (pred.data_ptr(), target.data_ptr(), keep.data_ptr(), w.data_ptr(),
                     gl.data_ptr(), ctx.scale, g.data_ptr(), pred.numel() // F, N, F,
                     ops.stream()), nbytes=12.0 * pred.numel())
        return g, None, None, None, None


class StdHead(torch.autograd.Function):
    """output_std branch of predict_step (base_graph_model.py:161-177):
    (prev_state, net_out (.., 2F)) -> (prev + net_out[..., :F] * diff_std + diff_mean,
    softplus(net_out[..., F:])) in one kernel each way."""

    @staticmethod
    def forward(ctx, prev_state, net_out, diff_std, diff_mean):
        prev_state, net_out = prev_state.contiguous(), net_out.contiguous()
        ops._require_dev(net_out, "net_out")
        F = net_out.shape[-1] // 2
        state = torch.empty_like(prev_state)
        std = torch.empty_like(prev_state)
        rows = prev_state.numel() // F
        ops._launch("nlam_std_head_fwd", lib.nlam_std_head_fwd,
                    (net_out.data_ptr(), prev_state.data_ptr(), diff_std.data_ptr(),
                     diff_mean.data_ptr(), state.data_ptr(), std.data_ptr(), rows, F,
                     ops.stream()), nbytes=20.0 * prev_state.numel())
        ctx.save_for_backward(net_out, diff_std)
        ctx.set_materialize_grads(False)
        return state, std

    @staticmethod
    def backward(ctx, g_state, g_std):
        net_out, diff_std = ctx.saved_tensors
        F = net_out.shape[-1] // 2
        g_state = g_state.contiguous() if g_state is not None else None
        g_std = g_std.contiguous() if g_std is not None else None
        g_out = torch.empty_like(net_out)
        ops._launch("nlam_std_head_bwd", lib.nlam_std_head_bwd,
                    (net_out.data_ptr(), g_state.data_ptr() if g_state is not None else None,
                     g_std.data_ptr() if g_std is not None else None, diff_std.data_ptr(),
                     g_out.data_ptr(), net_out.numel() // (2 * F), F, ops.stream()),
                    nbytes=8.0 * net_out.numel())
        return (g_state if ctx.needs_input_grad[0] else None), g_out, None, None


class MaskedNLL(torch.autograd.Function):
    """mean over the leading dims of sum_f mean_{kept n} -log N(target; pred, std^2)
    (metrics.py:166-190 under ar_model.py:294-298)."""

    @staticmethod
    def forward(ctx, pred, target, pred_std, keep, scale):
        pred, target, pred_std = pred.contiguous(), target.contiguous(), pred_std.contiguous()
        ops._require_dev(pred, "pred")
        N, F = pred.shape[-2], pred.shape[-1]
        rows = pred.numel() // F
        partial = torch.empty(lib.nlam_wmse_blocks(), dtype=torch.float32, device=pred.device)
        out = torch.empty(1, dtype=torch.float32, device=pred.device)
        ops._launch("nlam_nll_fwd", lib.nlam_nll_fwd,
                    (pred.data_ptr(), target.data_ptr(), pred_std.data_ptr(), keep.data_ptr(),
                     partial.data_ptr(), out.data_ptr(), rows, N, F, scale, ops.stream()),
                    nbytes=12.0 * pred.numel())
        ctx.save_for_backward(pred, target, pred_std, keep)
        ctx.scale = scale
        return out.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        pred, target, pred_std, keep = ctx.saved_tensors
        N, F = pred.shape[-2], pred.shape[-1]
        g, gs = torch.empty_like(pred), torch.empty_like(pred)
        gl = gloss.reshape(1).contiguous()
        ops._launch("nlam_nll_bwd", lib.nlam_nll_bwd,
                    (pred.data_ptr(), target.data_ptr(), pred_std.data_ptr(), keep.data_ptr(),
                     gl.data_ptr(), ctx.scale, g.data_ptr(), gs.data_ptr(), pred.numel() // F, N,
                     F, ops.stream()), nbytes=20.0 * pred.numel())
        return g, None, gs, None, None


class ConcatRows(torch.autograd.Function):
    """torch.cat(sources, dim=-1) of (B | 1, N, w_k) grid feature tensors in one kernel
    (base_graph_model.py:116-124); a stride-0 expand (expand_to_batch) is read in place."""

    @staticmethod
    def forward(ctx, *srcs):
        import ctypes

        from .fused import _base

        base = [_base(t.detach()) for t in srcs]
        B = max(t.shape[0] for t in srcs)
        N = srcs[0].shape[-2]
        mats = [ops.mat(t) for t in base]
        W = sum(m.cols for m in mats)
        from . import fused

        out = fused._pre_take(("concat",), srcs[0])   # written by fused.grid_encode in this step
        if out is not None and tuple(out.shape) != (B, N, W):
            out = None
        if out is None:
            out = torch.empty(B, N, W, dtype=torch.float32, device=srcs[0].device)
            n = len(mats)
            ops._launch(
                "nlam_concat_rows", lib.nlam_concat_rows,
                (n, (ctypes.c_void_p * n)(*[m.ptr for m in mats]),
                 (ctypes.c_int64 * n)(*[m.bstride for m in mats]),
                 (ctypes.c_int64 * n)(*[m.ld for m in mats]),
                 (ctypes.c_int32 * n)(*[m.cols for m in mats]), out.data_ptr(), B, N, ops.stream()),
                nbytes=8.0 * out.numel())
        ctx.widths = [m.cols for m in mats]
        return out

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for k, w in enumerate(ctx.widths):
            if ctx.needs_input_grad[k]:
                outs.append(g[..., o : o + w])   # (autograd's expand backward sums over B)
            else:
                outs.append(None)
            o += w
        return tuple(outs)


class SumMany(torch.autograd.Function):
    """a_0 + a_1 + ... (up to 8 equally shaped tensors per launch, fixed order) in one pass; the
    gradient of every term is the incoming gradient itself."""

    @staticmethod
    def forward(ctx, *terms):
        import ctypes

        ops._require_dev(terms[0], "term")
        terms = [t.contiguous() for t in terms]
        out = torch.empty_like(terms[0])
        acc = None
        for i in range(0, len(terms), 7 if len(terms) > 8 else 8):
            part = ([acc] if acc is not None else []) + terms[i : i + (7 if len(terms) > 8 else 8)]
            arr = (ctypes.c_void_p * len(part))(*[t.data_ptr() for t in part])
            ops._launch("nlam_sum_many", lib.nlam_sum_many,
                        (len(part), arr, out.data_ptr(), out.numel(), ops.stream()),
                        nbytes=4.0 * out.numel() * (len(part) + 1))
            acc = out
        ctx.n = len(terms)
        return out

    @staticmethod
    def backward(ctx, g):
        return (g,) * ctx.n


def sum_many(terms):
    terms = list(terms)
    if len(terms) == 1:
        return terms[0]
    return SumMany.apply(*terms)
