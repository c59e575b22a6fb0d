"""Shape-generic InteractionNet / MLP path (any hidden_dim, any MLP depth,
sum/mean, SplitMLPs), composed from the generic kernels of libnlam_hip.so.

This is the path every configuration reachable through the reference's public
constructors can take (interaction_net.py:19-84, utils.py:191-214); the fused
kernels (fused.py) take over for the shapes the BASELINE configs use.  Forward
and backward are explicit kernel sequences wrapped in torch.autograd.Function
(torch only allocates and tracks the graph).
"""
import torch

from . import ops
from .ops import mat


def _empty(B, rows, cols, device):
    return torch.empty(B, rows, cols, dtype=torch.float32, device=device)


# --------------------------------------------------------------------- MLP
def mlp_forward(x, weights, biases, ln, res, device):
    """x: Mat (B, rows, in).  weights[i]: (out_i, in_i).  ln: (gamma, beta) or
    None.  res: Mat or None (added to the output).  Returns (y tensor, saved)."""
    B, rows = x.B, x.rows
    n = len(weights)
    h = x
    pre, act = [], []
    for i, (W, b) in enumerate(zip(weights, biases)):
        a = _empty(B, rows, W.shape[0], device)
        ops.linear_fwd(h, W, b, mat(a))
        pre.append(a)
        if i != n - 1:
            s = torch.empty_like(a)
            ops.silu_fwd(a, s)
            act.append(s)
            h = mat(s)
    z = pre[-1]
    if ln is not None:
        y = torch.empty_like(z)
        ops.layernorm_fwd(mat(z), ln[0], ln[1], res, mat(y))
    elif res is not None:
        y = torch.empty_like(z)
        ops.add_rows(mat(z), res, mat(y))
    else:
        y = z
    return y, (pre, act)


def mlp_backward(gy, x, weights, ln, saved, need_gx, device):
    """gy: Mat (B, rows, out).  Returns (gx tensor or None, dWs, dbs, dln)."""
    pre, act = saved
    n = len(weights)
    B, rows = x.B, x.rows
    dWs, dbs = [None] * n, [None] * n
    dln = None
    if ln is not None:
        gz_t = torch.empty_like(pre[-1])
        dg, dbt = torch.empty_like(ln[0]), torch.empty_like(ln[1])
        ops.layernorm_bwd(mat(pre[-1]), ln[0], gy, mat(gz_t), dg, dbt)
        dln = (dg, dbt)
        gz = mat(gz_t)
    else:
        gz = gy
    gx = None
    for i in range(n - 1, -1, -1):
        W = weights[i]
        h_in = x if i == 0 else mat(act[i - 1])
        dWs[i] = torch.empty_like(W)
        dbs[i] = torch.empty(W.shape[0], dtype=torch.float32, device=device)
        ops.linear_bwd_weight(gz, h_in, dWs[i], dbs[i])
        if i == 0 and not need_gx:
            break
        gh = _empty(B, rows, W.shape[1], device)
        ops.linear_bwd_data(gz, W, mat(gh))
        if i == 0:
            gx = gh
        else:
            ga = torch.empty_like(gh)
            ops.silu_bwd(pre[i - 1], gh, ga)
            gz = mat(ga)
    return gx, dWs, dbs, dln


class MLPFunction(torch.autograd.Function):
    """y = [res +] [LN](W_n silu(... silu(W_0 x + b_0) ...) + b_n)."""

    @staticmethod
    def forward(ctx, x, res, n_layers, has_ln, *params):
        ctx.tag = ops._TAG[-1] if ops._TAG else "mlp"
        weights = list(params[0 : 2 * n_layers : 2])
        biases = list(params[1 : 2 * n_layers : 2])
        ln = (params[2 * n_layers], params[2 * n_layers + 1]) if has_ln else None
        xm = mat(x.detach())
        rm = mat(res.detach()) if res is not None else None
        y, saved = mlp_forward(xm, weights, biases, ln, rm, x.device)
        ctx.n_layers, ctx.has_ln = n_layers, has_ln
        ctx.x_shape = x.shape
        ctx.has_res = res is not None
        ctx.saved_bufs = saved
        ctx.xm = xm
        ctx.save_for_backward(*params)
        return y.reshape(*x.shape[:-1], weights[-1].shape[0])

    @staticmethod
    def backward(ctx, gy):
        params = ctx.saved_tensors
        n = ctx.n_layers
        weights = list(params[0 : 2 * n : 2])
        ln = (params[2 * n], params[2 * n + 1]) if ctx.has_ln else None
        gy = gy.contiguous()
        need_gx = ctx.needs_input_grad[0]
        with ops.tag(ctx.tag):
            gx, dWs, dbs, dln = mlp_backward(
                mat(gy), ctx.xm, weights, ln, ctx.saved_bufs, need_gx, gy.device
            )
        ctx.saved_bufs = None
        grads = []
        for dW, db in zip(dWs, dbs):
            grads += [dW, db]
        if dln is not None:
            grads += [dln[0], dln[1]]
        gx = gx.reshape(ctx.x_shape) if gx is not None else None
        gres = gy if ctx.has_res else None
        return (gx, gres, None, None) + tuple(grads)


def mlp_params(seq):
    """(weights, biases, ln) from a make_mlp-style Sequential (Linear at even
    indices, LayerNorm last)."""
    lin = [m for m in seq if isinstance(m, torch.nn.Linear)]
    lns = [m for m in seq if isinstance(m, torch.nn.LayerNorm)]
    flat = []
    for m in lin:
        flat += [m.weight, m.bias]
    if lns:
        flat += [lns[0].weight, lns[0].bias]
    return len(lin), bool(lns), flat


def apply_mlp(seq, x, res=None):
    n, has_ln, flat = mlp_params(seq)
    return MLPFunction.apply(x, res, n, has_ln, *flat)


# ---------------------------------------------------------- InteractionNet
class InteractionNetGenericFunction(torch.autograd.Function):
    """interaction_net.py:86-131 as an explicit kernel sequence.

    Inputs (send_rep, rec_rep, edge_rep) are (B, N, d) device tensors, any batch
    stride (stride-0 expands included).  `edge_blocks` / `aggr_blocks` are lists
    of (row_start, row_end, n_layers, has_ln, param_offset) describing one MLP
    (no chunking) or one per SplitMLPs chunk; all parameters are in *params."""

    @staticmethod
    def forward(ctx, send_rep, rec_rep, edge_rep, g, update_edges, mean, edge_blocks,
                aggr_blocks, *params):
        with ops.tag(g.tag):
            return InteractionNetGenericFunction._forward(
                ctx, send_rep, rec_rep, edge_rep, g, update_edges, mean, edge_blocks,
                aggr_blocks, *params)

    @staticmethod
    def backward(ctx, g_rec_out, g_edge_out=None):
        with ops.tag(ctx.g.tag):
            return InteractionNetGenericFunction._backward(ctx, g_rec_out, g_edge_out)

    @staticmethod
    def _forward(ctx, send_rep, rec_rep, edge_rep, g, update_edges, mean, edge_blocks,
                 aggr_blocks, *params):
        dev = edge_rep.device
        B = max(send_rep.shape[0], rec_rep.shape[0], edge_rep.shape[0])
        M, d = edge_rep.shape[-2], edge_rep.shape[-1]
        n_rec = rec_rep.shape[-2]
        sm, rm, em = mat(send_rep.detach()), mat(rec_rep.detach()), mat(edge_rep.detach())

        cat = _empty(B, M, 3 * d, dev)
        ops.copy_rows(em, mat(cat, 0, d))
        ops.gather_rows(sm, g.send, mat(cat, d, d))
        ops.gather_rows(rm, g.rec, mat(cat, 2 * d, d))

        def run_blocks(blocks, xt, res_t):
            """apply per-chunk MLPs on row ranges of xt (B, rows, in)."""
            out = None
            saved = []
            for (r0, r1, n, has_ln, off) in blocks:
                w = list(params[off : off + 2 * n : 2])
                bb = list(params[off + 1 : off + 2 * n : 2])
                ln = (params[off + 2 * n], params[off + 2 * n + 1]) if has_ln else None
                xs = mat(xt[:, r0:r1])
                rs = mat(res_t[:, r0:r1, : w[-1].shape[0]]) if res_t is not None else None
                y, sv = mlp_forward(xs, w, bb, ln, rs, dev)
                saved.append(sv)
                if len(blocks) == 1:
                    out = y
                else:
                    if out is None:
                        out = _empty(B, xt.shape[1], y.shape[-1], dev)
                    ops.copy_rows(mat(y), mat(out[:, r0:r1]))
            return out, saved

        msg, saved_e = run_blocks(edge_blocks, cat, None)
        cat2 = _empty(B, n_rec, 2 * d, dev)
        ops.copy_rows(rm, mat(cat2, 0, d))
        ops.segment_sum(mat(msg), g.csr_rowptr, g.csr_eid, mat(cat2, d, d),
                        scale=g.inv_deg if mean else None)
        rec_out, saved_a = run_blocks(aggr_blocks, cat2, cat2)
        if update_edges:
            edge_out = torch.empty_like(msg)
            ops.add_rows(mat(cat, 0, d), mat(msg), mat(edge_out))
        else:
            edge_out = None

        ctx.g, ctx.update_edges, ctx.mean = g, update_edges, mean
        ctx.edge_blocks, ctx.aggr_blocks = edge_blocks, aggr_blocks
        ctx.bufs = (cat, cat2, saved_e, saved_a)
        ctx.dims = (B, M, d, n_rec, send_rep.shape[-2])
        ctx.in_shapes = (send_rep.shape, rec_rep.shape, edge_rep.shape)
        ctx.save_for_backward(*params)
        if update_edges:
            return rec_out, edge_out
        return rec_out

    @staticmethod
    def _backward(ctx, g_rec_out, g_edge_out=None):
        params = ctx.saved_tensors
        g = ctx.g
        cat, cat2, saved_e, saved_a = ctx.bufs
        ctx.bufs = None
        B, M, d, n_rec, n_send = ctx.dims
        dev = cat.device
        grads = [None] * len(params)

        def back_blocks(blocks, saved, xt, gy_t, width):
            gx_full = _empty(B, xt.shape[1], width, dev)
            for (r0, r1, n, has_ln, off), sv in zip(blocks, saved):
                w = list(params[off : off + 2 * n : 2])
                ln = (params[off + 2 * n], params[off + 2 * n + 1]) if has_ln else None
                gx, dWs, dbs, dln = mlp_backward(
                    mat(gy_t[:, r0:r1]), mat(xt[:, r0:r1]), w, ln, sv, True, dev
                )
                if len(blocks) == 1:
                    gx_full = gx
                else:
                    ops.copy_rows(mat(gx), mat(gx_full[:, r0:r1]))
                for i, (dW, db) in enumerate(zip(dWs, dbs)):
                    grads[off + 2 * i], grads[off + 2 * i + 1] = dW, db
                if dln is not None:
                    grads[off + 2 * n], grads[off + 2 * n + 1] = dln
            return gx_full

        g_rec_out = g_rec_out.contiguous()
        g_cat2 = back_blocks(ctx.aggr_blocks, saved_a, cat2, g_rec_out, 2 * d)
        # receiver grad: residual + cat2[:, :, :d] slice
        g_rec = _empty(B, n_rec, d, dev)
        ops.add_rows(mat(g_rec_out), mat(g_cat2, 0, d), mat(g_rec))
        # message grad: gather of the aggregate's grad (+ edge residual path)
        g_msg = _empty(B, M, d, dev)
        ops.gather_rows(mat(g_cat2, d, d), g.rec, mat(g_msg),
                        row_scale=g.inv_deg if ctx.mean else None)
        if ctx.update_edges and g_edge_out is not None:
            g_edge_out = g_edge_out.contiguous()
            ops.add_rows(mat(g_msg), mat(g_edge_out), mat(g_msg))
        g_cat = back_blocks(ctx.edge_blocks, saved_e, cat, g_msg, 3 * d)
        if ctx.update_edges and g_edge_out is not None:
            g_edge = _empty(B, M, d, dev)
            ops.add_rows(mat(g_cat, 0, d), mat(g_edge_out), mat(g_edge))
        else:
            g_edge = g_cat[:, :, :d]
        # the sender table spans g.n_send rows (max - min + 1 of the sender ids,
        # interaction_net.py:56); send_rep may have more rows, whose grad is zero
        if n_send > g.n_send:
            g_send = torch.zeros(B, n_send, d, dtype=torch.float32, device=dev)
        else:
            g_send = _empty(B, n_send, d, dev)
        ops.segment_sum(mat(g_cat, d, d), g.csc_colptr, g.csc_eid, mat(g_send[:, : g.n_send]))
        ops.segment_sum(mat(g_cat, 2 * d, d), g.csr_rowptr, g.csr_eid, mat(g_rec), accumulate=True)

        def fit(gr, shape):
            # inputs that came in with fewer batch items than B (2-D / broadcast)
            return gr if tuple(gr.shape) == tuple(shape) else gr.sum(0).reshape(shape)

        s_shape, r_shape, e_shape = ctx.in_shapes
        return (fit(g_send, s_shape), fit(g_rec, r_shape), fit(g_edge, e_shape), None, None, None,
                None, None) + tuple(grads)
