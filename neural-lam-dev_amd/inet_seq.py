"""ctypes mirror of the `nlam_inet_*` structs (include/nlam_hip.h): one host call per
InteractionNet forward / backward at hidden 64 (csrc/inet_host.cpp).  fused.py allocates the
buffers and keeps the autograd bookkeeping; the launch sequence itself runs in C++."""
import ctypes
import os

import torch

from ._lib import check, lib

_P, _I64, _I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int

# NLAM_INET_SEQ=0: issue the launches one ctypes call at a time from Python instead (same kernels,
# same results; what the per-kernel profiler of bench.py uses)
ENABLED = os.environ.get("NLAM_INET_SEQ", "1") != "0"


class Graph(ctypes.Structure):
    _fields_ = [("tiles", _P), ("ntiles", _I64), ("csr_rowptr", _P), ("csr_eid", _P),
                ("csr_send", _P), ("csr_rec", _P), ("inv_deg", _P), ("csc_colptr", _P),
                ("csc_eid", _P), ("n_send", _I64), ("n_rec", _I64), ("M", _I64),
                ("part_slot", _P), ("pcsc_colptr", _P), ("pcsc_rows", _P)]


class View(ctypes.Structure):
    _fields_ = [("ptr", _P), ("B", _I64), ("bstride", _I64), ("ld", _I64)]


class Weights(ctypes.Structure):
    _fields_ = [("W1", _P), ("ldW1", _I64), ("b1", _P), ("W2", _P), ("ldW2", _I64), ("b2", _P),
                ("gam", _P), ("bet", _P), ("V1", _P), ("ldV1", _I64), ("c1", _P), ("V2", _P),
                ("ldV2", _I64), ("c2", _P), ("gam2", _P), ("bet2", _P)]


class Args(ctypes.Structure):
    _fields_ = [("g", Graph), ("w", Weights), ("send", View), ("rec", View), ("edge", View),
                ("n_send_rows", _I64), ("B", _I64), ("d", _I), ("update_edges", _I), ("mean", _I),
                ("P", _P), ("Pr", _P), ("Pe", _P), ("agg", _P), ("e_out", _P), ("rec_out", _P),
                ("ps_given", _I), ("pr_given", _I)]


class Grads(ctypes.Structure):
    _fields_ = [("g_rec_out", _P), ("g_edge_out", _P), ("g_send", _P), ("g_rec", _P), ("g_edge", _P),
                ("dW1", _P), ("db1", _P), ("dW2", _P), ("db2", _P), ("dgam", _P), ("dbet", _P),
                ("dV1", _P), ("dc1", _P), ("dV2", _P), ("dc2", _P), ("dgam2", _P), ("dbet2", _P),
                ("g_send_add", _P)]


# the ctypes mirrors must have the layout the library was compiled with (a field added on one side
# only shifts every later field: the advisor's round-4 finding on nlam_inet_grads)
if ctypes.sizeof(Args) != int(lib.nlam_sizeof_inet_args()) or \
        ctypes.sizeof(Grads) != int(lib.nlam_sizeof_inet_grads()):
    raise ImportError(
        f"inet_seq: struct layout mismatch with libnlam_hip.so (Args {ctypes.sizeof(Args)} vs "
        f"{int(lib.nlam_sizeof_inet_args())}, Grads {ctypes.sizeof(Grads)} vs "
        f"{int(lib.nlam_sizeof_inet_grads())}): rebuild the library or update include/nlam_hip.h's mirror")


def sender_parts_on(t):
    """Per-tile sender partials instead of gh rows in the edge backward of the grid-side nets
    (nlam_edge_bwd_parts); NLAM_SENDER_PARTS=0: the gh rows and the per-edge sender lists."""
    return bool(getattr(t, "has_sender_parts", False)) and os.environ.get("NLAM_SENDER_PARTS", "1") != "0"


def graph_struct(t):
    """Graph struct of an EdgeTables (cached per device copy of its buffers)."""
    key = (t.csr_rowptr.data_ptr(), sender_parts_on(t))
    cached = getattr(t, "_inet_graph", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    parts = sender_parts_on(t)
    g = Graph(t.tiles.data_ptr(), t.ntiles, t.csr_rowptr.data_ptr(), t.csr_eid.data_ptr(),
              t.csr_send.data_ptr(), t.csr_rec.data_ptr(), t.inv_deg.data_ptr(),
              t.csc_colptr.data_ptr(), t.csc_eid.data_ptr(), t.n_send, t.n_rec, t.M,
              t.part_slot.data_ptr() if parts else None, t.pcsc_colptr.data_ptr() if parts else None,
              t.pcsc_rows.data_ptr() if parts else None)
    t._inet_graph = (key, g)
    return g


def _view(m):
    return View(m.ptr, m.B, m.bstride, m.ld)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def weights_struct(weights):
    W1, b1, W2, b2, gam, bet, V1, c1, V2, c2, gam2, bet2 = weights
    return Weights(W1.data_ptr(), W1.stride(0), b1.data_ptr(), W2.data_ptr(), W2.stride(0),
                   b2.data_ptr(), gam.data_ptr(), bet.data_ptr(), V1.data_ptr(), V1.stride(0),
                   c1.data_ptr(), V2.data_ptr(), V2.stride(0), c2.data_ptr(), gam2.data_ptr(),
                   bet2.data_ptr())


def make_args(tables, sm, rm, em, same, update_edges, mean, B, weights, bufs, ps_given=False,
              pr_given=False):
    """weights: (W1, b1, W2, b2, gam, bet, V1, c1, V2, c2, gam2, bet2); bufs: dict of the
    caller-allocated tensors P, Pr, Pe, agg, e_out, rec_out (None where unused; an empty dict
    gives the shape-only block nlam_inet_supported() is asked with)."""
    W1, W2, V1, V2 = weights[0], weights[2], weights[6], weights[8]
    if any(w.stride(-1) != 1 for w in (W1, W2, V1, V2)):
        return None
    w = weights_struct(weights)
    rec = View(None, 0, 0, 0) if same else _view(rm)
    return Args(graph_struct(tables), w, _view(sm), rec, _view(em), sm.rows, B, 64,
                int(update_edges), int(mean), _ptr(bufs.get("P")), _ptr(bufs.get("Pr")),
                _ptr(bufs.get("Pe")), _ptr(bufs.get("agg")), _ptr(bufs.get("e_out")),
                _ptr(bufs.get("rec_out")), int(bool(ps_given)), int(bool(pr_given)))


def supported(args):
    return args is not None and bool(lib.nlam_inet_supported(ctypes.byref(args)))


def forward(args, stream):
    check(lib.nlam_inet_fwd(ctypes.byref(args), stream), "nlam_inet_fwd")


def backward(args, grads, device, stream):
    n = int(lib.nlam_inet_bwd_workspace(ctypes.byref(args)))
    ws = torch.empty(n, dtype=torch.float32, device=device)
    check(lib.nlam_inet_bwd(ctypes.byref(args), ctypes.byref(grads), ws.data_ptr(), n, stream),
          "nlam_inet_bwd")
    return ws
