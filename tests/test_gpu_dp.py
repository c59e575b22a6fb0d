"""Data-parallel equivalence on the GPU: two ranks (two processes sharing the one
visible card, gloo transport for the rehearsal) each take half of a batch; after the
flat-buffer all-reduce their averaged gradient equals the single-process gradient of
the whole batch, and one fused AdamW step leaves both ranks with identical weights."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _get(q):
    """A rank's result; a rank that died reports its traceback instead of letting the test wait."""
    item = q.get(timeout=180)
    if isinstance(item[1], str) and item[1] == "error":
        raise RuntimeError(f"rank {item[0]} failed:\n{item[2]}")
    return item


def _collect(procs, q, n):
    """Results of all ranks; whatever happens, no rank is left alive on the card (a rank blocked
    in a collective after its peer died would disturb the timing-sensitive tests that follow)."""
    try:
        res = sorted([_get(q) for _ in range(n)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        return res
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=30)


def _guarded(name, rank, world, port, tmp, q, backend):
    try:
        globals()[name](rank, world, port, tmp, q, backend)
    except BaseException:   # noqa: BLE001 - reported to the parent, then re-raised
        import traceback

        q.put((rank, "error", traceback.format_exc()))
        raise


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_graph(tmp):
    """Written ONCE by the parent before the ranks start (two ranks writing the same files at the
    same time can read each other's half-written tensors)."""
    from neural_lam_amd import graphgen

    return graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(30, 28, 5000.0), None, False)


def _build(tmp):
    import numpy as np
    from neural_lam_amd import synthetic
    from neural_lam_amd.models import GraphLAM

    n = 30 * 28
    gen = torch.Generator().manual_seed(0)
    ds = synthetic.SyntheticDatastore(
        tmp, torch.randn(n, 1, generator=gen).numpy(), np.zeros(5), np.ones(5), np.zeros(5),
        np.ones(5), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
    torch.manual_seed(7)
    model = GraphLAM(synthetic.model_args(graph="g", hidden_dim=64, processor_layers=1),
                     config=None, datastore=ds)
    return model, n


def _worker(rank, world, port, tmp, q, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # gloo: both ranks share the one visible card; nccl (= RCCL): one card per rank
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from neural_lam_amd import parallel, synthetic

    model, n = _build(tmp)
    model = model.cuda()
    flat = parallel.FlatParams(model)
    red = parallel.GradAllReduce(flat)
    red.broadcast_params()
    opt = parallel.FlatAdamW(flat, lr=1e-2)
    full = synthetic.random_batch(4, 2, n, n_state=5, n_forcing_window=6, seed=3, device="cuda")
    mine = tuple(t[rank * 2 : rank * 2 + 2] if t is not None else None for t in full)
    flat.zero_grad()
    model.training_step(mine).backward()
    red.reduce()
    opt.step(grad_scale=1.0 / world)
    q.put((rank, (flat.grad / world).cpu().tolist(), flat.flat.cpu().tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_gradients_match_single_process(backend):
    """gloo: rehearsal on the one card of the test box; nccl: the RCCL path itself, on boxes
    with at least two cards (skipped otherwise)."""
    from neural_lam_amd import parallel, synthetic

    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank (it refuses two ranks on one device)")
    with tempfile.TemporaryDirectory() as tmp:
        _make_graph(tmp)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_guarded, args=("_worker", r, 2, port, tmp, q, backend)) for r in range(2)]
        for p in procs:
            p.start()
        res = _collect(procs, q, 2)
        model, n = _build(tmp)
    model = model.cuda()
    flat = parallel.FlatParams(model)
    full = synthetic.random_batch(4, 2, n, n_state=5, n_forcing_window=6, seed=3, device="cuda")
    flat.zero_grad()
    model.training_step(full).backward()
    flat.pack_grads()
    want = flat.grad.cpu()
    g0, g1 = torch.tensor(res[0][1]), torch.tensor(res[1][1])
    assert torch.equal(g0, g1)
    assert float((g0 - want).abs().max() / want.abs().max()) < 1e-4
    assert torch.equal(torch.tensor(res[0][2]), torch.tensor(res[1][2]))


def _overlap_worker(rank, world, port, tmp, q, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from neural_lam_amd import parallel, synthetic

    out = {}
    for overlap in (False, True):
        model, n = _build(tmp)
        model = model.cuda()
        flat = parallel.FlatParams(model)
        # small buckets: several collectives per step, issued from the backward hooks on the
        # side stream while autograd keeps launching kernels on the main stream
        red = parallel.GradAllReduce(flat, bucket_bytes=64 << 10, overlap=overlap)
        red.broadcast_params()
        opt = parallel.FlatAdamW(flat, lr=1e-2)
        grads = []
        for step in range(3):
            full = synthetic.random_batch(4, 2, n, n_state=5, n_forcing_window=6, seed=3 + step,
                                          device="cuda")
            mine = tuple(t[rank * 2 : rank * 2 + 2] if t is not None else None for t in full)
            flat.zero_grad()
            model.training_step(mine).backward()
            red.reduce()
            grads.append(flat.grad.clone())
            opt.step(grad_scale=1.0 / world)
        torch.cuda.synchronize()
        out[overlap] = (torch.stack(grads).cpu(), flat.flat.cpu().clone(), dict(red.stats),
                        len(red.ranges))
    same_g = torch.equal(out[False][0], out[True][0])
    same_w = torch.equal(out[False][1], out[True][1])
    q.put((rank, same_g, same_w, out[True][2], out[True][3], out[False][2]))
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_overlapped_buckets_on_device_tensors_bit_identical(backend):
    """GradAllReduce(overlap=True) with small buckets on DEVICE tensors: the hook-issued,
    side-stream collectives (event fork from the launch stream, pack on the side stream, join
    in reduce()) give bit-identical gradients and weights to the single trailing schedule over
    three optimiser steps, and the hooks did issue buckets during backward.  gloo on the one
    card of the test box; RCCL when the box has two cards."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank (it refuses two ranks on one device)")
    with tempfile.TemporaryDirectory() as tmp:
        _make_graph(tmp)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_guarded, args=("_overlap_worker", r, 2, port, tmp, q, backend))
                 for r in range(2)]
        for p in procs:
            p.start()
        res = _collect(procs, q, 2)
    for rank, same_g, same_w, stats, nbuckets, stats_off in res:
        assert same_g and same_w, f"rank {rank}: overlap changed the result"
        assert nbuckets >= 3
        assert stats["launched_in_backward"] > 0, stats
        assert stats["launched_in_backward"] + stats["launched_in_reduce"] == 3 * nbuckets
        assert stats_off["launched_in_backward"] == 0


def _one_rank_rccl_worker(rank, world, port, tmp, q, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    from neural_lam_amd import parallel, synthetic

    # the plain single-process step, in THIS process (same library state as the RCCL runs below:
    # the parent may have run tests that switch kernel families or arithmetic modes)
    model, n = _build(tmp)
    model = model.cuda()
    flat = parallel.FlatParams(model)
    opt = parallel.FlatAdamW(flat, lr=1e-2)
    full = synthetic.random_batch(4, 2, n, n_state=5, n_forcing_window=6, seed=3, device="cuda")
    for _ in range(2):
        flat.zero_grad()
        model.training_step(full).backward()
        flat.pack_grads()
        opt.step(grad_scale=1.0)
    torch.cuda.synchronize()
    out = {"plain": (flat.grad.cpu(), flat.flat.cpu(), {})}
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    for overlap in (False, True):
        model, n = _build(tmp)
        model = model.cuda()
        flat = parallel.FlatParams(model)
        # (small buckets: several collectives, several of them issued from backward hooks)
        red = parallel.GradAllReduce(flat, bucket_bytes=64 << 10, overlap=overlap,
                                     single_rank_collectives=True)
        assert red.active and len(red.ranges) > 2
        red.broadcast_params()
        opt = parallel.FlatAdamW(flat, lr=1e-2)
        full = synthetic.random_batch(4, 2, n, n_state=5, n_forcing_window=6, seed=3, device="cuda")
        for _ in range(2):
            flat.zero_grad()
            model.training_step(full).backward()
            red.reduce()
            opt.step(grad_scale=1.0)
        torch.cuda.synchronize()
        out[overlap] = (flat.grad.cpu(), flat.flat.cpu(), dict(red.stats))
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((0, {k: (v[0].tolist(), v[1].tolist(), v[2]) for k, v in out.items()}, None))
    dist.destroy_process_group()


def test_one_rank_rccl_collectives_leave_the_step_unchanged():
    """The RCCL ("nccl") backend itself, executed on the one card of the test box: a one-rank
    group created on the device, parameter broadcast, the flat gradient buckets all-reduced as ONE
    trailing pass and as hook-issued collectives on the side stream during backward, then MAX
    all-reduce + barrier (the bench's timing protocol).  A one-rank all-reduce returns its input,
    so gradients and weights after two optimiser steps must be BIT-identical to the plain
    single-process step's -- any mis-ordered stream wait or mis-sliced bucket view shows."""
    with tempfile.TemporaryDirectory() as tmp:
        _make_graph(tmp)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_guarded,
                             args=("_one_rank_rccl_worker", 0, 1, _free_port(), tmp, q, "nccl"))]
        procs[0].start()
        res = _collect(procs, q, 1)[0][1]
    want_g, want_w = torch.tensor(res["plain"][0]), torch.tensor(res["plain"][1])
    assert want_g.abs().max() > 0
    for overlap in (False, True):
        g, w, stats = res[overlap]
        assert torch.equal(torch.tensor(g), want_g), overlap
        assert torch.equal(torch.tensor(w), want_w), overlap
        if overlap:
            assert stats["launched_in_backward"] > 0, stats
        else:
            assert stats["launched_in_backward"] == 0 and stats["launched_in_reduce"] > 0, stats
