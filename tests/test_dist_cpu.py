"""world_size-2 gloo tests (CPU) of the data-parallel exchange: bucketed flat
gradient all-reduce and parameter broadcast."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_lam_amd import parallel

    torch.manual_seed(rank)  # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(5, 9), torch.nn.SiLU(), torch.nn.Linear(9, 4),
                              torch.nn.LayerNorm(4))
    flat = parallel.FlatParams(net)
    red = parallel.GradAllReduce(flat, bucket_bytes=bucket_bytes)
    red.broadcast_params()
    p0 = flat.gather(flat.flat).clone()
    gen = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(6, 5, generator=gen)
    net(x).pow(2).sum().backward()
    local = torch.cat([p.grad.reshape(-1) for p in flat.params])
    red.reduce()
    # plain lists: tensors in an mp.Queue travel as fds that die with the sender
    q.put((rank, p0.tolist(), local.tolist(), flat.gather(flat.grad).tolist(), len(red.ranges)))
    dist.destroy_process_group()


def _run(bucket_bytes):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0a, la, ga, nb), (_, p0b, lb, gb, _) = [
        (r, torch.tensor(a), torch.tensor(b), torch.tensor(c), n) for r, a, b, c, n in res
    ]
    assert torch.equal(p0a, p0b), "parameters differ after broadcast"
    assert torch.allclose(ga, la + lb, atol=1e-6) and torch.equal(ga, gb)
    return nb


def test_single_bucket_allreduce():
    assert _run(bucket_bytes=1 << 20) == 1


def test_multi_bucket_allreduce():
    assert _run(bucket_bytes=64) > 2


def test_bucket_ranges_cover_all_params_in_reverse():
    from neural_lam_amd.parallel import bucket_ranges

    sizes = [10, 3, 50, 7, 7, 100, 1]
    r = bucket_ranges(sizes, 40)
    flat = sorted(i for lo, hi in r for i in range(lo, hi))
    assert flat == list(range(len(sizes)))
    assert r[0][1] == len(sizes) and r[-1][0] == 0


def _overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_lam_amd import parallel

    def make():
        torch.manual_seed(7)
        return torch.nn.Sequential(
            torch.nn.Linear(5, 16), torch.nn.SiLU(), torch.nn.Linear(16, 16), torch.nn.SiLU(),
            torch.nn.Linear(16, 4), torch.nn.LayerNorm(4))

    gen = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(6, 5, generator=gen)
    out = {}
    for name, overlap in (("plain", False), ("overlap", True)):
        net = make()
        flat = parallel.FlatParams(net)
        red = parallel.GradAllReduce(flat, bucket_bytes=256, overlap=overlap)
        red.broadcast_params()
        for step in range(2):          # second step: hook state must have been reset
            flat.zero_grad()
            net(x).pow(2).sum().backward()
            launched_before_reduce = red.stats["launched_in_backward"]
            red.reduce()
        out[name] = (flat.gather(flat.grad).tolist(), launched_before_reduce,
                     red.stats["launched_in_reduce"], len(red.ranges))
        # two backward passes accumulate into p.grad before ONE reduce (what a re-entrant
        # checkpointed rollout does): every hook fires twice; a parameter counts once per
        # step towards its bucket and late contributions are re-reduced
        flat.zero_grad()
        net(x).pow(2).sum().backward()
        net(0.5 * x).sum().backward()
        red.reduce()
        out[name + "_accum"] = flat.gather(flat.grad).tolist()
    q.put((rank, out))
    dist.destroy_process_group()


def test_overlap_hooks_fire_during_backward_and_match_plain_path():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        g_plain, lb_plain, lr_plain, nb = res[rank]["plain"]
        g_ovl, lb_ovl, lr_ovl, nb2 = res[rank]["overlap"]
        assert nb == nb2 and nb > 2
        assert lb_plain == 0 and lr_plain == 2 * nb          # everything issued by reduce()
        assert lb_ovl == 2 * nb and lr_ovl == 0              # every bucket issued by a hook
        assert g_plain == g_ovl                               # bit for bit
        assert res[rank]["plain_accum"] == res[rank]["overlap_accum"]
    assert res[0]["overlap"][0] == res[1]["overlap"][0]


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_lam_amd.models.ar_model import ARModel

    class Shell:   # the two methods need nothing of the model but the process group
        pass

    sh = Shell()
    sh._is_rank_zero = lambda: ARModel._is_rank_zero(sh)
    sh.all_gather_cat = lambda t: ARModel.all_gather_cat(sh, t)
    sh.state_std = torch.tensor([2.0, 0.5, 1.0])
    sh.eval_results = {}
    local = torch.full((2, 4, 3), float(rank + 1))          # (N_eval per rank, steps, d_f)
    out = ARModel.aggregate_metrics(sh, {"mse": [local], "mae": [local]}, prefix="val")
    gathered = ARModel.all_gather_cat(sh, local)
    q.put((rank, gathered.tolist(), {k: v.tolist() for k, v in out.items()}))
    dist.destroy_process_group()


def test_all_gather_cat_and_metric_aggregation_two_ranks():
    """Reference ar_model.py:311-320, 610-644: metrics are gathered over the ranks, averaged
    over all evaluated samples, *mse -> *rmse, rescaled by state_std, on rank 0 only."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (g, o) for r, g, o in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0 = torch.tensor(res[0][0])
    assert g0.shape == (4, 4, 3) and torch.equal(g0[:2], torch.ones(2, 4, 3)) and \
        torch.equal(g0[2:], torch.full((2, 4, 3), 2.0))
    assert res[1][1] == {}                                   # only rank 0 aggregates
    rmse = torch.tensor(res[0][1]["val_rmse"])
    mae = torch.tensor(res[0][1]["val_mae"])
    std = torch.tensor([2.0, 0.5, 1.0])
    assert torch.allclose(rmse, torch.sqrt(torch.tensor(1.5)) * std.expand(4, 3))
    assert torch.allclose(mae, 1.5 * std.expand(4, 3))
