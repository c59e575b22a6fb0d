#!/usr/bin/env python3
"""Hot-path benchmark (contract: see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W

One "step" = one full GraphLAM training step on one batch of synthetic MEPS
data resident in HBM: unroll_prediction (ar_steps AR steps of encode /
process / decode) + loss + backward + gradient all-reduce (N > 1) + AdamW.
metric: mesh node-updates/s (fwd+bwd) = B_global * ar_steps * P * N_mesh / t_step
(SURVEY.md 8d).  For N > 1 the driver launches one rank per GPU through
torch.distributed.run; per-GPU batch is fixed (weak scaling).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # bf16 MFMA dense peak (same guide; 2:1-sparsity figures excluded)
MFMA_MODES = {0: "fp32", 1: "bf16x3", 2: "bf16"}
# matrix work actually issued per algorithmic fp32 flop, and the unit it is issued on
MFMA_TERMS = {"fp32": (1.0, MFMA_F32_PEAK_TFLOPS, "fp32 MFMA"),
              "bf16x3": (3.0, MFMA_BF16_PEAK_TFLOPS, "bf16 MFMA, 3 split terms per product"),
              "bf16": (1.0, MFMA_BF16_PEAK_TFLOPS, "bf16 MFMA")}


def roofline_entry(name, st, mfma_mode, traffic=None):
    """Roofline of one entry point from its HIP-event time and ALGORITHMIC work: the roof is
    the slower of the HBM time (bytes / 8 TB/s) and the matrix time of the arithmetic that
    actually ran (fp32: flops / 157.3 TF on the fp32 MFMA; bf16x3: 3 x flops / 2.5 PF on the
    bf16 MFMA; bf16: flops / 2.5 PF); frac = roof time / measured time."""
    sec = st["ms"] / 1e3 / max(1, st["calls"])          # per launch
    flops = st["flops"] / max(1, st["calls"])
    nbytes = st["bytes"] / max(1, st["calls"])
    terms, mpeak, unit_name = MFMA_TERMS[mfma_mode]
    t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
    t_mfma = terms * flops / (mpeak * 1e12)
    frac_hbm = t_hbm / sec if sec > 0 else 0.0
    frac_mfma = t_mfma / sec if sec > 0 else 0.0
    ent = {"kernel": name, "mfma_mode": mfma_mode, "avg_launch_us": sec * 1e6,
           "algorithmic_bytes": nbytes, "algorithmic_flops": flops,
           "executed_matrix_flops": terms * flops, "matrix_unit": unit_name,
           "frac_hbm": frac_hbm, "frac_mfma": frac_mfma, "traffic": traffic}
    if t_hbm >= t_mfma:
        ent.update(bound="hbm", achieved=nbytes / sec / 1e9 if sec > 0 else 0.0,
                   peak=HBM_PEAK_GBS, unit="GB/s", frac=frac_hbm)
    else:
        ent.update(bound="mfma", achieved=terms * flops / sec / 1e12 if sec > 0 else 0.0,
                   peak=mpeak, unit="TFLOP/s", frac=frac_mfma)
    return ent


def site_traffic(name, args):
    tpath = os.path.join(ROOT, "profiles", "traffic_sites.json")
    if not os.path.exists(tpath):
        return None
    try:
        table = json.load(open(tpath))
    except ValueError:
        return None
    ent = table.get(f"{args.model}-{args.hidden_dim}", {}).get(name)
    return ent.get("hbm_bytes_per_launch") if ent else None


def layer_roofline(stats, nprof, args, info, B, model):
    """a fused InteractionNet layer fwd + bwd moves 20 d (N + M) bytes (+ indices)
    per sample algorithmically; the m2m processor layers of GraphLAM are P such layers on the
    same graph.  Sum of every kernel launched at that call site / P layers vs that figure."""
    site = "m2m"
    ms = sum(v["ms"] for k, v in stats.items() if k.endswith("@" + site)) / nprof
    if ms <= 0 or args.model != "graph_lam":
        return None
    N, M, d, P = info["num_mesh"][0], sum(info["m2m_edges"]), args.hidden_dim, args.processor_layers
    nbytes = B * (20.0 * d * (N + M) + 2.0 * (4 * M + 4 * N + 4))
    us = ms * 1e3 / P
    return {"site": site, "layers": P, "us_per_layer_fwd_bwd": us, "algorithmic_bytes_per_layer": nbytes,
            "hbm_floor_us": nbytes / (HBM_PEAK_GBS * 1e9) * 1e6,
            "frac_hbm": nbytes / (HBM_PEAK_GBS * 1e9) / (us * 1e-6),
            "launches_per_layer": sum(v["calls"] for k, v in stats.items()
                                      if k.endswith("@" + site)) / nprof / P}


def sig(x, n=5):
    """Every float of the JSON line at n significant digits (the line must fit the driver's 9 kB
    stdout tail: tests/test_gpu_bench.py asserts < 8192 bytes)."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    if isinstance(x, dict):
        return {k: sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, n) for v in x]
    return x


TOP_KERNELS = 12   # entries of the per-kernel table kept in the line (all of them: the side file)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="samples per GPU (reference default 4)")
    ap.add_argument("--ar-steps", type=int, default=1)
    ap.add_argument("--hidden-dim", type=int, default=64)
    ap.add_argument("--processor-layers", type=int, default=4)
    ap.add_argument("--model", default="graph_lam", choices=["graph_lam", "hi_lam", "hi_lam_parallel"])
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps each; the median one is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-light", action="store_true",
                    help="CPU baseline on B = 1 only, 1 warm-up + 3 timed steps (the other-config "
                         "children: a stated baseline within the default run's time budget)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the per-kernel HIP-event pass (roofline = null)")
    ap.add_argument("--no-fp32-compare", action="store_true",
                    help="skip the extra run with exact fp32 MFMA (NLAM_MFMA=fp32) at N=1")
    ap.add_argument("--no-graph", action="store_true",
                    help="eager launches instead of replaying the captured HIP graph")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the child runs of the other BASELINE configs (N=1 default workload only)")
    return ap.parse_args()


def single_rank_rccl():
    """NLAM_BENCH_SINGLE_RANK_RCCL=1 (N = 1 only): take the multi-rank code path with a ONE-rank
    RCCL group -- process-group init on the device, parameter broadcast, hook-issued bucket
    all-reduces on the side stream, the eager-vs-graph probe, barriers and the MAX over ranks all
    execute on a one-GPU box.  The reported line is still an N = 1 measurement (n_gpus 1)."""
    return (os.environ.get("NLAM_BENCH_SINGLE_RANK_RCCL") == "1"
            and int(os.environ.get("WORLD_SIZE", "1")) == 1)


def setup_dist(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    dev_index = local % ndev   # one rank per GPU; rehearsals may stack ranks on one card
    if world > 1 or single_rank_rccl():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        # RCCL ("nccl") over xGMI is the product path; NLAM_BENCH_BACKEND=gloo only exists
        # to rehearse the multi-rank control flow on a 1-GPU box
        backend = os.environ.get("NLAM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    local = dev_index
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; run "
                         "`python bench.py --gpus N` (it starts the ranks itself) or launch "
                         "N ranks with torch.distributed.run")
    return rank, world, local


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N child ranks (one per GPU,
    RCCL rendezvous on 127.0.0.1), relay rank 0's JSON line, exit with the worst child
    code.  The parent never touches the GPU (no exec after GPU init, no GPU init at all)."""
    import socket
    import subprocess

    ndev = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
    if ndev < args.gpus and os.environ.get("NLAM_BENCH_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible; RCCL needs one "
                         "device per rank (NLAM_BENCH_BACKEND=gloo rehearses the control flow on one)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else None,
                                      text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst != 0:   # one line saying why the run has no JSON line (the ranks' own stderr is above)
        bad = ", ".join(f"rank {r}: exit {rc}" for r, rc in enumerate(rcs) if rc != 0)
        sys.stderr.write(f"bench.py --gpus {args.gpus}: FAILED ({bad}); visible GPUs: "
                         f"{torch.cuda.device_count()}\n")
    raise SystemExit(0 if worst == 0 else (worst if worst < 256 else 1))


def build(args, tmp):
    from neural_lam_amd import synthetic
    from neural_lam_amd.models import MODELS

    hier = args.model != "graph_lam"
    ds, graph_name, info = synthetic.meps_setup(tmp, hierarchical=hier, n_levels=3 if hier else None)
    margs = synthetic.model_args(graph=graph_name, hidden_dim=args.hidden_dim,
                                 processor_layers=args.processor_layers)
    torch.manual_seed(42)
    model = MODELS[args.model](margs, config=None, datastore=ds)
    return model, info


def rec_updates_per_layer(args, info):
    """Receiver-node updates per processor layer: N_mesh for
    GraphLAM; for Hi-LAM the down sweep (top same + down/same per lower level)
    plus the up sweep (bottom same + up/same per upper level) = 22,842 at L=3."""
    n = info["num_mesh"]
    if args.model == "graph_lam":
        return n[0]
    if args.model == "hi_lam_parallel":   # one InteractionNet over all levels per layer
        return sum(n)
    return (n[-1] + 2 * sum(n[:-1])) + (n[0] + 2 * sum(n[1:]))


def all_receiver_updates(args, info):
    """Receiver-node updates of one sample-step over every InteractionNet."""
    mesh = info["num_mesh"]
    total = args.processor_layers * rec_updates_per_layer(args, info)
    total += mesh[0] + info["num_grid"]                      # g2m + m2g
    if args.model != "graph_lam":
        total += sum(mesh[1:]) + sum(mesh[:-1])              # init up-sweep + read-out
    return total


def host_cores():
    """Threads for the CPU baseline: the cores this process may actually use
    (affinity mask and cgroup CPU quota), capped at 32 -- beyond that the
    reference's torch CPU ops (index_select / scatter_add_ / small GEMMs) stop
    scaling and oversubscribed boxes thrash."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_baseline(args, tmp, info, model):
    """Oracle (CPU restatement of the reference's PyG path) fwd+loss+bwd on the host
    cores, bounded sample: ar_steps=1, B=1 and B=4, 2 warm-ups + median of 5
    each (about 30-40 s of CPU work for GraphLAM-64).  `value` is the better of the two."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import statistics

    import nlam_oracle as orc
    from neural_lam_amd import synthetic

    cores = host_cores()
    old = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        gname = "hierarchical" if args.model != "graph_lam" else "multiscale"
        _, graph = orc.load_graph(os.path.join(tmp, "graph", gname))
        sd = {k: v.detach().cpu().clone().requires_grad_(True)
              for k, v in model.state_dict().items()}
        data = {k: getattr(model, k).detach().cpu() for k in
                ("grid_static_features", "diff_mean", "diff_std", "boundary_mask", "per_var_std")}
        cfg = {"model": args.model, "hidden_layers": 1, "processor_layers": args.processor_layers,
               "mesh_aggr": "sum", "loss": "wmse"}
        T = 1
        per_b = {}
        budget_t0 = time.perf_counter()
        light = bool(getattr(args, "cpu_baseline_light", False))
        nwarm = 1 if light else 2
        for B in ((1,) if light else (1, 4)):
            init, target, forcing, _ = synthetic.random_batch(B, T, info["num_grid"])
            times = []
            for it in range(4 if light else 7):
                t0 = time.perf_counter()
                loss, _ = orc.training_loss(sd, graph, cfg, data, init, target, forcing)
                torch.autograd.grad(loss, list(sd.values()))
                times.append(time.perf_counter() - t0)
                # bounded: a slow host (or a wide model) stops after 3 timed steps / 60 s
                # (light: after one timed step once 30 s are spent)
                if light and it >= 1 and time.perf_counter() - budget_t0 > 30.0:
                    break
                if it >= 4 and time.perf_counter() - budget_t0 > 60.0:
                    break
            timed = times[nwarm:] if len(times) > nwarm else times[-1:]
            t = statistics.median(timed)
            upd = B * T * args.processor_layers * rec_updates_per_layer(args, info)
            per_b[B] = {"s_per_step": t, "value": upd / t, "timed_steps": len(timed)}
        bestB = max(per_b, key=lambda b: per_b[b]["value"])
        per = ", ".join(f"B={b} {v['s_per_step']:.2f} s/step" for b, v in per_b.items())
        return {"value": per_b[bestB]["value"], "unit": "mesh node-updates/s", "cores": cores,
                "kind": "port",
                "per_batch": {str(b): v for b, v in per_b.items()},
                "sample": f"oracle (pure-torch CPU restatement of the PyG path) full train step "
                          f"fwd+loss+bwd, ar_steps={T}, median of "
                          f"{per_b[bestB]['timed_steps']} after {nwarm} warm-up(s) "
                          f"({per}; value = B={bestB}), "
                          f"torch {torch.__version__}, {cores} threads"}
    finally:
        torch.set_num_threads(old)


# The other BASELINE.json configs, run as fresh child processes of the default N = 1 run so that
# the driver's one bench line witnesses them too (children, never a re-exec of this
# GPU-initialised process).  Bounded in total; a failure is reported in the line, never fatal.
OTHER_CONFIGS = (
    ("hi_lam-128 (configs[2])", ["--model", "hi_lam", "--hidden-dim", "128", "--cpu-baseline-light"], {}),
    ("hi_lam-256 bf16 (configs[4], per GPU)", ["--model", "hi_lam", "--hidden-dim", "256"],
     {"NLAM_MFMA": "bf16"}),
    ("graph_lam-64 ar_steps=4 (configs[3], per GPU)", ["--ar-steps", "4"], {}),
    ("hi_lam-64", ["--model", "hi_lam", "--hidden-dim", "64"], {}),
)
OTHER_CONFIGS_BUDGET_S = 190.0   # (incl. ~40 s of CPU baseline for configs[2])


def run_other_configs(args):
    import subprocess

    out = {}
    t_start = time.perf_counter()
    for name, extra, env in OTHER_CONFIGS:
        left = OTHER_CONFIGS_BUDGET_S - (time.perf_counter() - t_start)
        if left < 20.0:
            out[name] = {"error": "skipped: the time budget of the other configs is spent"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3",
               "--windows", "3", "--batch", str(args.batch), "--processor-layers",
               str(args.processor_layers), "--no-fp32-compare", "--no-other-configs"] + extra
        if "--cpu-baseline-light" not in extra:
            cmd.append("--no-cpu-baseline")
        try:
            r = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True,
                               timeout=left)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            roof = j.get("roofline") or {}
            out[name] = {"ms_per_step": j["ms_per_step"], "value": j["value"], "dtype": j["dtype"],
                         "roofline": {"kernel": roof.get("kernel"), "frac": roof.get("frac"),
                                      "bound": roof.get("bound"),
                                      "avg_launch_us": roof.get("avg_launch_us")},
                         "launches_per_step": j.get("launches_per_step"),
                         "workload": j["config"]["workload"], "ar_steps": j["config"]["ar_steps"],
                         "batch_per_gpu": j["config"]["batch_per_gpu"],
                         "wall_s": time.perf_counter() - t_start}
            cb = j.get("cpu_baseline")
            if cb:
                out[name]["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"],
                                             "cores": cb["cores"], "kind": cb["kind"],
                                             "sample": cb["sample"]}
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the bench line
            out[name] = {"error": repr(e)[:300]}
    return out


def main():
    args = parse()
    if args.hidden_dim == 256:
        # BASELINE configs[4] is "hidden_dim=256, bf16": the bench line of that width is quoted in
        # bf16-mixed arithmetic (the feature-split kernels also run in the default split-bf16 mode,
        # fp32-grade and ~1.8x slower; NLAM_MFMA is read once, when the library loads; an explicit
        # setting -- e.g. bf16x3 or fp32 -- is respected)
        os.environ.setdefault("NLAM_MFMA", "bf16")
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)   # never returns
    rank, world, local = setup_dist(args)
    dev = torch.device("cuda", local)
    from neural_lam_amd import ops, parallel, synthetic

    tmpdir = tempfile.TemporaryDirectory(prefix=f"nlam_bench_r{rank}_")
    tmp = tmpdir.name
    model, info = build(args, tmp)
    model = model.to(dev)
    flat = parallel.FlatParams(model)
    multi = world > 1 or single_rank_rccl()   # (a process group exists, collectives are issued)
    reducer = parallel.GradAllReduce(flat, single_rank_collectives=single_rank_rccl())
    reducer.broadcast_params()
    opt = parallel.FlatAdamW(flat, lr=1e-3)
    B, T = args.batch, args.ar_steps
    batch = synthetic.random_batch(B, T, info["num_grid"], seed=100 + rank, device=dev)
    gscale = 1.0 / world

    def step():
        flat.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        reducer.reduce()
        opt.step(grad_scale=gscale)
        return loss

    def barrier():
        if multi:
            dist.barrier()

    graphed = None
    launch = {"mode": "eager"}
    # Multi-rank runs time BOTH schedules for a few steps and keep the faster one (max over ranks):
    # eager launches with gradient buckets all-reduced from backward hooks (overlap, but ~40 us of
    # host time per launch: host-bound once a step is ~100 short kernels), or HIP-graph replay of
    # forward + backward + packing with the buckets all-reduced after it.
    # NLAM_BENCH_MULTIRANK_GRAPH=off skips the capture and stays eager; =probe forces the probe.
    # Default: probe on RCCL (one process per GPU); off for the gloo rehearsal, where the ranks
    # are stacked on ONE card: replaying a HIP graph there adds hardware queues per process, the
    # card's queues become oversubscribed and every later step of both ranks (eager ones too) is
    # time-sliced at ~100 ms granularity (measured: GraphLAM-64 14 ms eager -> 600 ms; r03 notes).
    backend_name = os.environ.get("NLAM_BENCH_BACKEND", "nccl")
    multirank_graph = os.environ.get(
        "NLAM_BENCH_MULTIRANK_GRAPH", "probe" if backend_name == "nccl" else "off") == "probe"
    if not args.no_graph and (not multi or multirank_graph):
        reducer.hooks_enabled = False      # no collectives inside the capture
        graphed = parallel.GraphedTrainStep(model, flat, batch)
        reducer.hooks_enabled = True

        def gstep():
            loss = graphed()
            reducer.reduce(packed=True)
            opt.step(grad_scale=gscale)
            return loss

    use_graph = graphed is not None and graphed.graph is not None
    if use_graph and multi:
        # Two schedules for a multi-rank step: (a) eager launches, gradient buckets all-reduced
        # from backward hooks (overlap, but ~40 us of host time per launch: Hi-LAM at hidden 64
        # is then host-bound), (b) HIP-graph replay of forward + backward + packing, buckets
        # all-reduced after it.  Probe both, take the slower rank's view, use the faster one.
        def probe(fn, hooks, n=3):
            reducer.hooks_enabled = hooks
            fn()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        t_pair = torch.tensor([probe(step, True), probe(gstep, False)], device=dev,
                              dtype=torch.float64)
        dist.all_reduce(t_pair, op=dist.ReduceOp.MAX)
        use_graph = bool(t_pair[1] < t_pair[0])
        launch["probe_ms"] = {"eager_overlap": float(t_pair[0]) * 1e3,
                              "hip_graph_trailing_allreduce": float(t_pair[1]) * 1e3}
        if not use_graph:   # do not keep a captured graph (and its memory pool) around unused
            graphed = None
            torch.cuda.empty_cache()
    reducer.hooks_enabled = not use_graph
    if use_graph:
        launch["mode"] = "hip_graph" if not multi else "hip_graph+trailing_allreduce"
    elif multi and reducer.overlap:
        launch["mode"] = "eager+overlapped_allreduce"
    timed_step = gstep if use_graph else step

    for _ in range(args.warmup):
        timed_step()
    # `--windows` timed windows of EXACTLY `--steps` steps each, every one bracketed by barrier +
    # synchronize on both sides and reduced with MAX over ranks; the MEDIAN window is the
    # reported one (ms_per_step, value), the others give the spread (a 20-step window of a 3 ms
    # step is 60 ms: one window says nothing about run-to-run variation)
    window_s = []
    for _ in range(max(1, args.windows)):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = timed_step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        w = time.perf_counter() - t0
        if multi:
            t = torch.tensor([w], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = float(t.item())
        window_s.append(w)
    elapsed = sorted(window_s)[len(window_s) // 2]
    loss_val = float(loss.detach())

    # per-kernel HIP-event pass (same steps, events bracket every C-ABI launch on the
    # launch stream); kept out of the timed region above so `value` is unperturbed
    roofline = None
    kernels = None
    scatter = None
    layer = None
    from neural_lam_amd._lib import lib as _nlam_lib
    mfma_mode = MFMA_MODES[int(_nlam_lib.nlam_mfma_mode())]
    nprof = max(1, min(3, args.steps))
    reducer.hooks_enabled = True   # the per-kernel pass below launches eagerly
    if not args.no_kernel_timing and rank != 0:
        for _ in range(nprof):   # every rank takes part in the steps' collective
            step()
        torch.cuda.synchronize()
    if not args.no_kernel_timing and rank == 0:
        ops.PROFILER = ops.KernelProfiler()
        for _ in range(nprof):
            # the host must run AHEAD of the device for an event pair to bracket only its kernel:
            # with an empty queue the start event is reached before the launch packet arrives and
            # the interval includes the host's gap (round 4: the first launch of a step, the static
            # embedders' forward, read 87 us here against 46 us under rocprofv3).  ~1 ms of device
            # sleep in front of each profiled step keeps the queue fed.
            torch.cuda._sleep(2_000_000)
            step()
        dump = os.environ.get("NLAM_BENCH_DUMP_ORDER")
        if dump:   # launch order of ONE step, for tools/site_stats.py (rocprof -> call sites)
            seq = [n for (n, _s, _e, _f, _b) in ops.PROFILER.pending]
            json.dump(seq[: len(seq) // nprof], open(dump, "w"))
        stats = ops.PROFILER.collect()
        ops.PROFILER = None
        kernels = {
            k: {"calls_per_step": v["calls"] / nprof, "ms_per_step": v["ms"] / nprof,
                "gflop_per_step": v["flops"] / nprof / 1e9, "mb_per_step": v["bytes"] / nprof / 1e6}
            for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])
        }
        name, st = max(stats.items(), key=lambda kv: kv[1]["ms"])
        # PMC bytes per launch of exactly this entry point at exactly this call site
        # (profiles/traffic_sites.json, written by tools/site_stats.py from rocprofv3 --pmc passes
        # of this script); None when that site was not profiled -- never another site's number
        traffic = site_traffic(name, args)
        roofline = roofline_entry(name, st, mfma_mode, traffic)
        roofline["note"] = ("per-kernel times come from an EAGER pass with HIP events around every "
                            "launch; `value` / ms_per_step from the timed windows (HIP-graph replay "
                            "at N = 1), so the kernel times sum to slightly more than a step")
        layer = layer_roofline(stats, nprof, args, info, B, model)
        # the scatter-add the north star names: the m2m aggregate.  In the step it has no launch of
        # its own any more (forward: receiver sums inside nlam_edge_fwd@m2m; backward: the
        # sender-side sums are a CSC gather inside nlam_node_bwd / nlam_lin_bwd_multi), so the bare
        # kernel is timed here on the same m2m tables: 50 launches between ONE event pair (a
        # 12-16 us kernel carries a few us of per-launch event overhead otherwise).  Algorithmic
        # bytes: SURVEY.md 8(d) "aggregate only" = 4 d (M + N_r) + indices per sample.
        net0 = getattr(model, "processor", None)
        net0 = getattr(net0, "module_0", None) if net0 is not None else None
        if net0 is not None and getattr(net0, "tables", None) is not None:
            tb = net0.tables
            Mm, Nr, dd = int(tb.M), int(tb.n_rec), args.hidden_dim
            msg = torch.randn(B, Mm, dd, device=dev)
            out_ = torch.empty(B, Nr, dd, device=dev)
            for _ in range(3):
                ops.segment_sum(ops.mat(msg), tb.csr_rowptr, tb.csr_eid, ops.mat(out_))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.segment_sum(ops.mat(msg), tb.csr_rowptr, tb.csr_eid, ops.mat(out_))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            nbytes = B * (4.0 * dd * (Mm + Nr) + 4.0 * Mm + 4.0 * (Nr + 1))
            g = nbytes / us / 1e3
            scatter = {"kernel": "nlam_segment_sum on the m2m tables (stand-alone aggregate)",
                       "bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": g / HBM_PEAK_GBS, "avg_launch_us": us, "launches": 50,
                       "algorithmic_bytes": nbytes,
                       "note": "bare kernel, back to back; inside the step the forward scatter-add "
                               "is fused into nlam_edge_fwd@m2m (fused_forward) and the backward's "
                               "sender-side sums into nlam_node_bwd@m2m"}
            fw = stats.get("nlam_edge_fwd@m2m")
            if fw and fw["ms"] > 0:
                scatter["fused_forward"] = roofline_entry("nlam_edge_fwd@m2m", fw, mfma_mode)
            del msg, out_

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, tmp, info, model)

    fp32_cmp = None
    if rank == 0:
        if (world == 1 and mfma_mode != "fp32" and not args.no_fp32_compare
                and not args.no_cpu_baseline):
            # the same step with exact fp32 MFMA (the mode is fixed per process): reported
            # next to the default so that the split-bf16 gain is visible in the bench line
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--batch", str(args.batch),
                   "--ar-steps", str(args.ar_steps), "--hidden-dim", str(args.hidden_dim),
                   "--processor-layers", str(args.processor_layers), "--model", args.model,
                   "--windows", "3", "--no-cpu-baseline", "--no-kernel-timing", "--no-fp32-compare",
                   "--no-other-configs"]
            if args.no_graph:
                cmd.append("--no-graph")
            try:
                r = subprocess.run(cmd, env=dict(os.environ, NLAM_MFMA="fp32"),
                                   capture_output=True, text=True, timeout=600)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                j = json.loads(line)
                fp32_cmp = {"mfma_mode": j.get("mfma_mode"), "ms_per_step": j["ms_per_step"],
                            "value": j["value"]}
            except Exception as e:  # reported, never fatal for the bench line
                fp32_cmp = {"error": repr(e)[:200]}
        other = None
        default_workload = (args.model == "graph_lam" and args.hidden_dim == 64 and T == 1)
        if world == 1 and default_workload and not args.no_other_configs:
            other = run_other_configs(args)
        upd_per_layer = rec_updates_per_layer(args, info)
        ms = elapsed / args.steps * 1e3
        value = world * B * T * args.processor_layers * upd_per_layer / (elapsed / args.steps)
        out = {
            "metric": "mesh node-updates/sec (fwd+bwd)", "value": value,
            "unit": "mesh node-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "f32(bf16x3)", "bf16": "bf16"}[mfma_mode],
            "data": "synthetic",
            # fp32 storage / accumulation / elementwise; GEMM products on the matrix cores as
            # exact fp32 MFMA ("fp32") or as 3 bf16 MFMA terms of hi/lo-split operands
            # ("bf16x3", ~2^-16 relative per product; parity tests hold 1e-4 / 2e-3 either way)
            "mfma_mode": mfma_mode, "exact_fp32_mfma": fp32_cmp,
            "config": {
                "workload": f"{args.model} on synthetic MEPS 238x268 grid "
                            f"({info['num_grid']} grid nodes, mesh {info['num_mesh']}, "
                            f"m2m {sum(info['m2m_edges'])} / g2m {info['g2m_edges']} / "
                            f"m2g {info['m2g_edges']} edges), hidden_dim {args.hidden_dim}, "
                            f"{args.processor_layers} processor layers, full train step "
                            "(fwd+loss+bwd+allreduce+AdamW)",
                "batch_per_gpu": B, "global_batch": world * B, "ar_steps": T,
                "parallelism": f"dp{world}",
            },
            "steps_per_s": 1e3 / ms, "loss": loss_val,
            "windows": {"n": len(window_s), "steps_each": args.steps,
                        "ms_per_step": [w / args.steps * 1e3 for w in window_s],
                        "median_ms_per_step": ms,
                        "min_ms_per_step": min(window_s) / args.steps * 1e3},
            # processor updates + g2m (mesh receivers) + m2g (grid receivers)
            # [+ Hi-LAM init/read-out sweeps] per AR step
            "all_receiver_updates_per_s": world * B * T * all_receiver_updates(args, info)
            / (elapsed / args.steps),
            "hip_graph": bool(use_graph), "launch": launch,
            "single_rank_rccl": single_rank_rccl(),
            "rccl_ranks": dist.get_world_size() if multi else 1,
            "backend": (dist.get_backend() if multi else None),
            "grad_allreduce": reducer.describe(),
            "roofline": roofline, "layer_roofline": layer, "scatter_add_roofline": scatter,
            "cpu_baseline": cpu,
            "launches_per_step": (sum(v["calls_per_step"] for v in kernels.values())
                                  if kernels else None),
            "other_configs": other,
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        # the per-kernel table goes LAST and short (top entries by time, 4 significant digits:
        # [calls, ms, GFLOP, MB] per step); the whole table is written to a side file that gpurun
        # pulls back (gpurun_out/) -- the line itself stays under 8 kB so that the driver's stdout
        # tail holds all of it
        if kernels:
            side = os.environ.get("NLAM_BENCH_KERNELS_FILE") or os.path.join(
                ROOT, "gpurun_out", f"bench_kernels_{args.model}-{args.hidden_dim}"
                                    f"{'' if T == 1 else f'-ar{T}'}.json")
            try:
                os.makedirs(os.path.dirname(side), exist_ok=True)
                json.dump({"ms_per_step": ms, "kernels": kernels}, open(side, "w"), indent=0)
                out["kernels_file"] = os.path.relpath(side, ROOT)
            except OSError as e:
                out["kernels_file"] = f"not written: {e!r}"[:120]
            top = list(kernels.items())[:TOP_KERNELS]
            out["kernels_top"] = {
                "columns": ["calls", "ms", "gflop", "mb"], "per": "step",
                "sum_ms_all": sum(v["ms_per_step"] for v in kernels.values()),
                "rows": {k: sig([v["calls_per_step"], v["ms_per_step"], v["gflop_per_step"],
                                 v["mb_per_step"]], 4) for k, v in top}}
        print(json.dumps(sig(out), separators=(",", ":")), flush=True)
    if multi:
        dist.destroy_process_group()
    tmpdir.cleanup()


if __name__ == "__main__":
    main()
