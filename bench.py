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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the per-kernel HIP-event pass (roofline = null)")
    ap.add_argument("--no-fp32-compare", action="store_true",
                    help="skip the extra run with exact fp32 MFMA (NLAM_MFMA=fp32) at N=1")
    ap.add_argument("--no-graph", action="store_true",
                    help="eager launches instead of replaying the captured HIP graph")
    return ap.parse_args()


def setup_dist(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    dev_index = local % ndev   # one rank per GPU; rehearsals may stack ranks on one card
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    assert world == args.gpus or world == 1, (
        f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    )
    return rank, world, local


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
    cores, bounded sample: B=1, ar_steps=1, 1 warm-up + 2 timed steps."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
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
        B, T = 1, 1
        init, target, forcing, _ = synthetic.random_batch(B, T, info["num_grid"])
        times = []
        for it in range(3):
            t0 = time.perf_counter()
            loss, _ = orc.training_loss(sd, graph, cfg, data, init, target, forcing)
            torch.autograd.grad(loss, list(sd.values()))
            times.append(time.perf_counter() - t0)
        t = sum(times[1:]) / len(times[1:])
        upd = B * T * args.processor_layers * rec_updates_per_layer(args, info)
        return {"value": upd / t, "unit": "mesh node-updates/s", "cores": cores, "kind": "port",
                "sample": f"oracle (pure-torch CPU restatement of the PyG path) full train step "
                          f"fwd+loss+bwd, B={B}, ar_steps={T}, mean of 2 after 1 warm-up, "
                          f"{t:.2f} s/step, torch {torch.__version__}, {cores} threads"}
    finally:
        torch.set_num_threads(old)


def main():
    args = parse()
    rank, world, local = setup_dist(args)
    dev = torch.device("cuda", local)
    from neural_lam_amd import ops, parallel, synthetic

    tmpdir = tempfile.TemporaryDirectory(prefix=f"nlam_bench_r{rank}_")
    tmp = tmpdir.name
    model, info = build(args, tmp)
    model = model.to(dev)
    flat = parallel.FlatParams(model)
    reducer = parallel.GradAllReduce(flat)
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

    graphed = None
    if not args.no_graph and world == 1:   # multi-rank runs launch eagerly (the gain is <1 %)
        graphed = parallel.GraphedTrainStep(model, flat, batch)

        def gstep():
            loss = graphed()
            reducer.reduce(packed=True)
            opt.step(grad_scale=gscale)
            return loss

    timed_step = gstep if (graphed is not None and graphed.graph is not None) else step

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        timed_step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = timed_step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss)

    # per-kernel HIP-event pass (same steps, events bracket every C-ABI launch on the
    # launch stream); kept out of the timed region above so `value` is unperturbed
    roofline = None
    kernels = None
    scatter = None
    nprof = max(1, min(3, args.steps))
    if not args.no_kernel_timing and rank != 0:
        for _ in range(nprof):   # every rank takes part in the steps' collective
            step()
        torch.cuda.synchronize()
    if not args.no_kernel_timing and rank == 0:
        ops.PROFILER = ops.KernelProfiler()
        for _ in range(nprof):
            step()
        stats = ops.PROFILER.collect()
        ops.PROFILER = None
        kernels = {
            k: {"calls_per_step": v["calls"] / nprof, "ms_per_step": v["ms"] / nprof,
                "gflop_per_step": v["flops"] / nprof / 1e9, "mb_per_step": v["bytes"] / nprof / 1e6}
            for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])
        }
        name, st = max(stats.items(), key=lambda kv: kv[1]["ms"])
        sec = st["ms"] / 1e3
        tf = st["flops"] / sec / 1e12 if sec > 0 else 0.0
        gbs = st["bytes"] / sec / 1e9 if sec > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            # PMC bytes per launch of the HIP kernel behind this entry point / call site
            base, _, site = name.partition("@")
            flag = "true" if site == "m2m" else "false"
            prefix = {"nlam_edge_bwd": "edge_bwd_kernel<64, %s" % flag,
                      "nlam_edge_fwd": "edge_fwd_kernel<64, %s" % flag,
                      "nlam_segment_sum": "segment_sum_"}.get(base)
            table = json.load(open(tpath))
            ent = next((v for k, v in sorted(table.items()) if prefix and k.startswith(prefix)),
                       None)
            traffic = ent["hbm_bytes_per_launch"] if ent else None
        if tf / MFMA_F32_PEAK_TFLOPS >= gbs / HBM_PEAK_GBS:
            roofline = {"kernel": name, "bound": "mfma", "achieved": tf,
                        "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                        "avg_launch_us": st["ms"] * 1e3 / st["calls"]}
        else:
            roofline = {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                        "avg_launch_us": st["ms"] * 1e3 / st["calls"]}
        # the scatter-add the north star names: m2m aggregate (segment-sum) launches
        agg = stats.get("nlam_segment_sum@m2m")
        if agg and agg["ms"] > 0:
            g = agg["bytes"] / (agg["ms"] / 1e3) / 1e9
            scatter = {"kernel": "nlam_segment_sum@m2m", "bound": "hbm", "achieved": g,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                       "avg_launch_us": agg["ms"] * 1e3 / agg["calls"]}
            # the same kernel on the same m2m tables, 50 launches between ONE event pair:
            # a 12-16 us kernel carries a few us of per-launch event overhead above
            net0 = getattr(model, "processor", None)
            net0 = getattr(net0, "module_0", None) if net0 is not None else None
            if net0 is not None:
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
                scatter["back_to_back"] = {"launches": 50, "avg_launch_us": us,
                                           "achieved": nbytes / us / 1e3,
                                           "frac": nbytes / us / 1e3 / HBM_PEAK_GBS}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, tmp, info, model)

    fp32_cmp = None
    if rank == 0:
        from neural_lam_amd._lib import lib as _nlam_lib
        mfma_mode = "bf16x3" if _nlam_lib.nlam_mfma_mode() else "fp32"
        if (world == 1 and mfma_mode != "fp32" and not args.no_fp32_compare
                and not args.no_cpu_baseline):
            # the same step with exact fp32 MFMA (the mode is fixed per process): reported
            # next to the default so that the split-bf16 gain is visible in the bench line
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--batch", str(args.batch),
                   "--ar-steps", str(args.ar_steps), "--hidden-dim", str(args.hidden_dim),
                   "--processor-layers", str(args.processor_layers), "--model", args.model,
                   "--no-cpu-baseline", "--no-kernel-timing", "--no-fp32-compare"]
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
        if roofline is not None and roofline.get("bound") == "mfma":
            roofline["note"] = ("achieved = algorithmic fp32 flops / s; peak = exact-fp32 MFMA "
                                "rate" + ("; products run as 3 bf16 MFMA terms (bf16x3)"
                                          if mfma_mode == "bf16x3" else ""))
        upd_per_layer = rec_updates_per_layer(args, info)
        ms = elapsed / args.steps * 1e3
        value = world * B * T * args.processor_layers * upd_per_layer / (elapsed / args.steps)
        out = {
            "metric": "mesh node-updates/sec (fwd+bwd)", "value": value,
            "unit": "mesh node-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
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
            # processor updates + g2m (mesh receivers) + m2g (grid receivers)
            # [+ Hi-LAM init/read-out sweeps] per AR step
            "all_receiver_updates_per_s": world * B * T * all_receiver_updates(args, info)
            / (elapsed / args.steps),
            "hip_graph": bool(graphed is not None and graphed.graph is not None),
            "roofline": roofline, "scatter_add_roofline": scatter, "cpu_baseline": cpu,
            "kernels": kernels,
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    tmpdir.cleanup()


if __name__ == "__main__":
    main()
