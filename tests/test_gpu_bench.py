"""The bench line's contract (one JSON object on the last stdout line): the fields the driver
reads, the roofline object of the dominant kernel, the CPU baseline object and the stand-alone
scatter-add figure the north star names.  A short run (3 timed steps) in a process of its own."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
         "--windows", "1", "--no-fp32-compare"],
        capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    raw = out.stdout.strip().splitlines()[-1]
    # the driver keeps a 9 kB tail of stdout: the whole line has to fit into it
    assert len(raw) < 8192, len(raw)
    line = json.loads(raw)
    for key, want in (("metric", "mesh node-updates/sec (fwd+bwd)"), ("unit", "mesh node-updates/s"),
                      ("n_gpus", 1), ("steps", 3), ("warmup", 1), ("higher_is_better", True),
                      ("scaling", "weak"), ("vs_baseline", None), ("data", "synthetic")):
        assert line[key] == want, (key, line[key])
    assert line["value"] > 1e6 and line["ms_per_step"] > 0
    # value = B * ar_steps * P * N_mesh / t_step
    # (floats of the line carry 5 significant digits)
    assert abs(line["value"] - 4 * 1 * 4 * 6561 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-4
    assert "graph_lam" in line["config"]["workload"] and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert 0 < roof["frac"] <= 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert cpu["unit"] == line["unit"]
    # the m2m scatter-add on its own (north star: >= 50 % of the HBM roof; measured 0.67)
    sc = line["scatter_add_roofline"]
    assert sc["bound"] == "hbm" and sc["frac"] > 0.4, sc
    assert line["launch"]["mode"] == "hip_graph"
    # the other BASELINE configs, run as child processes of the same bench invocation
    other = line["other_configs"]
    assert set(other) == {n for n, _a, _e in _bench_module().OTHER_CONFIGS}
    for name, ent in other.items():
        assert "error" not in ent, (name, ent)
        assert ent["ms_per_step"] > 0 and ent["value"] > 0 and ent["roofline"]["kernel"], (name, ent)
        assert 0 < ent["roofline"]["frac"] <= 1
    assert other["hi_lam-256 bf16 (configs[4], per GPU)"]["dtype"] == "bf16"
    # configs[2] carries its own stated CPU baseline (B = 1, bounded)
    cb = other["hi_lam-128 (configs[2])"]["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    # the per-kernel table: top entries in the line, all of them in the side file
    top = line["kernels_top"]
    assert top["columns"] == ["calls", "ms", "gflop", "mb"] and 1 <= len(top["rows"]) <= 12
    side = json.load(open(os.path.join(ROOT, line["kernels_file"])))
    assert set(top["rows"]) <= set(side["kernels"]) and len(side["kernels"]) >= len(top["rows"])
    assert roof["kernel"] in side["kernels"]
    assert other["graph_lam-64 ar_steps=4 (configs[3], per GPU)"]["ar_steps"] == 4


def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_two_rank_bench_control_flow_over_gloo():
    """`bench.py --gpus 2` on ONE card with the gloo rehearsal backend: the spawn / rendezvous /
    barrier / MAX-over-ranks control flow of the multi-GPU bench (the RCCL run itself needs two
    cards; reference DDP: train_model.py:279)."""
    env = dict(os.environ, NLAM_BENCH_BACKEND="gloo")
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
         "--windows", "1", "--no-cpu-baseline"],
        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo"
    assert line["config"]["global_batch"] == 8 and line["config"]["parallelism"] == "dp2"
    assert line["launch"]["mode"] in ("eager", "eager+overlapped_allreduce")
    assert line["scaling"] == "weak" and line["other_configs"] is None
    # value = world * B * ar_steps * P * N_mesh / t_step
    assert abs(line["value"] - 2 * 4 * 1 * 4 * 6561 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-4


def test_gpus_1_is_the_default_path():
    """`--gpus 1` is bit-for-bit the default invocation (same workload, same launch mode, no
    process group): SCALE's N = 1 point is the BENCH line."""
    outs = []
    for extra in ([], ["--gpus", "1"]):
        out = subprocess.run(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--windows",
             "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-other-configs"] + extra,
            capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        outs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    a, b = outs
    for key in ("metric", "unit", "n_gpus", "config", "launch", "hip_graph", "rccl_ranks", "backend",
                "dtype", "loss"):
        assert a[key] == b[key], key


@pytest.mark.parametrize("model_args,overlap", [
    ((), False),                                            # GraphLAM-64: 0.86 MB, one collective
    (("--model", "hi_lam", "--hidden-dim", "128"), True),   # 22 MB: buckets issued from hooks
])
def test_single_rank_rccl_rehearsal(model_args, overlap):
    """The RCCL ("nccl") branch of bench.py EXECUTED on the one card of the test box: a one-rank
    process group on the device, parameter broadcast, bucket all-reduces (hook-issued on the side
    stream for the 22 MB payload), the eager-vs-graph probe, barriers and the MAX over ranks.
    (Bit-identity of a step under one-rank collectives: tests/test_gpu_dp.py.)"""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
            "--windows", "1", "--no-fp32-compare", "--no-cpu-baseline", "--no-other-configs",
            "--no-kernel-timing", *model_args]
    lines = []
    for env_extra in ({"NLAM_BENCH_SINGLE_RANK_RCCL": "1"}, {}):
        env = {k: v for k, v in os.environ.items() if k != "NLAM_BENCH_SINGLE_RANK_RCCL"}
        env.update(env_extra)
        out = subprocess.run(base, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines.append(json.loads(out.stdout.strip().splitlines()[-1]))
    rccl, plain = lines
    assert rccl["backend"] == "nccl" and rccl["rccl_ranks"] == 1 and rccl["n_gpus"] == 1
    assert rccl["single_rank_rccl"] is True and plain["backend"] is None
    assert rccl["grad_allreduce"]["overlap_with_backward"] is overlap
    # both schedules of a multi-rank step ran (and were timed) under RCCL
    probe = rccl["launch"]["probe_ms"]
    assert probe["eager_overlap"] > 0 and probe["hip_graph_trailing_allreduce"] > 0
    assert rccl["launch"]["mode"] in ("hip_graph+trailing_allreduce", "eager+overlapped_allreduce",
                                      "eager")
    # (the probe runs extra optimiser steps before the timed ones: compare the step COUNT first)
    assert rccl["ms_per_step"] > 0 and rccl["loss"] == rccl["loss"]
    assert rccl["value"] > 0.5 * plain["value"], (rccl["ms_per_step"], plain["ms_per_step"])
