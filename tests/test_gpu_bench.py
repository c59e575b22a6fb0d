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
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key, want in (("metric", "mesh node-updates/sec (fwd+bwd)"), ("unit", "mesh node-updates/s"),
                      ("n_gpus", 1), ("steps", 3), ("warmup", 1), ("higher_is_better", True),
                      ("scaling", "weak"), ("vs_baseline", None), ("data", "synthetic")):
        assert line[key] == want, (key, line[key])
    assert line["value"] > 1e6 and line["ms_per_step"] > 0
    # value = B * ar_steps * P * N_mesh / t_step
    assert abs(line["value"] - 4 * 1 * 4 * 6561 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6
    assert "graph_lam" in line["config"]["workload"] and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert 0 < roof["frac"] <= 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert cpu["unit"] == line["unit"]
    # the m2m scatter-add on its own (north star: >= 50 % of the HBM roof; measured 0.67)
    sc = line["scatter_add_roofline"]
    assert sc["bound"] == "hbm" and sc["frac"] > 0.4, sc
    assert line["launch"]["mode"] == "hip_graph"
