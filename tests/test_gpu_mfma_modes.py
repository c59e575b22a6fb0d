"""Both GEMM arithmetic modes of the fused kernels (NLAM_MFMA=fp32 | bf16x3, read once per
process) hold the model-level parity bars against the reference goldens; each mode runs in
its own process.  bf16x3 must also stay an order of magnitude inside the bars."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# fp32 / bf16x3: an order of magnitude inside the 1e-4 / 2e-3 bars, at hidden 64 AND 128 (the
# d = 128 golden runs the wide kernels in bf16x3 and the generic exact-fp32 kernels in fp32 mode).
# bf16: plain bf16 products, fp32 accumulate; only the d = 128 kernels change arithmetic in that
# mode.  Bars 1e-2 / 6e-2 (measured 2e-3 / 3e-2; SURVEY.md 8c allows 2e-2 on the forward).
@pytest.mark.parametrize("mode,pred_bar,grad_bar",
                         [("fp32", 2e-6, 1e-5), ("bf16x3", 2e-5, 2e-4), ("bf16", 1e-2, 6e-2)])
def test_model_parity_in_mode(mode, pred_bar, grad_bar):
    env = dict(os.environ, NLAM_MFMA=mode)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_margin.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert f"mfma mode: {mode}" in out.stdout
    rows = re.findall(r"pred (\S+)\s+loss (\S+)\s+worst grad (\S+)", out.stdout)
    assert len(rows) >= 3, out.stdout
    for pred, loss, grad in rows:
        loss_bar = 1e-5 if mode != "bf16" else 1e-2
        assert float(pred) < pred_bar and float(loss) < loss_bar and float(grad) < grad_bar, out.stdout


@pytest.mark.parametrize("mode,fwd_bar,grad_bar", [("bf16x3", 1e-4, 1e-3), ("bf16", 1e-2, 5e-2)])
def test_operator_parity_wide_widths_in_mode(mode, fwd_bar, grad_bar):
    """InteractionNet at hidden 128 (wide kernels) and 256 (feature-split kernels; BASELINE
    configs[4] width) against the reference's fp32 goldens: fp32 bars in the default mode;
    forward 1e-2 / gradients 5e-2 with plain bf16 products (NLAM_MFMA=bf16; measured 4e-3 / 6e-3)."""
    env = dict(os.environ, NLAM_MFMA=mode)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_op.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert f"mfma mode: {mode}" in out.stdout
    rows = re.findall(r"d(\d+)\s+fwd (\S+)\s+input grads (\S+)\s+param grads (\S+)", out.stdout)
    assert {int(r[0]) for r in rows} >= {128, 256}, out.stdout
    for d, fwd, gin, gpar in rows:
        assert float(fwd) < fwd_bar and float(gin) < grad_bar and float(gpar) < grad_bar, out.stdout


@pytest.mark.parametrize("mode,width", [("bf16", 256), ("bf16", 128), ("bf16x3", 256)])
def test_wide_cases_in_mode(mode, width):
    """Every operator / MLP case of test_gpu_wide.py against the CPU oracle at hidden 256 (the
    feature-split kernels of csrc/fused_fs.hip, in bf16 arithmetic and in the default split-bf16
    arithmetic at the fp32 bars 1e-4 / 1e-3) and at hidden 128 in bf16 arithmetic; a GraphLAM, a 2-level and a 3-level / 2-processor-layer Hi-LAM training step
    against the oracle; and the reference's own CPU bf16-autocast outputs (tests/golden/
    op_d{128,256}_sum_upd_bf16.pt, model_hilam_3level_d128_bf16.pt).  Bars: forward 1e-2,
    gradients 5e-2 of max|ref|."""
    env = dict(os.environ, NLAM_MFMA=mode, NLAM_WIDE_D=str(width))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_wide.py")],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    assert f"mfma mode: {mode} width: {width}" in out.stdout
    assert "all wide cases passed" in out.stdout
    if mode == "bf16":   # (the autocast fixtures are the bf16 mode's pin)
        assert f"ok autocast golden op d{width}" in out.stdout
    assert "levels=3 layers=2" in out.stdout


@pytest.mark.parametrize("env", [{"NLAM_K16": "0"}, {"NLAM_INET_SEQ": "0"}, {"NLAM_K16": "63"}],
                         ids=["32-row-kernels", "launch-by-launch", "no-node-chain"])
def test_alternate_kernel_selections_hold_the_goldens(env):
    """The kernel families are selectable per process (NLAM_K16: 16-row vs 32-row forms, node
    chain; NLAM_INET_SEQ: C++ sequencer vs one ctypes call per launch): every selection must hold
    the operator and model goldens, not only the default one."""
    out = subprocess.run(
        [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
         os.path.join(ROOT, "tests", "test_gpu_interaction_net.py"),
         os.path.join(ROOT, "tests", "test_gpu_models.py")],
        env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2500:], out.stderr[-1500:])
    assert " passed" in out.stdout and "failed" not in out.stdout


def test_precision_is_selectable_per_run_in_one_process():
    """The reference takes `--precision` per run (train_model.py:72-77,285); here
    neural_lam_amd.set_precision() switches the GEMM arithmetic between runs of ONE process:
    the hidden-128 operator golden is held at the fp32 bars in both fp32-grade modes and at the
    bf16 bars in bf16-mixed, the three forwards differ from one another (the switch is real), and
    the hidden-64 operator golden (split-bf16 in every bf16 mode, exact MFMA in fp32-exact) holds
    the fp32 bars throughout."""
    import torch

    import neural_lam_amd
    from conftest import GOLDEN, load_fixture
    from neural_lam_amd.interaction_net import InteractionNet

    def rel(a, b):
        return float((a.detach().cpu() - b).abs().max() / (b.abs().max() + 1e-30))

    def run(fx):
        net = InteractionNet(fx["edge_index"], fx["d"], **fx["kwargs"])
        net.load_state_dict(fx["state_dict"], strict=True)
        net = net.cuda()
        s = fx["send"].cuda().requires_grad_(True)
        r = s if fx["shared"] else fx["rec"].cuda().requires_grad_(True)
        e = fx["edge"].cuda().requires_grad_(True)
        o_rec, o_edge = net(s, r, e)
        ((o_rec * fx["cot_rec"].cuda()).sum() + (o_edge * fx["cot_edge"].cuda()).sum()).backward()
        fwd = max(rel(o_rec, fx["out_rec"]), rel(o_edge, fx["out_edge"]))
        grad = max([rel(s.grad, fx["grad_send"]), rel(e.grad, fx["grad_edge"])] +
                   [rel(p.grad, fx["grad_params"][k]) for k, p in net.named_parameters()])
        return fwd, grad, o_rec.detach().clone()

    fx128 = load_fixture(os.path.join(GOLDEN, "op_d128_sum_upd.pt"))
    fx64 = load_fixture(os.path.join(GOLDEN, "op_d64_sum_upd.pt"))
    start = neural_lam_amd.get_precision()
    outs = {}
    try:
        for prec, fbar, gbar in (("fp32-exact", 1e-4, 1e-3), ("bf16-mixed", 1e-2, 5e-2), ("32-true", 1e-4, 1e-3)):
            neural_lam_amd.set_precision(prec)
            assert neural_lam_amd.get_precision() == prec
            f, g, o = run(fx128)
            assert f < fbar and g < gbar, (prec, f, g)
            outs[prec] = o
            f64, g64, _ = run(fx64)
            assert f64 < 1e-4 and g64 < 1e-3, (prec, f64, g64)
    finally:
        neural_lam_amd.set_precision(start)
    assert not torch.equal(outs["fp32-exact"], outs["32-true"])
    assert not torch.equal(outs["bf16-mixed"], outs["32-true"])
    # the bf16 run is the coarse one: visibly further from the exact run than split-bf16 is
    assert rel(outs["bf16-mixed"], outs["fp32-exact"].cpu()) > 10 * rel(outs["32-true"], outs["fp32-exact"].cpu())
