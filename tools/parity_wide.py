#!/usr/bin/env python3
"""Runs the operator / MLP parity cases of tests/test_gpu_wide.py in THIS process's arithmetic
mode and width (NLAM_MFMA, NLAM_WIDE_D) -- e.g. the hidden-256 feature-split kernels, which exist
in bf16 arithmetic only:  NLAM_MFMA=bf16 NLAM_WIDE_D=256 python tools/parity_wide.py
Prints one `ok <case>` line per case; exits non-zero on the first failure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import neural_lam_amd  # noqa: F401  (binds libnlam_hip.so)
from neural_lam_amd._lib import lib
import test_gpu_wide as T

print("mfma mode:", {0: "fp32", 1: "bf16x3", 2: "bf16"}[int(lib.nlam_mfma_mode())], "width:", T.D)
for shared, upd, aggr, B in T.INET_CASES + [(True, True, "mean", 5)]:
    T.test_wide_interaction_net_vs_oracle(shared, upd, aggr, B)
    print(f"ok inet shared={shared} upd={upd} aggr={aggr} B={B}", flush=True)
for shared in (False, True):
    T.test_wide_tiles_of_empty_receivers(shared)
    print(f"ok inet tiles of empty receivers shared={shared}", flush=True)
for shared, upd, aggr in ((True, True, "sum"), (False, False, "mean")):
    T.test_wide_high_in_degree_runs_on_virtual_receivers(shared, upd, aggr)
    print(f"ok inet in-degree > 32 on virtual receivers shared={shared}", flush=True)
for upd, aggr in ((True, "sum"), (False, "mean")):
    T.test_wide_split_mlps_vs_oracle(upd, aggr)
    print(f"ok inet SplitMLPs upd={upd} aggr={aggr}", flush=True)
T.test_wide_stride0_batch_inputs_match_oracle()
print("ok inet stride-0 inputs", flush=True)
for blueprint, ln, res, rows, B in T.MLP_CASES:
    T.test_wide_mlp_vs_oracle(blueprint, ln, res, rows, B)
    print(f"ok mlp {blueprint} ln={ln} res={res} rows={rows} B={B}", flush=True)


def model_case(kind, hierarchical, levels, grid=(30, 28), layers=1):
    """One training step of a small model at hidden T.D against the CPU oracle: loss and every
    parameter gradient (the model wiring is pinned by the reference goldens at hidden 64 / 128;
    this is the same check at this width, in this process's arithmetic)."""
    import tempfile
    import numpy as np
    import torch
    import nlam_oracle as orc
    from neural_lam_amd import graphgen, synthetic, models

    gen = torch.Generator().manual_seed(11)
    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(grid[0], grid[1], 5000.0), levels,
                                     hierarchical)
        ng = info["num_grid"]
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(ng, 1, generator=gen).numpy(), np.zeros(5), np.ones(5), np.zeros(5),
            np.ones(5), (torch.rand(ng, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        torch.manual_seed(2)
        cls = {"graph_lam": models.GraphLAM, "hi_lam": models.HiLAM,
               "hi_lam_parallel": models.HiLAMParallel}[kind]
        model = cls(synthetic.model_args(graph="g", hidden_dim=T.D, processor_layers=layers),
                    config=None, datastore=ds)
        _, graph = orc.load_graph(tmp + "/graph/g")
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()
          if v.dtype.is_floating_point}
    data = {k: getattr(model, k).detach().clone() for k in
            ("grid_static_features", "diff_mean", "diff_std", "boundary_mask", "per_var_std")}
    model = model.cuda()
    batch = synthetic.random_batch(2, 1, ng, n_state=5, n_forcing_window=6, seed=5)
    loss = model.training_step(tuple(t.cuda() if t is not None else None for t in batch))
    loss.backward()
    cfg = {"model": kind, "hidden_layers": 1, "processor_layers": layers, "mesh_aggr": "sum",
           "loss": "wmse"}
    want, _ = orc.training_loss(sd, graph, cfg, data, batch[0], batch[1], batch[2])
    names = [k for k, _ in model.named_parameters()]
    grads = torch.autograd.grad(want, [sd[k] for k in names])
    lerr = abs(float(loss) - float(want)) / abs(float(want))
    gerr = max(T.rel(p.grad, g) for (k, p), g in zip(model.named_parameters(), grads))
    print(f"ok model {kind} d{T.D} levels={levels} layers={layers}: loss rel {lerr:.2e}  "
          f"worst param grad {gerr:.2e}", flush=True)
    assert lerr < T.FWD_BAR and gerr < T.GRAD_BAR, (lerr, gerr)


def autocast_goldens():
    """bf16 mode only: the kernels against what the REFERENCE computes under CPU bf16 autocast
    (tests/golden/*_bf16.pt), same bars.  LayerNorm affine gradients are left to the fp32
    fixtures: torch's CPU kernel accumulates them in bf16 (tests/test_oracle_golden.py)."""
    import tempfile
    import torch
    from conftest import GOLDEN, load_fixture
    from neural_lam_amd.interaction_net import InteractionNet

    def ln_affine(grads, k):
        return grads[k[: k.rfind(".")] + ".weight"].dim() == 1

    path = os.path.join(GOLDEN, f"op_d{T.D}_sum_upd_bf16.pt")
    if os.path.exists(path):
        fx = load_fixture(path)
        net = InteractionNet(fx["edge_index"], fx["d"], **fx["kwargs"])
        net.load_state_dict(fx["state_dict"], strict=True)
        net = net.cuda()
        s = fx["send"].cuda().requires_grad_(True)
        e = fx["edge"].cuda().requires_grad_(True)
        o_rec, o_edge = net(s, s, e)
        ((o_rec * fx["cot_rec"].cuda()).sum() + (o_edge * fx["cot_edge"].cuda()).sum()).backward()
        fwd = max(T.rel(o_rec, fx["out_rec"]), T.rel(o_edge, fx["out_edge"]))
        gin = max(T.rel(s.grad, fx["grad_send"]), T.rel(e.grad, fx["grad_edge"]))
        gpar = max(T.rel(p.grad, fx["grad_params"][k]) for k, p in net.named_parameters()
                   if not ln_affine(fx["grad_params"], k))
        print(f"ok autocast golden op d{T.D}: fwd {fwd:.2e}  input grads {gin:.2e}  "
              f"param grads {gpar:.2e}", flush=True)
        assert fwd < T.FWD_BAR and gin < T.GRAD_BAR and gpar < T.GRAD_BAR, (fwd, gin, gpar)
    path = os.path.join(GOLDEN, f"model_hilam_3level_d{T.D}_bf16.pt")
    if os.path.exists(path):
        from test_gpu_models import build_model

        fx = load_fixture(path)
        with tempfile.TemporaryDirectory() as tmp:
            model = build_model(fx, tmp).cuda()
        batch = (fx["init_states"].cuda(), fx["target_states"].cuda(), fx["forcing"].cuda(), None)
        pred, _, _, _ = model.common_step(batch)
        loss = model.training_step(batch)
        loss.backward()
        perr = T.rel(pred, fx["prediction"])
        lerr = abs(float(loss) - fx["loss"]) / abs(fx["loss"])
        gerr = max(T.rel(p.grad, fx["grad_params"][k]) for k, p in model.named_parameters()
                   if not ln_affine(fx["grad_params"], k))
        print(f"ok autocast golden hi_lam 3 levels d{T.D}: pred {perr:.2e}  loss {lerr:.2e}  "
              f"worst param grad {gerr:.2e}", flush=True)
        assert perr < T.FWD_BAR and lerr < T.FWD_BAR and gerr < T.GRAD_BAR, (perr, lerr, gerr)


model_case("graph_lam", False, None)
model_case("hi_lam", True, None)
# BASELINE configs[2] / [4] structure: 3 mesh levels (needs the 81 x 83 grid), two processor layers
model_case("hi_lam", True, 3, grid=(81, 83), layers=2)
model_case("hi_lam_parallel", True, None)   # SplitMLPs on the wide kernels (wide.apply_inet_split)
if T._BF16:
    autocast_goldens()
print("all wide cases passed")
