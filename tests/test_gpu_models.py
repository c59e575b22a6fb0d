"""Model-level parity on the GPU: GraphLAM / HiLAM / HiLAMParallel drop-ins
(loaded with the reference's state_dict) against the golden vectors captured
from the reference's own training_step: rollout prediction, loss and every
parameter gradient.  fp32 tolerances: prediction 1e-4, loss 1e-4, grads 2e-3
(relative to max|ref| per tensor; SURVEY.md section 8c, rollout T<=3)."""
import glob
import os
import tempfile

import pytest
import torch

from conftest import GOLDEN, fixture_files

pytestmark = pytest.mark.gpu
MODEL_FILES = fixture_files("model_*.pt")   # the bf16-autocast one: tests/test_gpu_mfma_modes.py


def rel(a, b):
    return float((a.detach().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


def build_model(fx, tmp, load_weights=True):
    from neural_lam_amd import graphgen, synthetic
    from neural_lam_amd.models import MODELS

    gi, cfg, data = fx["graph"], fx["cfg"], fx["data"]
    graphgen.create_graph(
        os.path.join(tmp, "graph", "g"), graphgen.make_xy(gi["nx"], gi["ny"], gi["spacing"]),
        gi["n_max_levels"], gi["hierarchical"],
    )
    d_state = data["diff_std"].shape[0]
    # feature weights are uniform 1/d in the fixtures: per_var_std = diff_std*sqrt(d)
    ds = synthetic.SyntheticDatastore(
        tmp, data["grid_static_features"].numpy(), [0.0] * d_state, [1.0] * d_state,
        data["diff_mean"].numpy(), data["diff_std"].numpy(),
        data["boundary_mask"][:, 0].numpy(), n_forcing=2,
    )
    args = synthetic.model_args(
        graph="g", hidden_dim=cfg["hidden_dim"], processor_layers=cfg["processor_layers"],
        mesh_aggr=cfg["mesh_aggr"], loss=cfg["loss"], hidden_layers=cfg.get("hidden_layers", 1),
    )
    model = MODELS[cfg["model"]](args, config=None, datastore=ds)
    if load_weights:
        res = model.load_state_dict(fx["state_dict"], strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    return model


@pytest.mark.parametrize("path", MODEL_FILES, ids=[os.path.basename(p)[:-3] for p in MODEL_FILES])
def test_training_step_vs_reference_golden(path):
    fx = torch.load(path, weights_only=False)
    with tempfile.TemporaryDirectory() as tmp:
        model = build_model(fx, tmp).cuda()
    batch = (fx["init_states"].cuda(), fx["target_states"].cuda(), fx["forcing"].cuda(), None)
    pred, _, _, _ = model.common_step(batch)
    assert rel(pred, fx["prediction"]) < 1e-4
    loss = model.training_step(batch)
    assert abs(float(loss) - fx["loss"]) < 1e-4 * abs(fx["loss"])
    loss.backward()
    from conftest import elementwise_excess, l2_rel

    assert l2_rel(pred, fx["prediction"]) < 1e-4
    assert elementwise_excess(pred, fx["prediction"], 1e-4) < 1.0
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        assert rel(p.grad, fx["grad_params"][k]) < 2e-3, k
        # (2-norm and element-by-element forms of the same bar: conftest.elementwise_excess)
        assert l2_rel(p.grad, fx["grad_params"][k]) < 2e-3, k
        assert elementwise_excess(p.grad, fx["grad_params"][k], 2e-3) < 1.0, k
