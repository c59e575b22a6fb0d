"""GPU tests of the training-step glue: flat AdamW kernel vs torch.optim.AdamW,
FlatParams packing, and a short GraphLAM training run whose loss decreases."""
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flat_adamw_matches_torch():
    from neural_lam_amd import parallel

    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Linear(13, 3))
    mine = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.Linear(13, 3))
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda()
    flat = parallel.FlatParams(mine)
    opt = parallel.FlatAdamW(flat, lr=1e-2)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.9, 0.95))
    gen = torch.Generator().manual_seed(1)
    for _ in range(5):
        x = torch.randn(11, 7, generator=gen)
        topt.zero_grad()
        ref(x).pow(2).sum().backward()
        topt.step()
        flat.zero_grad()
        mine(x.cuda()).pow(2).sum().backward()
        flat.pack_grads()
        opt.step()
    for (k, a), (_, b) in zip(ref.state_dict().items(), mine.state_dict().items()):
        assert torch.allclose(a, b.cpu(), rtol=1e-5, atol=1e-6), k


def test_graphlam_training_loss_decreases():
    from neural_lam_amd import graphgen, parallel, synthetic
    from neural_lam_amd.models import GraphLAM
    import numpy as np

    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(30, 28, 5000.0), None, False)
        n = info["num_grid"]
        gen = torch.Generator().manual_seed(0)
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(n, 1, generator=gen).numpy(), np.zeros(5), np.ones(5), np.zeros(5),
            np.ones(5), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        torch.manual_seed(1)
        model = GraphLAM(synthetic.model_args(graph="g", hidden_dim=64, processor_layers=2),
                         config=None, datastore=ds).cuda()
    flat = parallel.FlatParams(model)
    opt = parallel.FlatAdamW(flat, lr=2e-3)
    batch = synthetic.random_batch(2, 2, n, n_state=5, n_forcing_window=6, device="cuda")
    losses = []
    for _ in range(12):
        flat.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        flat.pack_grads()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.8 * losses[0], losses


def test_ar_checkpointing_matches_plain_rollout():
    """args.ar_checkpoint recomputes each predict_step in backward: same loss
    and the same parameter gradients as the plain BPTT rollout (ar_model.py:220-267)."""
    from neural_lam_amd import graphgen, synthetic
    from neural_lam_amd.models import GraphLAM
    import numpy as np

    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(30, 28, 5000.0), None, False)
        n = info["num_grid"]
        gen = torch.Generator().manual_seed(0)
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(n, 1, generator=gen).numpy(), np.zeros(5), np.ones(5), np.zeros(5),
            np.ones(5), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        models = []
        for ck in (False, True):
            torch.manual_seed(1)
            models.append(GraphLAM(
                synthetic.model_args(graph="g", hidden_dim=64, processor_layers=2, ar_checkpoint=ck),
                config=None, datastore=ds).cuda())
    batch = synthetic.random_batch(2, 3, n, n_state=5, n_forcing_window=6, device="cuda")
    losses, grads = [], []
    for m in models:
        loss = m.training_step(batch)
        loss.backward()
        losses.append(float(loss))
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
    assert abs(losses[0] - losses[1]) <= 1e-6 * abs(losses[0])
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max() + 1e-30), k
