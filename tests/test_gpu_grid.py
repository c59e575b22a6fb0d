"""The grid-side encoder chain of predict_step in one pass (csrc/fused16_grid.hip; reference
models/base_graph_model.py:116-143,157 and the grid-row thirds of interaction_net.py:121):
(a) the kernel against the launch-by-launch sequence it replaces -- concat_rows -> grid_embedder
-> g2m sender projection -> encoding MLP (+ residual) -> m2g receiver projection -- BITWISE (same
building blocks, same order of operations); (b) the same against a plain fp32 torch restatement;
(c) a whole training step with the fused pass on and off: identical loss and gradients, fewer
launches.  The full-size oracle tests (tests/test_gpu_fullsize.py) run with the pass on."""
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.detach().cpu() - b.detach().cpu()).abs().max() / (b.detach().abs().max().cpu() + 1e-30))


def _mlp(k):
    from neural_lam_amd.utils import make_mlp

    return make_mlp([k, 64, 64]).cuda()


def _params(m):
    from neural_lam_amd.fused import _mlp_parts

    lin, ln = _mlp_parts(m)
    return [t.detach() for t in (lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, ln.weight, ln.bias)]


@pytest.mark.parametrize("widths,B,N", [((17, 17, 18, 4), 3, 1000), ((17, 17, 18, 4), 2, 16 * 40 + 5),
                                        ((5, 5, 6, 4), 2, 333), ((32, 32), 1, 47), ((1, 8, 3), 4, 16),
                                        ((9, 9, 12, 2), 2, 90)])
def test_grid_encode_kernel_matches_the_launch_sequence_bitwise(widths, B, N):
    from neural_lam_amd import glue, ops
    from neural_lam_amd.ops import mat

    torch.manual_seed(60)
    gen = torch.Generator(device="cuda").manual_seed(61)
    k_in = sum(widths)
    # sources as predict_step hands them over: slices of larger batch tensors (batch pitch > N * w),
    # the last one batch-invariant (the static features)
    big = [torch.randn(B, 2, N, w, device="cuda", generator=gen) for w in widths[:-1]]
    srcs = [t[:, 1] for t in big]
    static = torch.randn(N, widths[-1], device="cuda", generator=gen)
    srcs.append(static.unsqueeze(0).expand(B, -1, -1) if len(widths) > 1 else
                torch.randn(B, N, widths[-1], device="cuda", generator=gen))
    emb_m, enc_m = _mlp(k_in), _mlp(64)
    Wg2m = torch.randn(64, 192, device="cuda", generator=gen) * 0.1
    Wm2g = torch.randn(64, 192, device="cuda", generator=gen) * 0.1
    bm2g = torch.randn(64, device="cuda", generator=gen) * 0.1
    Ws, Wr = Wg2m[:, 64:128], Wm2g[:, 128:]

    from neural_lam_amd.fused import _base

    feat, emb, ps, rep, pr = (torch.empty(B, N, c, device="cuda") for c in (k_in, 64, 64, 64, 64))
    ops.grid_encode_fwd([mat(_base(t)) for t in srcs], _params(emb_m), Ws, _params(enc_m), Wr, bm2g,
                        feat, emb, ps, rep, pr)
    # ---- launch by launch
    with torch.no_grad():
        feat2 = glue.ConcatRows.apply(*srcs)
        emb2 = emb_m(feat2)
        ps2, pr2 = torch.empty_like(emb2), torch.empty_like(emb2)
        ops.fused_lin_fwd(mat(emb2), Ws, None, None, None, mat(ps2))
        rep2 = enc_m(emb2, res=emb2)
        ops.fused_lin_fwd(mat(rep2), Wr, bm2g, None, None, mat(pr2))
    assert torch.equal(feat, torch.cat(srcs, dim=-1)) and torch.equal(feat, feat2)
    for name, a, b in (("emb", emb, emb2), ("ps", ps, ps2), ("rep", rep, rep2), ("pr", pr, pr2)):
        assert torch.equal(a, b), (name, rel(a, b))
    # ---- plain fp32 torch restatement (forward bar 1e-4)
    with torch.no_grad():
        x = torch.cat(srcs, dim=-1).double()

        def mlp(m, v):
            W1, b1, W2, b2, g, bt = [t.double() for t in _params(m)]
            h = torch.nn.functional.silu(v @ W1.T + b1)
            return torch.nn.functional.layer_norm(h @ W2.T + b2, (64,), g, bt, 1e-5)

        e_ = mlp(emb_m, x)
        r_ = e_ + mlp(enc_m, e_)
        assert rel(emb, e_.float()) < 1e-4 and rel(ps, (e_ @ Ws.double().T).float()) < 1e-4
        assert rel(rep, r_.float()) < 1e-4 and rel(pr, (r_ @ Wr.double().T + bm2g.double()).float()) < 1e-4


@pytest.mark.parametrize("model_name", ["graph_lam", "hi_lam"])
def test_training_step_is_unchanged_by_the_fused_grid_pass(model_name, monkeypatch):
    """One training step (ar_steps = 2, B = 2) with the fused grid pass and with the
    launch-by-launch sequence: identical loss and parameter gradients (bitwise: same arithmetic,
    and every backward kernel is the same one reading the same saved tensors), five launches
    fewer per AR step."""
    import numpy as np

    from neural_lam_amd import fused, graphgen, ops, synthetic
    from neural_lam_amd.models import MODELS

    hier = model_name != "graph_lam"
    with tempfile.TemporaryDirectory() as tmp:
        info = graphgen.create_graph(tmp + "/graph/g", graphgen.make_xy(38, 35, 5000.0),
                                     3 if hier else None, hier)
        n = info["num_grid"]
        gen = torch.Generator().manual_seed(0)
        ds = synthetic.SyntheticDatastore(
            tmp, torch.randn(n, 4, generator=gen).numpy(), np.zeros(7), np.ones(7), np.zeros(7),
            np.ones(7), (torch.rand(n, generator=gen) < 0.2).float().numpy(), n_forcing=2)
        torch.manual_seed(1)
        model = MODELS[model_name](synthetic.model_args(graph="g", hidden_dim=64, processor_layers=2),
                                   config=None, datastore=ds).cuda()
    batch = synthetic.random_batch(2, 2, n, n_state=7, n_forcing_window=6, device="cuda")

    def run(on):
        monkeypatch.setenv("NLAM_GRID_ENCODE", "1" if on else "0")
        for p in model.parameters():
            p.grad = None
        ops.PROFILER = ops.KernelProfiler()
        try:
            loss = model.training_step(batch)
            loss.backward()
            stats = ops.PROFILER.collect()
        finally:
            ops.PROFILER = None
        assert not fused.PRE
        return float(loss.detach()), [p.grad.clone() for p in model.parameters()], stats

    l_on, g_on, s_on = run(True)
    l_off, g_off, s_off = run(False)
    assert any(k.startswith("nlam_grid_encode_fwd") for k in s_on)
    assert not any(k.startswith("nlam_grid_encode_fwd") for k in s_off)
    assert not any(k.startswith("nlam_concat_rows") for k in s_on)
    n_on, n_off = (sum(v["calls"] for v in s.values()) for s in (s_on, s_off))
    assert n_on < n_off, (n_on, n_off)
    assert l_on == l_off
    for (k, _), a, b in zip(model.named_parameters(), g_on, g_off):
        assert torch.equal(a, b), (k, rel(a, b))
