import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_fixture(path):
    """torch.load of a golden fixture.  The *_bf16 fixtures (reference run under CPU bf16
    autocast) hold outputs only and name the fixture whose weights and inputs they used."""
    import torch

    fx = torch.load(path, weights_only=False)
    if fx.get("base"):
        base = torch.load(os.path.join(GOLDEN, fx["base"] + ".pt"), weights_only=False)
        for k, v in base.items():
            fx.setdefault(k, v)
    return fx


def elementwise_excess(a, b, tol):
    """Worst |a - b| / (tol * (|b| + rms(b))) over the elements: < 1 means every element is
    inside a tolerance scaled by ITS OWN magnitude plus the tensor's rms -- the max-norm bars
    (error / max|ref|) say nothing about the small-magnitude entries of a tensor; this one does.
    rms(b) (not max|b|) is the floor: typically 3-10 x smaller."""
    import torch

    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    rms = float(b.pow(2).mean().sqrt())
    return float(((a - b).abs() / (tol * (b.abs() + rms) + 1e-300)).max())


def l2_rel(a, b):
    """||a - b||_2 / ||b||_2."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def fixture_files(pattern, autocast=False):
    """Golden files matching `pattern`: the fp32 ones, or the bf16-autocast ones."""
    import glob

    files = sorted(glob.glob(os.path.join(GOLDEN, pattern)))
    return [f for f in files if f.endswith("_bf16.pt") == autocast]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so that a
    plain `pytest tests/` works in the CPU-only build container."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
