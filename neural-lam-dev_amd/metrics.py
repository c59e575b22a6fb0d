"""Training losses on the hot path (reference neural_lam/metrics.py:21-108:
get_metric, mask_and_reduce_metric, wmse, mse)."""
import torch


def mask_and_reduce_metric(metric_entry_vals, mask, average_grid, sum_vars):
    """metrics.py:21-53: keep masked grid nodes, mean over grid (dim -2), then
    sum over variables (dim -1)."""
    if mask is not None:
        metric_entry_vals = metric_entry_vals[..., mask, :]
    if average_grid:
        metric_entry_vals = torch.mean(metric_entry_vals, dim=-2)
    if sum_vars:
        metric_entry_vals = torch.sum(metric_entry_vals, dim=-1)
    return metric_entry_vals


def wmse(pred, target, pred_std, mask=None, average_grid=True, sum_vars=True):
    """metrics.py:56-84: squared error weighted by 1/std^2."""
    entry = (pred - target) ** 2 / (pred_std**2)
    return mask_and_reduce_metric(entry, mask, average_grid, sum_vars)


def mse(pred, target, pred_std, mask=None, average_grid=True, sum_vars=True):
    """metrics.py:87-108."""
    return wmse(pred, target, torch.ones_like(pred_std), mask, average_grid, sum_vars)


DEFINED_METRICS = {"mse": mse, "wmse": wmse}


def get_metric(metric_name):
    name = metric_name.lower()
    assert name in DEFINED_METRICS, f"Unknown metric: {metric_name}"
    return DEFINED_METRICS[name]
