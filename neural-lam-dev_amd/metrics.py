"""Metrics of reference neural_lam/metrics.py: the training losses on the hot path (wmse,
mse: :21-108; the fused kernels of glue.MaskedWMSE take over in training_step) and the
evaluation metrics of the validation / test loops (wmae, mae, nll, crps_gauss: :111-237).
Plain tensor expressions on whatever device the tensors live on."""
import math

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


def wmae(pred, target, pred_std, mask=None, average_grid=True, sum_vars=True):
    """metrics.py:111-139: absolute error weighted by 1/std."""
    entry = (pred - target).abs() / pred_std
    return mask_and_reduce_metric(entry, mask, average_grid, sum_vars)


def mae(pred, target, pred_std, mask=None, average_grid=True, sum_vars=True):
    """metrics.py:142-163."""
    return wmae(pred, target, torch.ones_like(pred_std), mask, average_grid, sum_vars)


def nll(pred, target, pred_std, mask=None, average_grid=True, sum_vars=True):
    """metrics.py:166-190: -log N(target; pred, pred_std^2) per entry."""
    z = (target - pred) / pred_std
    entry = 0.5 * z * z + torch.log(pred_std) + 0.5 * math.log(2.0 * math.pi)
    return mask_and_reduce_metric(entry, mask, average_grid, sum_vars)


def crps_gauss(pred, target, pred_std, mask=None, average_grid=True, sum_vars=True):
    """metrics.py:193-227: closed-form (negative) CRPS of a Gaussian forecast,
    sigma * ( z (2 Phi(z) - 1) + 2 phi(z) - 1/sqrt(pi) ),  z = (target - pred) / sigma."""
    z = (target - pred) / pred_std
    phi = torch.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)
    cdf = 0.5 * (1.0 + torch.erf(z / math.sqrt(2.0)))
    entry = pred_std * (z * (2.0 * cdf - 1.0) + 2.0 * phi - 1.0 / math.sqrt(math.pi))
    return mask_and_reduce_metric(entry, mask, average_grid, sum_vars)


DEFINED_METRICS = {"mse": mse, "mae": mae, "wmse": wmse, "wmae": wmae, "nll": nll,
                   "crps_gauss": crps_gauss}


def get_metric(metric_name):
    name = metric_name.lower()
    assert name in DEFINED_METRICS, f"Unknown metric: {metric_name}"
    return DEFINED_METRICS[name]
