"""The two metrics the reference calls inside every train / val / test step
(bayesrul/results/metrics.py:210-274, called at models/bayesian.py:159-160) plus `nasa_score`.
Device-agnostic: the reference's `torch.linspace(..., device=y_true.get_device())` raises on CPU
tensors; here devices come from the tensors themselves.  No host sync: the
reference's `assert y_std.min() >= 0` (a D2H sync per step) is an explicit opt-in."""
from __future__ import annotations

import torch
from torch import Tensor


def sharpness(sigma_hat: Tensor) -> Tensor:
    """metrics.py:210-213"""
    return torch.sqrt(torch.square(sigma_hat).mean())


def get_proportion_lists(y_pred: Tensor, y_std: Tensor, y_true: Tensor, num_bins: int, prop_type: str = "interval"):
    """metrics.py:216-252"""
    dev, dt = y_true.device, y_pred.dtype
    exp_proportions = torch.linspace(0, 1, num_bins, device=dev, dtype=dt)
    residuals = y_pred - y_true
    normalized_residuals = (residuals.flatten() / y_std.flatten()).reshape(-1, 1)
    dist = torch.distributions.Normal(torch.zeros(1, device=dev, dtype=dt), torch.ones(1, device=dev, dtype=dt))
    if prop_type == "interval":
        lower = dist.icdf(0.5 - exp_proportions / 2.0)
        upper = dist.icdf(0.5 + exp_proportions / 2.0)
        within = (normalized_residuals >= lower) * (normalized_residuals <= upper)
        obs_proportions = torch.sum(within, dim=0).flatten() / len(residuals)
    elif prop_type == "quantile":
        bound = dist.icdf(exp_proportions)
        obs_proportions = torch.sum(normalized_residuals <= bound, dim=0).flatten() / len(residuals)
    else:
        raise ValueError(prop_type)
    return exp_proportions, obs_proportions


def rms_calibration_error(y_pred: Tensor, y_std: Tensor, y_true: Tensor, num_bins: int = 100,
                          prop_type: str = "interval", check_positive: bool = False) -> Tensor:
    """metrics.py:255-274"""
    assert y_pred.shape == y_std.shape == y_true.shape
    if check_positive:
        assert y_std.min() >= 0, "Not all values are positive"
    exp_props, obs_props = get_proportion_lists(y_pred, y_std, y_true, num_bins, prop_type)
    return torch.sqrt(torch.mean(torch.square(exp_props - obs_props)))


def nasa_score(y_true: Tensor, y_pred: Tensor) -> Tensor:
    """metrics.py:205-207"""
    d = y_pred - y_true
    return torch.where(d > 0, torch.exp(d / 10) - 1, torch.exp(-d / 13) - 1)
