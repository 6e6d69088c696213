"""Diffusion schedules and the fp32 buffers derived from them.

Mirrors models/ETDiff/utils.py:75-87 (linear / cosine betas in float64) and the buffer block of
models/ETDiff/mixed_diffusion.py:103-158 (= gaussian_diffusion.py:40-72 for the Gaussian subset): same
torch float64 operations in the same order, cast to float32 at the end, so the tables -- which live in
the checkpoint -- come out bit-identical.  Host-side, init-only.
"""
import math

import torch


def linear_beta_schedule(timesteps):
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps, s=0.008):
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def _log_1_min_a(a):
    return torch.log(1 - a.exp() + 1e-40)


def make_tables(timesteps, beta_schedule="cosine"):
    """-> ordered dict name -> (T,) float32 tensor, in the reference's registration order."""
    if beta_schedule == "linear":
        betas = linear_beta_schedule(timesteps)
    elif beta_schedule == "cosine":
        betas = cosine_beta_schedule(timesteps)
    else:
        raise ValueError(f"unknown beta schedule {beta_schedule}")
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.nn.functional.pad(ac[:-1], (1, 0), value=1.0)
    log_alpha = torch.log(alphas)
    log_cum = torch.cumsum(log_alpha, dim=0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    out = {
        "betas": betas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev,
        "log_alpha": log_alpha, "log_1_min_alpha": _log_1_min_a(log_alpha),
        "log_cumprod_alpha": log_cum, "log_1_min_cumprod_alpha": _log_1_min_a(log_cum),
        "sqrt_alphas_cumprod": torch.sqrt(ac), "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac), "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": pv, "posterior_log_variance_clipped": torch.log(pv.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "loss_weight": torch.ones_like(ac),
    }
    return {k: v.to(torch.float32) for k, v in out.items()}
