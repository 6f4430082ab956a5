"""ORACLE (test infrastructure, not product): fp32 CPU restatement of the PLMS sampler loop.

Restates GLIGEN/ldm/models/diffusion/plms.py:25-163 (make_schedule, plms_sampling, p_sample_plms),
the schedule helpers it calls (diffusionmodules/util.py:30-34, 55-83; ddpm.py:19-54) and the
alpha schedule of interface.py:41-75.  The denoiser is passed in as ``eps_fn`` so this file pins
the *sampler* arithmetic independently of the UNet.

Pinned by tests/test_oracle_golden.py against schedule tables and a 10-step latent produced by
the reference's own PLMSSampler (tools/make_goldens.py).
"""
from __future__ import annotations

from typing import Callable, List

import numpy as np
import torch


def alphas_cumprod(timesteps: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.012) -> np.ndarray:
    """make_beta_schedule('linear') in float64 (util.py:31-34) -> cumprod (ddpm.py:21-23), kept as the
    float32 buffer the reference registers (ddpm.py:34)."""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas, axis=0).astype(np.float32)


def make_schedule(S: int, num_ddpm: int = 1000, acp: np.ndarray | None = None):
    """ddim_timesteps = arange(0, T, T//S) + 1 (util.py:55-69); alphas = acp[t];
    alphas_prev = [acp[0], acp[t[:-1]]] (util.py:72-83); sigmas = 0 (eta = 0, plms.py:26-27)."""
    acp = alphas_cumprod(num_ddpm) if acp is None else acp
    c = num_ddpm // S
    ts = np.asarray(list(range(0, num_ddpm, c))) + 1
    a = acp[ts]
    a_prev = np.asarray([acp[0]] + acp[ts[:-1]].tolist())
    return dict(ddim_timesteps=ts, ddim_alphas=a, ddim_alphas_prev=a_prev,
                ddim_sqrt_one_minus_alphas=np.sqrt(1.0 - a), ddim_sigmas=np.zeros_like(a))


def alpha_generator(length: int, type=None) -> List[float]:
    """interface.py:41-75."""
    if type is None:
        type = [1, 0, 0]
    assert len(type) == 3 and type[0] + type[1] + type[2] == 1
    n0 = int(type[0] * length)
    n1 = int(type[1] * length)
    n2 = length - n0 - n1
    decay = list(np.arange(start=0, stop=1, step=1 / n1)[::-1]) if n1 != 0 else []
    out = [1] * n0 + decay + [0] * n2
    assert len(out) == length
    return out


def plms_sample(eps_fn: Callable[[torch.Tensor, torch.Tensor, int, float], torch.Tensor], x: torch.Tensor, S: int,
                alpha_type=None, sched=None) -> torch.Tensor:
    """PLMS loop with sigma = 0.

    ``eps_fn(x, t_long[B], step_index_i, fuser_alpha)`` must return the *guided* epsilon
    e_u + s (e_c - e_u) (plms.py:115-124); ``step_index_i`` counts loop iterations from 0 so the
    caller can apply the first-conv switch of plms.py:86-87.
    """
    sched = make_schedule(S) if sched is None else sched
    ts = sched["ddim_timesteps"]
    time_range = np.flip(ts)
    total = ts.shape[0]
    alphas = alpha_generator(len(time_range), alpha_type)
    b = x.shape[0]
    old_eps: List[torch.Tensor] = []

    def x_prev_of(xc, e, index):
        # get_x_prev_and_pred_x0 (plms.py:126-140) with sigma = 0.  The three scalar square roots are taken
        # with numpy float32 (IEEE correctly rounded, = what the reference gets on its GPU): torch's
        # vectorised CPU sqrt is 1 ulp off on some hosts, which would make this checker platform-dependent.
        a_t = np.float32(sched["ddim_alphas"][index])
        a_prev = np.float32(sched["ddim_alphas_prev"][index])
        s1m = torch.full((b, 1, 1, 1), float(np.float32(sched["ddim_sqrt_one_minus_alphas"][index])))
        sqrt_at = torch.full((b, 1, 1, 1), float(np.sqrt(a_t)))
        sqrt_aprev = torch.full((b, 1, 1, 1), float(np.sqrt(a_prev)))
        dir_coef = torch.full((b, 1, 1, 1), float(np.sqrt(np.float32(1.0) - a_prev)))
        pred_x0 = (xc - s1m * e) / sqrt_at
        dir_xt = dir_coef * e
        return sqrt_aprev * pred_x0 + dir_xt

    for i, step in enumerate(time_range):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        t_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        e_t = eps_fn(x, t, i, alphas[i])
        if len(old_eps) == 0:
            x_mid = x_prev_of(x, e_t, index)
            e_next = eps_fn(x_mid, t_next, i, alphas[i])
            e_prime = (e_t + e_next) / 2
        elif len(old_eps) == 1:
            e_prime = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        x = x_prev_of(x, e_prime, index)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
    return x
