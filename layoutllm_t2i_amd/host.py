"""Host-side (CPU, once per image) pieces of the denoising path: integer box rectangles for the
relation injection, the alpha (fuser-scale) schedule and the PLMS schedule tables.

Everything here is tiny scalar/table work that the reference redoes inside its hot loop
(attention.py:321-346 with 4 ``.tolist()`` device syncs per call x 16 layers x 102 forwards;
plms.py:60 per sample() call); it is hoisted out of the loop, result-identically.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def box_rects(boxes: np.ndarray, masks: np.ndarray, h: int, w: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Pixel rectangles of RelationCrossAttention.forward (attention.py:321-346) for one resolution.

    boxes [B, n, 4] float32 (x0, y0, x1, y1 normalised), masks [B, n] float32.
    Returns (rects [B, n, 4] int32 = (top, bottom, left, right) *already normalised to the effective
    python-slice range*, nvalid [B] int32 = number of boxes used before the reference's ``break`` at
    the first padded or degenerate box, poison [B] int32 = 1 where a used box has an empty slice
    (right < left or bottom < top), for which the reference yields NaN over the whole sample).

    Arithmetic mirrors the reference exactly: float32 multiply by the python int, truncation toward
    zero (``.to(torch.int)``), x1/y1 clamped to w/h with torch.minimum, x0/y0 not clamped.
    """
    boxes = np.asarray(boxes, dtype=np.float32)
    masks = np.asarray(masks, dtype=np.float32)
    B, n, _ = boxes.shape
    count = masks.sum(axis=-1)
    x0 = (boxes[:, :, 0] * w).astype(np.int32)
    y0 = (boxes[:, :, 1] * h).astype(np.int32)
    x1 = np.minimum(boxes[:, :, 2] * w, np.float32(w)).astype(np.int32)
    y1 = np.minimum(boxes[:, :, 3] * h, np.float32(h)).astype(np.int32)
    rects = np.zeros((B, n, 4), np.int32)
    nvalid = np.zeros((B,), np.int32)
    poison = np.zeros((B,), np.int32)
    for k in range(B):
        for i in range(n):
            left, right, top, bottom = int(x0[k, i]), int(x1[k, i]), int(y0[k, i]), int(y1[k, i])
            if i < count[k] and left != right and top != bottom:
                t, b, _ = slice(top, bottom).indices(h)     # python slice semantics incl. negatives
                l, r, _ = slice(left, right).indices(w)
                b, r = max(b, t), max(r, l)
                rects[k, i] = (t, b, l, r)
                if (b - t) * (r - l) == 0:
                    poison[k] = 1
                nvalid[k] = i + 1
            else:
                break
    return rects, nvalid, poison


def alpha_generator(length: int, type=None) -> List[float]:
    """Fuser scale per sampling step (interface.py:41-75): [1]*n0 + linear decay + [0]*n2."""
    if type is None:
        type = [1, 0, 0]
    assert len(type) == 3
    assert type[0] + type[1] + type[2] == 1
    n0 = int(type[0] * length)
    n1 = int(type[1] * length)
    n2 = length - n0 - n1
    decay = list(np.arange(start=0, stop=1, step=1 / n1)[::-1]) if n1 != 0 else []
    alphas = [1] * n0 + decay + [0] * n2
    assert len(alphas) == length
    return alphas


def alphas_cumprod(timesteps: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.012) -> np.ndarray:
    """DDPM.register_schedule for the 'linear' schedule (util.py:31-34, ddpm.py:19-35): float64 betas,
    cumprod, stored as float32."""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas, axis=0).astype(np.float32)


def make_schedule(S: int, acp: np.ndarray) -> Dict[str, np.ndarray]:
    """PLMSSampler.make_schedule with eta = 0 (plms.py:25-56; util.py:55-83)."""
    T = acp.shape[0]
    c = T // S
    ts = np.asarray(list(range(0, T, c))) + 1
    a = acp[ts].astype(np.float32)
    a_prev = np.asarray([acp[0]] + acp[ts[:-1]].tolist(), dtype=np.float64)
    return dict(ddim_timesteps=ts, ddim_alphas=a, ddim_alphas_prev=a_prev,
                ddim_sqrt_one_minus_alphas=np.sqrt(np.float32(1.0) - a), ddim_sigmas=np.zeros_like(a_prev))


# Adams-Bashforth combinations of plms.py:144-159: (coefficients of [e_t, old[-1], old[-2], old[-3]], divisor)
PLMS_COEFS = {
    0: ((1.0, 1.0), 2.0),                       # (e_t + e_t_next) / 2   (second evaluation on step 0)
    1: ((3.0, -1.0), 2.0),
    2: ((23.0, -16.0, 5.0), 12.0),
    3: ((55.0, -59.0, 37.0, -9.0), 24.0),
}


def step_coefs(sched: Dict[str, np.ndarray], index: int) -> Tuple[float, float, float, float]:
    """float32 scalars of get_x_prev_and_pred_x0 (plms.py:126-140) with sigma = 0:
    (sqrt(a_t), sqrt(1 - a_t), sqrt(a_prev), sqrt(1 - a_prev))."""
    a_t = np.float32(sched["ddim_alphas"][index])
    a_prev = np.float32(sched["ddim_alphas_prev"][index])
    s1m = np.float32(sched["ddim_sqrt_one_minus_alphas"][index])
    return (float(np.sqrt(a_t)), float(s1m), float(np.sqrt(a_prev)), float(np.sqrt(np.float32(1.0) - a_prev)))
