"""ORACLE (test infrastructure, not product): fp32 CPU restatement of the GPU part of the rollout's reward scoring.

Restates models/policy.py:106-124,135 (``Reward_Model.forward``: CLIP similarities of F.normalize'd features, aesthetic
score of the re-normalised image embedding, the weighted sum) and tools/aesthetic.py:15-31 (AestheticMLP = five Linear
layers with Dropouts, no activations), tools/aesthetic.py:52-57 (``normalized``).

PARITY UNPINNED: tools/aesthetic.py imports pytorch_lightning and models/policy.py imports modules that are absent
from this image, so the reference classes cannot be imported to generate goldens; tests/test_reward.py checks this
restatement against an independently built ``torch.nn.Sequential`` of the same layer list instead.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

LINEARS = (0, 2, 4, 6, 7)


def aesthetic_mlp(sd, x: torch.Tensor) -> torch.Tensor:
    for li in LINEARS:
        x = F.linear(x, sd[f"layers.{li}.weight"], sd[f"layers.{li}.bias"])
    return x


def normalized(a: np.ndarray, axis=-1, order=2) -> np.ndarray:
    l2 = np.atleast_1d(np.linalg.norm(a, order, axis))
    l2[l2 == 0] = 1
    return a / np.expand_dims(l2, axis)


def reward_scores(sd, txt, img_pred, img_gt, miou=None, laysim=None):
    t, p, g = (F.normalize(v.float(), dim=-1) for v in (txt, img_pred, img_gt))
    sims_ti = (t * p).sum(dim=-1)
    sims_ii = (g * p).sum(dim=-1)
    clip_reward = sims_ti + sims_ii
    emb = torch.from_numpy(normalized(p.numpy())).float()
    aes = aesthetic_mlp(sd, emb).flatten()
    reward = clip_reward + aes * 0.1
    if miou is not None:
        reward = reward + miou * 10
    if laysim is not None:
        reward = reward + laysim * 10
    return dict(sims_ti=sims_ti, sims_ii=sims_ii, clip_reward=clip_reward, aes_reward=aes, reward=reward)
