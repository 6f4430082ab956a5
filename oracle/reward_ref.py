"""ORACLE (test infrastructure, not product): fp32 CPU restatement of the GPU part of the rollout's reward scoring.

Restates models/policy.py:114-124,137 (``Reward.forward``, the class SURVEY calls Reward_Model: CLIP similarities of
F.normalize'd features, aesthetic score of the re-normalised image embedding, the weighted sum) and tools/aesthetic.py:9-31
(AestheticMLP = five Linear layers with Dropouts, no activations), tools/aesthetic.py:52-57 (``normalized``).

PINNED by tests/golden/reward_ref.npz: tools/make_reward_goldens.py takes ``AestheticMLP`` / ``normalized`` and the
statements of ``Reward.forward`` named above out of the reference FILES with ``ast`` (the modules themselves cannot be imported
in this image: pytorch_lightning and others are absent), executes them on recipe weights / features and stores the outputs;
tests/test_reward.py::test_oracle_matches_reference_generated_golden holds this restatement to them (fp32, <= 2e-6).
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
