"""Reward scoring stage: oracle vs an independently built nn.Sequential (CPU), HIP kernel vs oracle (GPU)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from layoutllm_t2i_amd import recipe
from oracle import reward_ref


def mlp_sd(seed=0, D=768):
    shapes = {0: (1024, D), 2: (128, 1024), 4: (64, 128), 6: (16, 64), 7: (1, 16)}
    sd = {}
    for li, (n, k) in shapes.items():
        sd[f"layers.{li}.weight"] = torch.from_numpy(recipe.uniform(f"aes.{li}.w", (n, k), seed)) * float(np.sqrt(3.0 / k))
        sd[f"layers.{li}.bias"] = torch.from_numpy(recipe.uniform(f"aes.{li}.b", (n,), seed)) * 0.1
    return sd


def feats(B, seed, D=768):
    g = lambda n: torch.from_numpy(recipe.normal(f"rw.{n}", (B, D), seed)) * 3.0
    return g("txt"), g("pred") + 0.5 * g("txt"), g("gt") + 0.3 * g("pred")


def test_oracle_equals_torch_sequential():
    """the layer list of tools/aesthetic.py:21-34 as a plain nn.Sequential (eval mode: Dropout = identity)"""
    import torch.nn as nn
    sd = mlp_sd()
    net = nn.Sequential(nn.Linear(768, 1024), nn.Dropout(0.2), nn.Linear(1024, 128), nn.Dropout(0.2), nn.Linear(128, 64), nn.Dropout(0.1),
                        nn.Linear(64, 16), nn.Linear(16, 1)).eval()
    net.load_state_dict({k.replace("layers.", ""): v for k, v in sd.items()})
    t, p, g = feats(5, 1)
    with torch.no_grad():
        pn = torch.nn.functional.normalize(p, dim=-1)
        want = net(pn / pn.norm(dim=-1, keepdim=True)).flatten()
    got = reward_ref.reward_scores(sd, t, p, g)
    torch.testing.assert_close(got["aes_reward"], want, rtol=1e-5, atol=1e-6)
    cos = torch.nn.functional.cosine_similarity
    torch.testing.assert_close(got["clip_reward"], cos(t, p) + cos(g, p), rtol=1e-5, atol=1e-6)
    z = torch.zeros(2, 768)
    out = reward_ref.reward_scores(sd, z, z, z)              # zero features: F.normalize eps / normalized()'s zero-norm rule
    assert torch.isfinite(out["reward"]).all() and float(out["clip_reward"].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 16, 64])
def test_hip_reward_score_vs_oracle(B):
    from layoutllm_t2i_amd.reward import RewardScorer
    sd = mlp_sd(3)
    t, p, g = feats(B, 7)
    if B > 1:
        p[1] = 0.0                                           # a zero embedding: norm clamps, no NaN
    miou = torch.rand(B)
    laysim = torch.rand(B)
    sc = RewardScorer(sd, "cuda:0")
    got = sc.score(t, p, g, miou, laysim)
    want = reward_ref.reward_scores(sd, t, p, g, miou, laysim)
    for k in ("sims_ti", "sims_ii", "clip_reward", "aes_reward", "reward"):
        torch.testing.assert_close(got[k].cpu(), want[k], rtol=2e-5, atol=2e-5, msg=k)
    again = sc.score(t, p, g, miou, laysim)
    assert all(torch.equal(got[k], again[k]) for k in got), "deterministic (fixed summation order)"
