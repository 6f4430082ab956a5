"""ORACLE (test infrastructure, not product): fp32 CPU restatement of the VAE decode stage.

Restates AutoencoderKL.decode (GLIGEN/ldm/models/autoencoder.py:40-44) and Decoder.forward
(GLIGEN/ldm/modules/diffusionmodules/model.py:535-568) with its ResnetBlock (:82-141, temb = None),
single-head AttnBlock (:150-202) and Upsample (:42-56), over a {name: tensor} state_dict that uses the
reference's names (post_quant_conv.*, decoder.*).  GroupNorm eps 1e-6 (model.py:38-39), swish = x*sigmoid(x).

Pinned by tests/test_oracle_golden.py against tests/golden/vae_tiny.npz (reference-generated).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _gn(sd: SD, p: str, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def _conv(sd: SD, p: str, x, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def resnet_block(sd: SD, p: str, x):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x)), 1)
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h)), 1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def attn_block(sd: SD, p: str, x):
    h = _gn(sd, p + ".norm", x)
    q, k, v = _conv(sd, p + ".q", h), _conv(sd, p + ".k", h), _conv(sd, p + ".v", h)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h)


def decode(sd: SD, z: torch.Tensor, ch_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks: int = 2,
           scale_factor: float = 0.18215) -> torch.Tensor:
    """z [B, 4, h, w] -> image [B, 3, 8h.., 8w..] in [-1, 1] nominal range (not clamped)."""
    z = 1.0 / scale_factor * z
    z = _conv(sd, "post_quant_conv", z)
    h = _conv(sd, "decoder.conv_in", z, 1)
    h = resnet_block(sd, "decoder.mid.block_1", h)
    h = attn_block(sd, "decoder.mid.attn_1", h)
    h = resnet_block(sd, "decoder.mid.block_2", h)
    for lvl in reversed(range(len(ch_mult))):
        for i in range(num_res_blocks + 1):
            h = resnet_block(sd, f"decoder.up.{lvl}.block.{i}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up.{lvl}.upsample.conv", h, 1)
    h = F.silu(_gn(sd, "decoder.norm_out", h))
    return _conv(sd, "decoder.conv_out", h, 1)
