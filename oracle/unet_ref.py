"""ORACLE (test infrastructure, not product): fp32 CPU restatement of the layout-conditioned UNet.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; it is the *checker*, never the thing shipped or measured.

It restates, as plain functional PyTorch over a ``{name: tensor}`` state_dict with the reference's
parameter names, what these reference functions compute:

  UNetModel.forward            GLIGEN/ldm/modules/diffusionmodules/openaimodel.py:413-459
  ResBlock._forward            openaimodel.py:211-231
  Upsample / Downsample        openaimodel.py:75-85, 112-114
  SpatialTransformer.forward   GLIGEN/ldm/modules/attention.py:436-446
  BasicTransformerBlock        attention.py:394-402
  SelfAttention / CrossAttn    attention.py:159-178 / 122-143
  GatedSelfAttentionDense      attention.py:226-234
  RelationCrossAttention       attention.py:315-359
  FeedForward / GEGLU          attention.py:38-65
  PositionNet / Fourier        text_grounding_net.py:26-43 / diffusionmodules/util.py:12-26
  timestep_embedding           diffusionmodules/util.py:161-181

Parity pin: ``tests/test_oracle_golden.py`` checks every function here against golden vectors
produced by importing the reference itself in the build container
(``tools/make_goldens.py`` -> ``tests/golden/*.npz``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- small pieces

def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """[cos | sin] sinusoid (util.py:161-181; cos first, util.py:176)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def fourier_embed(x: torch.Tensor, num_freqs: int = 8, temperature: float = 100.0) -> torch.Tensor:
    """util.py:12-26: per frequency a sin block then a cos block, concatenated on the last dim."""
    bands = temperature ** (torch.arange(num_freqs) / num_freqs)
    parts = []
    for f in bands:
        parts.append(torch.sin(f * x))
        parts.append(torch.cos(f * x))
    return torch.cat(parts, dim=-1)


def _lin(sd: SD, p: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def _layer_norm(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _group_norm(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def position_net(sd: SD, boxes, masks, positive_embeddings, num_freqs: int = 8) -> torch.Tensor:
    """text_grounding_net.py:26-43."""
    p = "position_net"
    m = masks.unsqueeze(-1)
    xyxy = fourier_embed(boxes, num_freqs)
    pos_null = sd[p + ".null_positive_feature"].view(1, 1, -1)
    xyxy_null = sd[p + ".null_position_feature"].view(1, 1, -1)
    pe = positive_embeddings * m + (1 - m) * pos_null
    xy = xyxy * m + (1 - m) * xyxy_null
    h = torch.cat([pe, xy], dim=-1)
    h = F.silu(_lin(sd, p + ".linears.0", h))
    h = F.silu(_lin(sd, p + ".linears.2", h))
    return _lin(sd, p + ".linears.4", h)


def attention(sd: SD, p: str, x, key, value, heads: int) -> torch.Tensor:
    """Dense multi-head attention, no q/k/v bias, scale after QK^T, softmax over keys, to_out with
    bias (attention.py:122-143 with mask=None and :159-178)."""
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(key, sd[p + ".to_k.weight"])
    v = F.linear(value, sd[p + ".to_v.weight"])
    B, N, HC = q.shape
    M = k.shape[1]
    d = HC // heads
    q = q.view(B, N, heads, d).transpose(1, 2)
    k = k.view(B, M, heads, d).transpose(1, 2)
    v = v.view(B, M, heads, d).transpose(1, 2)
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.matmul(attn, v).transpose(1, 2).reshape(B, N, HC)
    return _lin(sd, p + ".to_out.0", out)


def feed_forward(sd: SD, p: str, x) -> torch.Tensor:
    """GEGLU feed-forward (attention.py:38-65); exact-erf GELU on the gate half."""
    h = _lin(sd, p + ".net.0.proj", x)
    a, g = h.chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(g))


def gated_self_attention(sd: SD, p: str, x, objs, heads: int, scale: float) -> torch.Tensor:
    """attention.py:226-234."""
    n_vis = x.shape[1]
    o = _lin(sd, p + ".linear", objs)
    cat = torch.cat([x, o], dim=1)
    a = attention(sd, p + ".attn", *(3 * [_layer_norm(sd, p + ".norm1", cat)]), heads)[:, :n_vis]
    x = x + scale * torch.tanh(sd[p + ".alpha_attn"]) * a
    x = x + scale * torch.tanh(sd[p + ".alpha_dense"]) * feed_forward(sd, p + ".ff", _layer_norm(sd, p + ".norm2", x))
    return x


def box_rects(boxes: torch.Tensor, masks: torch.Tensor, h: int, w: int):
    """Integer pixel rectangles exactly as attention.py:321-330 derives them (fp32 multiply, trunc
    toward zero, x1/y1 clamped to w/h, x0/y0 not).  Returns python lists."""
    n_valid = torch.sum(masks, dim=-1).tolist()
    width = torch.full(boxes.shape[:2], w).to(boxes)
    height = torch.full(boxes.shape[:2], h).to(boxes)
    x0 = (boxes[:, :, 0] * w).to(torch.int).tolist()
    y0 = (boxes[:, :, 1] * h).to(torch.int).tolist()
    x1 = torch.minimum(boxes[:, :, 2] * w, width).to(torch.int).tolist()
    y1 = torch.minimum(boxes[:, :, 3] * h, height).to(torch.int).tolist()
    return n_valid, x0, y0, x1, y1


def relation_cross_attention(sd: SD, p: str, x, relations, boxes, masks, h: int, w: int, heads: int) -> torch.Tensor:
    """attention.py:315-359.  Follows the reference step by step (including the ``break`` at the
    first padded or degenerate box and python slice semantics) but accumulates the masked
    broadcast-add instead of materialising three B x 30 x h x w x C tensors."""
    b, _, c = x.shape
    hid = _layer_norm(sd, p + ".norm3", x).view(b, h, w, c)
    mo = boxes.size(1)
    n_valid, x0, y0, x1, y1 = box_rects(boxes, masks, h, w)
    feats = torch.zeros((b, mo, c), dtype=x.dtype)
    used = [[] for _ in range(b)]
    for k in range(b):
        for i in range(mo):
            left, right, top, bottom = x0[k][i], x1[k][i], y0[k][i], y1[k][i]
            if i < n_valid[k] and left != right and top != bottom:
                region = hid[k, top:bottom, left:right, :].reshape(-1, c)
                feats[k, i] = torch.mean(region, dim=0)     # NaN for an empty slice, like the reference
                used[k].append((i, top, bottom, left, right))
            else:
                break
    feats = feats + torch.tanh(sd[p + ".alpha_attn"]) * attention(
        sd, p + ".attn", _layer_norm(sd, p + ".norm1", feats), relations, relations, heads)
    feats = feats + torch.tanh(sd[p + ".alpha_dense"]) * feed_forward(sd, p + ".ff", _layer_norm(sd, p + ".norm2", feats))
    # hidden.repeat(mo) + obj_mask * obj_features, then mean over mo (attention.py:354-358)
    acc = hid * mo
    for k in range(b):
        for (i, top, bottom, left, right) in used[k]:
            acc[k, top:bottom, left:right, :] += feats[k, i]
        if any(torch.isnan(feats[k, i]).any() for (i, *_r) in used[k]):
            acc[k] = float("nan")      # reference: 0 * NaN = NaN on every pixel of this sample
    return (acc / mo).view(b, h * w, c)


def transformer_block(sd: SD, p: str, x, context, objs, relations, boxes, masks, h, w, heads, fuser_scale):
    """attention.py:394-402."""
    n1 = _layer_norm(sd, p + ".norm1", x)
    x = attention(sd, p + ".attn1", n1, n1, n1, heads) + x
    x = gated_self_attention(sd, p + ".fuser", x, objs, heads, fuser_scale)
    x = (relation_cross_attention(sd, p + ".rela_fuse", x, relations, boxes, masks, h, w, heads) + x) / 2
    x = attention(sd, p + ".attn2", _layer_norm(sd, p + ".norm2", x), context, context, heads) + x
    x = feed_forward(sd, p + ".ff", _layer_norm(sd, p + ".norm3", x)) + x
    return x


def spatial_transformer(sd: SD, p: str, x, context, objs, relations, boxes, masks, heads, fuser_scale):
    """attention.py:436-446 (GroupNorm eps 1e-6, attention.py:79)."""
    b, c, h, w = x.shape
    x_in = x
    y = _group_norm(sd, p + ".norm", x, 1e-6)
    y = F.conv2d(y, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    y = y.flatten(2).transpose(1, 2)
    y = transformer_block(sd, p + ".transformer_blocks.0", y, context, objs, relations, boxes, masks, h, w,
                          heads, fuser_scale)
    y = y.transpose(1, 2).reshape(b, c, h, w)
    y = F.conv2d(y, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return y + x_in


def res_block(sd: SD, p: str, x, emb) -> torch.Tensor:
    """openaimodel.py:211-231 (GroupNorm32 eps 1e-5, no scale-shift norm, dropout p=0)."""
    h = F.silu(_group_norm(sd, p + ".in_layers.0", x, 1e-5))
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + e[:, :, None, None]
    h = F.silu(_group_norm(sd, p + ".out_layers.0", h, 1e-5))
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


# ----------------------------------------------------------------------------- whole model

def unet_forward(sd: SD, cfg, x, timesteps, context, relations, boxes, masks, positive_embeddings,
                 fuser_scale: float = 1.0, first_conv: Optional[dict] = None) -> torch.Tensor:
    """UNetModel.forward (openaimodel.py:413-459) for the text_layout grounding tokenizer.

    ``cfg`` needs: model_channels, channel_mult, num_res_blocks, attention_resolutions, num_heads,
    fourier_freqs.  Null grounding = all-zero boxes/masks/positive_embeddings
    (text_layout_tokinzer_input.py:47-62).  ``first_conv`` overrides input_blocks.0.0
    (restore_first_conv_from_SD, openaimodel.py:393-405).
    """
    mc, heads = cfg.model_channels, cfg.num_heads
    objs = position_net(sd, boxes, masks, positive_embeddings, cfg.fourier_freqs)
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", timestep_embedding(timesteps, mc))))

    def st(p, h):
        return spatial_transformer(sd, p, h, context, objs, relations, boxes, masks, heads, fuser_scale)

    w0 = first_conv["weight"] if first_conv is not None else sd["input_blocks.0.0.weight"]
    b0 = first_conv["bias"] if first_conv is not None else sd["input_blocks.0.0.bias"]
    h = F.conv2d(x, w0, b0, padding=1)
    skips = [h]
    idx, ds = 1, 1
    n_levels = len(cfg.channel_mult)
    for level in range(n_levels):
        for _ in range(cfg.num_res_blocks):
            h = res_block(sd, f"input_blocks.{idx}.0", h, emb)
            if ds in cfg.attention_resolutions:
                h = st(f"input_blocks.{idx}.1", h)
            skips.append(h)
            idx += 1
        if level != n_levels - 1:
            h = F.conv2d(h, sd[f"input_blocks.{idx}.0.op.weight"], sd[f"input_blocks.{idx}.0.op.bias"],
                         stride=2, padding=1)
            skips.append(h)
            idx += 1
            ds *= 2
    h = res_block(sd, "middle_block.0", h, emb)
    h = st("middle_block.1", h)
    h = res_block(sd, "middle_block.2", h, emb)
    idx = 0
    for level in reversed(range(n_levels)):
        for i in range(cfg.num_res_blocks + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = res_block(sd, f"output_blocks.{idx}.0", h, emb)
            j = 1
            if ds in cfg.attention_resolutions:
                h = st(f"output_blocks.{idx}.{j}", h)
                j += 1
            if level and i == cfg.num_res_blocks:
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = F.conv2d(h, sd[f"output_blocks.{idx}.{j}.conv.weight"], sd[f"output_blocks.{idx}.{j}.conv.bias"],
                             padding=1)
                ds //= 2
            idx += 1
    h = F.silu(_group_norm(sd, "out.0", h, 1e-5))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)
