"""Checkpoint ingestion: reference-named fp32 state_dict -> packed fp16/fp32 device tensors.

The engine never sees torch modules; it consumes the flat dict this module produces.  Packing is
layout work only (no arithmetic beyond dtype casts and tanh of the scalar gates):

  * Linear / 1x1-conv weights  [N, K]            -> fp16 [N, K]
  * 3x3 conv weights [Cout, Cin, 3, 3]          -> fp16 [Cout, 9*Cin], K = (64-channel block, tap, channel) (NHWC implicit GEMM);
                                                   the 4-channel first conv is zero-padded to Cin = 64
  * GEGLU proj [8C, C] (x rows | gate rows)      -> rows interleaved in blocks of 32 (x32 | gate32 | x32 | ...)
                                                   so x_j and gate_j land in adjacent MFMA tiles (bias likewise)
  * self-attention to_q/to_k/to_v                -> one [3C, C] matrix (single QKV GEMM)
  * cross-attention to_k/to_v (context/relations)-> one [2C, 768] matrix (hoisted K/V GEMM)
  * the 22 ResBlock emb_layers                   -> one [sum Cout, 4*mc] matrix (one GEMM per step)
  * biases, norm affine params                   -> fp32
  * alpha_attn / alpha_dense                     -> tanh() as python floats (fuser: multiplied by the
                                                   sampler's scale per step; rela_fuse: constant)
State-dict names follow SURVEY.md App-C, so a real GLIGEN checkpoint's ``saved_ckpt['model']`` loads
(strict=False semantics, interface.py:91: missing keys raise here, unexpected keys are ignored).
"""
from __future__ import annotations

import math
from typing import Dict, Mapping

import numpy as np
import torch

from .arch import Plan, UNetConfig, build_plan, param_shapes

CIN_PAD = 64
LOG2E = 1.4426950408889634


def q_fold(d_head: int) -> float:
    """The softmax scale d^-1/2 (attention.py:136,172) times log2(e), folded into every q projection at pack time: the
    attention kernel then gets logits in exp2 units straight out of the MFMA (gl_attn_args.q_prescaled)."""
    return float(d_head) ** -0.5 * LOG2E


def _t(v, device) -> torch.Tensor:
    if not torch.is_tensor(v):
        v = torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float32)))
    return v.to(device=device, dtype=torch.float32)


def _h(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).contiguous()


def pack_conv3x3(w: torch.Tensor, cin_pad: int | None = None) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> fp16 [Cout, 9*Cin] with K ordered (64-channel block, tap, channel-in-block): the
    implicit-GEMM kernel then visits the 9 taps of one channel block back to back (cache-hot halo reuse)."""
    cout, cin = w.shape[0], w.shape[1]
    if cin_pad is not None and cin_pad > cin:
        z = torch.zeros(cout, cin_pad - cin, 3, 3, dtype=w.dtype, device=w.device)
        w = torch.cat([w, z], dim=1)
        cin = cin_pad
    assert cin % 64 == 0, "conv input channels must be a multiple of 64 (pad the first conv)"
    wp = w.reshape(cout, cin // 64, 64, 3, 3).permute(0, 1, 3, 4, 2)    # [Cout, Cin/64, 3, 3, 64]
    return _h(wp.reshape(cout, -1))


def pack_first_conv(w: torch.Tensor, cin_pad: int) -> torch.Tensor:
    """first conv [Cout, C, 3, 3] fp32 (C = 4 latent channels of 64 padded ones): input channels [Whi | Whi | Wlo | 0] for the latent packed as
    [hi | lo | hi] (gl_pack_latent split): x.W = xhi.Whi + xlo.Whi + xhi.Wlo at no cost.  The first conv alone carries 7 % of the error of storing
    the UNet's weights in fp16 (profiles/r4_weight_rounding_attribution.txt)."""
    w = w.float()
    if 3 * w.shape[1] > cin_pad:
        return pack_conv3x3(w, cin_pad)
    hi = w.half().float()
    lo = (w - hi).half().float()
    return pack_conv3x3(torch.cat([hi, hi, lo], dim=1), cin_pad)


def geglu_interleave(t: torch.Tensor) -> torch.Tensor:
    """rows [x (4C) | gate (4C)] -> blocks of 32: x[0:32], gate[0:32], x[32:64], gate[32:64], ..."""
    half = t.shape[0] // 2
    assert half % 32 == 0, "GEGLU inner dim must be a multiple of 32"
    x = t[:half].reshape(half // 32, 1, 32, *t.shape[1:])
    g = t[half:].reshape(half // 32, 1, 32, *t.shape[1:])
    return torch.cat([x, g], dim=1).reshape(t.shape)


class PackedWeights:
    """``w[name]`` packed tensors + python scalars ``s[name]``.  After ``to_flat()`` every tensor is a VIEW into one
    flat byte buffer ``flat`` whose layout is defined by the C engine's weight table (gl_weight_at, include/gligen_hip.h):
    that buffer is what gl_load_weights consumes and what dist.broadcast_packed sends in one collective."""

    def __init__(self, cfg: UNetConfig, plan: Plan, device):
        self.cfg, self.plan, self.device = cfg, plan, device
        self.w: Dict[str, torch.Tensor] = {}
        self.s: Dict[str, float] = {}
        self.emb_offsets: Dict[str, int] = {}
        self.emb_total = 0
        self.flat: torch.Tensor | None = None
        self.has_sd_conv = False

    def nbytes(self) -> int:
        return int(self.flat.numel()) if self.flat is not None else sum(t.numel() * t.element_size() for t in self.w.values())

    def to_flat(self) -> "PackedWeights":
        """Moves every packed tensor into the flat buffer at the offsets the C engine dictates and rebinds ``w`` to views."""
        from . import _lib
        handle = _lib.create_engine(self.cfg)
        try:
            table, total = _lib.weight_table(handle)
        finally:
            _lib.check(_lib.lib().gl_destroy(handle), "gl_destroy")
        flat = torch.zeros(total, dtype=torch.uint8, device=self.device)
        self.has_sd_conv = "sd_first_conv.w" in self.w
        views: Dict[str, torch.Tensor] = {}
        for name, off, nbytes, dtype, shape in table:
            td = torch.float16 if dtype == 0 else torch.float32
            dst = flat[off:off + nbytes].view(td).view(shape)
            if name in self.w:
                src = self.w[name]
                if tuple(src.shape) != tuple(shape) or src.dtype != td:
                    raise ValueError(f"{name}: packed {tuple(src.shape)} {src.dtype} != engine table {shape} {td}")
                dst.copy_(src)
            elif name in self.s:
                dst.fill_(float(self.s[name]))
                self.s[name] = float(dst.item())       # the gates are fp32 on the device: keep the host copy identical
            elif name.startswith("sd_first_conv."):
                pass                                   # no SD conv given: slot stays zero, gl_load_weights(has_sd_conv=0)
            else:
                raise KeyError(f"engine weight table wants {name}, which the packer did not produce")
            views[name] = dst
        extra = set(self.w) - set(views)
        if extra:
            raise KeyError(f"packer produced tensors the engine does not know: {sorted(extra)[:3]}")
        self.w, self.flat = views, flat
        return self

    @staticmethod
    def from_flat(flat: torch.Tensor, cfg: UNetConfig, device, has_sd_conv: bool) -> "PackedWeights":
        """Rebuilds the views (and the scalar gates) from a flat buffer, e.g. on the receiving ranks of the broadcast."""
        from . import _lib
        P = PackedWeights(cfg, build_plan(cfg), device)
        handle = _lib.create_engine(cfg)
        try:
            table, total = _lib.weight_table(handle)
        finally:
            _lib.check(_lib.lib().gl_destroy(handle), "gl_destroy")
        if flat.numel() != total:
            raise ValueError(f"flat buffer has {flat.numel()} bytes, the engine table needs {total}")
        for name, off, nbytes, dtype, shape in table:
            td = torch.float16 if dtype == 0 else torch.float32
            v = flat[off:off + nbytes].view(td).view(shape)
            if name.endswith((".tanh_attn", ".tanh_dense")):
                P.s[name] = float(v.item())
            P.w[name] = v
        off = 0
        for l in P.plan.all_layers():
            if l.kind == "res":
                P.emb_offsets[l.prefix] = off
                off += l.cout
        P.emb_total = off
        P.flat, P.has_sd_conv = flat, has_sd_conv
        return P


def pack_state_dict(sd: Mapping[str, object], cfg: UNetConfig, device, sd_first_conv: Mapping[str, object] | None = None
                    ) -> PackedWeights:
    plan = build_plan(cfg)
    need = param_shapes(cfg)
    missing = [k for k in need if k not in sd]
    if missing:
        raise KeyError(f"state_dict is missing {len(missing)} tensors, e.g. {missing[:3]}")
    for k, shp in need.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != expected {shp}")
    P = PackedWeights(cfg, plan, device)
    W, S = P.w, P.s
    g = lambda k: _t(sd[k], device)
    split_all = bool(getattr(cfg, "split_weights", False))

    def mat(w32: torch.Tensor) -> torch.Tensor:
        """fp32 [N, K] -> the stored matrix: fp16 [N, K], or rows [Whi | Wlo] (fp16 [N, 2K]) in a split_weights table
        (gl_unet_config.split_weights: strict mode's third pass x.Wlo restores the fp32 weight to ~22 bits)"""
        w32 = w32.reshape(w32.shape[0], -1).float()
        hi = _h(w32)
        if not split_all:
            return hi
        return torch.cat([hi, _h(w32 - hi.float())], 1).contiguous()

    def lin(p, dst=None, bias=True):
        dst = dst or p
        W[dst + ".w"] = mat(g(p + ".weight"))
        if bias:
            W[dst + ".b"] = g(p + ".bias").contiguous()

    def lin_split(p):
        """the three kinds of 1x1 conv: rows [Whi | Wlo], Whi = fp16(W), Wlo = fp16(W - Whi) (engine.hip add_lin_split; they carry 57 % of the
        error of storing the weights in fp16, profiles/r4_weight_rounding_attribution.txt)"""
        w32 = g(p + ".weight").reshape(g(p + ".weight").shape[0], -1).float()
        hi = _h(w32)
        lo = _h(w32 - hi.float())
        W[p + ".w"] = torch.cat([hi, lo], 1).contiguous()
        W[p + ".b"] = g(p + ".bias").contiguous()

    def norm(p):
        W[p + ".g"] = g(p + ".weight").contiguous()
        W[p + ".b"] = g(p + ".bias").contiguous()

    def conv3(p, cin_pad=None):
        w32 = g(p + ".weight").float()
        hi = pack_conv3x3(w32, cin_pad)
        if split_all:        # rows [Whi | Wlo], each half in the (64-channel block, tap, channel) order (gl_conv_args.w_split)
            lo = pack_conv3x3(w32 - w32.half().float(), cin_pad)
            hi = torch.cat([hi, lo], 1).contiguous()
        W[p + ".w"] = hi
        W[p + ".b"] = g(p + ".bias").contiguous()

    def ff(p):
        W[p + ".ff1.w"] = mat(geglu_interleave(g(p + ".net.0.proj.weight")))
        W[p + ".ff1.b"] = geglu_interleave(g(p + ".net.0.proj.bias")).contiguous()
        lin(p + ".net.2", p + ".ff2")

    def self_attn(p, d):
        W[p + ".qkv.w"] = mat(torch.cat([g(p + ".to_q.weight") * q_fold(d), g(p + ".to_k.weight"), g(p + ".to_v.weight")], 0))
        lin(p + ".to_out.0", p + ".o")

    def cross_attn(p, d):
        W[p + ".q.w"] = mat(g(p + ".to_q.weight") * q_fold(d))
        W[p + ".kv.w"] = mat(torch.cat([g(p + ".to_k.weight"), g(p + ".to_v.weight")], 0))
        lin(p + ".to_out.0", p + ".o")

    lin("time_embed.0")
    lin("time_embed.2")
    # first conv: GLIGEN's own and the "SD" replacement (openaimodel.py:393-405)
    W["input_blocks.0.0.w"] = pack_first_conv(g("input_blocks.0.0.weight"), CIN_PAD)
    W["input_blocks.0.0.b"] = g("input_blocks.0.0.bias").contiguous()
    if sd_first_conv is not None:
        W["sd_first_conv.w"] = pack_first_conv(_t(sd_first_conv["weight"], device), CIN_PAD)
        W["sd_first_conv.b"] = _t(sd_first_conv["bias"], device).contiguous()

    emb_w, emb_b, off = [], [], 0
    for l in plan.all_layers():
        p = l.prefix
        if l.kind in ("down", "up"):
            conv3(p)
        elif l.kind == "res":
            norm(p + ".in_layers.0")
            conv3(p + ".in_layers.2")
            norm(p + ".out_layers.0")
            conv3(p + ".out_layers.3")
            if l.cin != l.cout:
                lin_split(p + ".skip_connection")
            emb_w.append(g(p + ".emb_layers.1.weight"))
            emb_b.append(g(p + ".emb_layers.1.bias"))
            P.emb_offsets[p] = off
            off += l.cout
        elif l.kind == "st":
            norm(p + ".norm")
            lin_split(p + ".proj_in")
            lin_split(p + ".proj_out")
            t = p + ".transformer_blocks.0"
            self_attn(t + ".attn1", l.d_head)
            cross_attn(t + ".attn2", l.d_head)
            ff(t + ".ff")
            for n in ("norm1", "norm2", "norm3"):
                norm(f"{t}.{n}")
            f = t + ".fuser"
            lin(f + ".linear")
            self_attn(f + ".attn", l.d_head)
            ff(f + ".ff")
            norm(f + ".norm1")
            norm(f + ".norm2")
            S[f + ".tanh_attn"] = math.tanh(float(g(f + ".alpha_attn")))
            S[f + ".tanh_dense"] = math.tanh(float(g(f + ".alpha_dense")))
            r = t + ".rela_fuse"
            cross_attn(r + ".attn", l.d_head)
            ff(r + ".ff")
            for n in ("norm1", "norm2", "norm3"):
                norm(f"{r}.{n}")
            S[r + ".tanh_attn"] = math.tanh(float(g(r + ".alpha_attn")))
            S[r + ".tanh_dense"] = math.tanh(float(g(r + ".alpha_dense")))
    W["emb_all.w"] = mat(torch.cat(emb_w, 0))
    W["emb_all.b"] = torch.cat(emb_b, 0).contiguous()
    P.emb_total = off
    norm("out.0")
    conv3("out.2")
    W["position_net.null_pos"] = g("position_net.null_positive_feature").contiguous()
    W["position_net.null_xyxy"] = g("position_net.null_position_feature").contiguous()
    for i in (0, 2, 4):
        lin(f"position_net.linears.{i}")
    return P.to_flat()


@torch.no_grad()
def random_state_dict(cfg: UNetConfig, device, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Fast on-device random weights with the recipe's *scaling rules* (not its values): used by the
    benchmark, where only shapes and magnitudes matter (no checkpoint is obtainable offline)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        u = lambda: torch.rand(shape if len(shape) else (1,), generator=gen, device=device, dtype=torch.float32) * 2 - 1
        if leaf in ("alpha_attn", "alpha_dense"):
            v = u().reshape(())
            t = torch.sign(v) * (0.3 + 0.35 * v.abs())
        elif leaf.startswith("null_"):
            t = u() * 0.5
        elif (".norm" in name or "in_layers.0" in name or "out_layers.0" in name or name.startswith("out.0")) and len(shape) == 1:
            t = 1.0 + 0.1 * u() if leaf == "weight" else 0.05 * u()
        elif leaf == "bias":
            t = 0.02 * u()
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = u() * math.sqrt(3.0 / max(fan_in, 1))
        out[name] = t
    return out


@torch.no_grad()
def random_vae_state_dict(cfg, device, seed: int = 1) -> Dict[str, torch.Tensor]:
    """On-device random VAE-decoder weights with the recipe's scaling rules (benchmark only)."""
    from .arch import vae_decoder_param_shapes
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in vae_decoder_param_shapes(cfg).items():
        u = torch.rand(shape, generator=gen, device=device, dtype=torch.float32) * 2 - 1
        leaf = name.rsplit(".", 1)[-1]
        if "norm" in name and len(shape) == 1:
            out[name] = 1.0 + 0.1 * u if leaf == "weight" else 0.05 * u
        elif leaf == "bias":
            out[name] = 0.02 * u
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            out[name] = u * math.sqrt(3.0 / max(fan_in, 1))
    return out
