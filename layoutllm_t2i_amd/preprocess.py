"""CLIP image preprocessing of the reward stage on the GPU.

The reference's ``Reward_Model.forward`` (models/policy.py:108-111) hands PIL images to the HuggingFace CLIP feature extractor
(``self.processor(images=imgs_pred, return_tensors="pt")``; transformers 4.19.2 pinned in env_docker/Dockerfile:3), which

1. resizes the SHORTEST edge to 224 with ``PIL.Image.resize(..., resample=BICUBIC)`` (the long edge becomes
   ``int(224 * long / short)``), 2. centre-crops 224 x 224, 3. ``float32(u8) / 255``, 4. ``(x - mean) / std``, channels first.

``ClipImagePreprocessor`` does the same on uint8 images that are already in HBM (``gl_resample_h_u8`` / ``gl_resample_v_norm``,
csrc/preprocess.hip) -- Pillow's two-pass fixed-point resampling bit for bit -- and ``from_decoded`` starts from the VAE
decoder's fp32 output with the uint8 conversion of GLIGEN/interface.py:543-547, so a rollout's images go decoder -> uint8 ->
224 x 224 ``pixel_values`` -> CLIP vision tower without leaving the GPU.  Only the coefficient tables (a few KB per distinct
image size, Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` in float64) are computed on the host, once per size.
No fallback: a missing GPU / library raises.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from . import ops

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    """Pillow's bicubic_filter (a = -0.5), same operation order."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's precompute_coeffs(in_size, 0, in_size, out_size, BICUBIC) + normalize_coeffs_8bpc: bounds int32 [out, 2]
    (first input sample, tap count) and 22-bit fixed-point coefficients int32 [out, ksize].  in_size == out_size gives the
    identity table (Pillow skips that pass; one tap of 2^22 reproduces the input exactly)."""
    if in_size <= 0 or out_size <= 0:
        raise ValueError("sizes must be positive")
    if in_size == out_size:
        bounds = np.stack([np.arange(out_size, dtype=np.int32), np.ones(out_size, dtype=np.int32)], 1)
        return np.ascontiguousarray(bounds), np.full((out_size, 1), 1 << PRECISION_BITS, dtype=np.int32)
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # Pillow: box edges are C floats
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def resized_shape(h: int, w: int, size: int) -> Tuple[int, int]:
    """transformers 4.19.2 ImageFeatureExtractionMixin.resize with an int size, default_to_square=False."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


class ClipImagePreprocessor:
    def __init__(self, size: int = 224, crop_size: int = 224, image_mean: Sequence[float] = CLIP_MEAN, image_std: Sequence[float] = CLIP_STD,
                 device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("ClipImagePreprocessor needs a GPU: the reward stage has no CPU fallback")
        self.size, self.crop, self.device = int(size), int(crop_size), torch.device(device)
        self.mean = np.asarray(image_mean, dtype=np.float32)      # np.array(mean).astype(image.dtype) in the reference's normalize
        self.std = np.asarray(image_std, dtype=np.float32)
        self._tables: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, int]] = {}

    def _table(self, n_in: int, n_out: int):
        key = (n_in, n_out)
        t = self._tables.get(key)
        if t is None:
            b, k = pil_bicubic_tables(n_in, n_out)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), int(k.shape[1]))
            self._tables[key] = t
        return t

    @torch.no_grad()
    def __call__(self, images_u8: torch.Tensor, return_u8: bool = False):
        """images_u8: uint8 [B, H, W, 3] on the GPU (one size per call) -> pixel_values fp32 [B, 3, crop, crop]."""
        x = images_u8
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[-1] != 3 or not x.is_cuda:
            raise ValueError("images_u8 must be a uint8 [B, H, W, 3] tensor on the GPU")
        x = x.contiguous()
        B, H, W, _ = x.shape
        Ho, Wo = resized_shape(H, W, self.size)
        if Ho < self.crop or Wo < self.crop:
            raise NotImplementedError("centre crop larger than the resized image (the reference pads): not built")
        top, left = (Ho - self.crop) // 2, (Wo - self.crop) // 2
        with torch.cuda.device(self.device):
            hb, hk, hks = self._table(W, Wo)
            vb, vk, vks = self._table(H, Ho)
            tmp = torch.empty(B, H, Wo, 3, dtype=torch.uint8, device=self.device)
            ops.resample_h_u8(x, hb, hk, hks, tmp)
            out = torch.empty(B, 3, self.crop, self.crop, dtype=torch.float32, device=self.device)
            u8 = torch.empty(B, self.crop, self.crop, 3, dtype=torch.uint8, device=self.device) if return_u8 else None
            ops.resample_v_norm(tmp, vb, vk, vks, Ho, top, left, self.crop, self.crop, self.mean, self.std, out, u8)
        return (out, u8) if return_u8 else out

    @torch.no_grad()
    def to_u8(self, decoded: torch.Tensor) -> torch.Tensor:
        """VAE decoder output fp32 [B, 3, H, W] -> uint8 [B, H, W, 3]: the pixels of interface.py:543-547's PIL image."""
        d = decoded.to(self.device, torch.float32).contiguous()
        B, _, H, W = d.shape
        with torch.cuda.device(self.device):
            return ops.image_to_u8(d, torch.empty(B, H, W, 3, dtype=torch.uint8, device=self.device))

    def from_decoded(self, decoded: torch.Tensor) -> torch.Tensor:
        return self(self.to_u8(decoded))

    def from_pil(self, images) -> torch.Tensor:
        """list of PIL images / uint8 HWC arrays of ANY sizes (ground-truth images): grouped by size, order preserved."""
        arrs = [np.asarray(im.convert("RGB") if hasattr(im, "convert") else im, dtype=np.uint8) for im in images]
        for a in arrs:
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError(f"images must be RGB (PIL images are converted; arrays must be uint8 [H, W, 3]), got {a.shape}")
        out = torch.empty(len(arrs), 3, self.crop, self.crop, dtype=torch.float32, device=self.device)
        groups: Dict[Tuple[int, int], list] = {}
        for i, a in enumerate(arrs):
            groups.setdefault(a.shape[:2], []).append(i)
        for idx in groups.values():
            batch = torch.from_numpy(np.stack([arrs[i] for i in idx])).to(self.device)
            out[torch.as_tensor(idx, device=self.device)] = self(batch)
        return out
