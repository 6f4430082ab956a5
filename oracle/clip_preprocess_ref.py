"""ORACLE (test infrastructure, not product): CPU restatement of the CLIP image preprocessing the reward model runs.

The reference's ``Reward_Model.forward`` (models/policy.py:108-111) calls ``self.processor(images=imgs, return_tensors="pt")``,
the HuggingFace CLIP feature extractor -- third-party code that is NOT part of /root/reference (``transformers==4.19.2``,
env_docker/Dockerfile:3; it resizes through Pillow).  Restated here from their published sources:

* transformers 4.19.2 ``CLIPFeatureExtractor.__call__`` / ``ImageFeatureExtractionMixin``: resize the shortest edge to 224
  (``int(size * long / short)`` for the other one) with ``PIL.Image.resize(resample=BICUBIC)``, centre crop 224 x 224
  (``top = (h - 224) // 2``), ``image.astype(np.float32) / 255.0``, ``(image - mean) / std`` with fp32 mean / std, channels first;
* Pillow ``src/libImaging/Resample.c`` (``precompute_coeffs``, ``normalize_coeffs_8bpc``, ``ImagingResampleHorizontal_8bpc``,
  ``ImagingResampleVertical_8bpc``): separable two-pass resampling on 8-bit data, horizontal pass first, 22-bit fixed-point
  coefficients, the intermediate image rounded to 8 bits;
* the uint8 conversion before it is the reference's own: GLIGEN/interface.py:543-547 (clamp, * 0.5 + 0.5 in torch fp32,
  ``.numpy() * 255``, ``astype(np.uint8)``).

Pinned by tests/test_preprocess.py against Pillow itself (``Image.resize``, bit-exact on uint8) and against the installed
transformers' PIL-backend CLIP image processor (bit-exact on the fp32 pixel_values) -- both importable wherever the tests run.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Resample.c precompute_coeffs(inSize, 0, inSize, outSize, &BICUBIC): bounds [out, 2], float64 coefficients [out, ksize]."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        for x in range(xmax):
            w = bicubic_filter((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def normalize_coeffs_8bpc(kk: np.ndarray) -> np.ndarray:
    out = np.zeros(kk.shape, dtype=np.int64)
    for idx, v in np.ndenumerate(kk):
        out[idx] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
    return out


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    bounds, kk, _ = precompute_coeffs(img.shape[0], out_size)
    ik = normalize_coeffs_8bpc(kk)
    x = img.astype(np.int64)
    out = np.zeros((out_size,) + img.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        ss = (1 << (PRECISION_BITS - 1)) + (x[xmin:xmin + n] * ik[xx, :n].reshape((n,) + (1,) * (img.ndim - 1))).sum(0)
        out[xx] = np.clip(ss >> PRECISION_BITS, 0, 255).astype(np.uint8)          # clip8: arithmetic shift, then the lookup's clamp
    return out


def pil_resize_bicubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """ImagingResample on an RGB uint8 [H, W, 3] image: horizontal pass, then vertical, each skipped when the size is unchanged."""
    h, w, _ = img.shape
    t = img if out_w == w else _resample_axis0(img.transpose(1, 0, 2), out_w).transpose(1, 0, 2)
    return t if out_h == h else _resample_axis0(t, out_h)


def resized_shape(h: int, w: int, size: int) -> Tuple[int, int]:
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def clip_feature_extractor(img: np.ndarray, size: int = 224, crop: int = 224) -> Tuple[np.ndarray, np.ndarray]:
    """one RGB uint8 [H, W, 3] image -> (pixel_values fp32 [3, crop, crop], the cropped uint8 pixels [crop, crop, 3])"""
    h, w, _ = img.shape
    oh, ow = resized_shape(h, w, size)
    r = pil_resize_bicubic(img, oh, ow)
    top, left = (oh - crop) // 2, (ow - crop) // 2
    c = r[top:top + crop, left:left + crop]
    x = c.astype(np.float32) / 255.0
    mean = np.array(CLIP_MEAN).astype(np.float32)
    std = np.array(CLIP_STD).astype(np.float32)
    return ((x.transpose(2, 0, 1) - mean[:, None, None]) / std[:, None, None]).astype(np.float32), c


def decoded_to_u8(img) -> np.ndarray:
    """GLIGEN/interface.py:543-547 on a batch: torch fp32 [B, 3, H, W] -> uint8 [B, H, W, 3]"""
    import torch
    s = torch.clamp(torch.as_tensor(img, dtype=torch.float32), min=-1, max=1) * 0.5 + 0.5
    return (s.cpu().numpy().transpose(0, 2, 3, 1) * 255).astype(np.uint8)
