"""Recipe weights and inputs: deterministic, reference-independent tensors.

No GLIGEN checkpoint is obtainable offline (reference README.md:39 links only),
so parity tests and the benchmark use *recipe weights*: every tensor is a pure
function of ``(name, shape, seed)`` through a counter-based Philox stream, with
a per-kind scale that keeps activations O(1) through the whole UNet.  The same
recipe is evaluated in the build container (to make golden vectors by loading
the tensors into the reference classes) and on the GPU box (to feed the HIP
engine), so fixtures only need to hold *outputs*.

All ``alpha_attn`` / ``alpha_dense`` gates are non-zero: they initialise to 0 in
the reference (attention.py:219-220, :300-301) which would silence the fuser and
the rela_fuse attention/FF paths.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Tuple

import numpy as np

from .arch import UNetConfig, VAEConfig, param_shapes, vae_decoder_param_shapes


def _key(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:8], "little")


def uniform(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """float32 uniform in [-1, 1), a pure function of (name, shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    bits = np.random.Philox(key=_key(name, seed)).random_raw(n)
    u = (bits >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
    return u.reshape(shape)


def normal(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """float32 approx-normal (sum of 4 uniforms, unit variance); tails are bounded at ~3.5 sigma."""
    acc = np.zeros(shape, dtype=np.float32)
    for j in range(4):
        acc += uniform(f"{name}#n{j}", shape, seed)
    return acc * np.float32(np.sqrt(3.0 / 4.0))


def _fan_in(shape: Tuple[int, ...]) -> int:
    k = 1
    for s in shape[1:]:
        k *= s
    return max(k, 1)


def tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    """One UNet parameter by state_dict name."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("alpha_attn", "alpha_dense"):
        # non-zero gates, sign and size vary per module: tanh in roughly +-[0.3, 0.6]
        u = float(uniform(name, (1,), seed)[0])
        return np.float32(np.copysign(0.3 + 0.35 * abs(u), u if u != 0 else 1.0)).reshape(())
    if leaf.startswith("null_"):
        return uniform(name, shape, seed) * np.float32(0.5)
    is_norm = (".norm" in name or "in_layers.0" in name or "out_layers.0" in name
               or name.startswith("out.0") or "norm_out" in name)
    if is_norm and len(shape) == 1:
        if leaf == "weight":
            return np.float32(1.0) + np.float32(0.1) * uniform(name, shape, seed)
        return np.float32(0.05) * uniform(name, shape, seed)
    if leaf == "bias":
        return np.float32(0.02) * uniform(name, shape, seed)
    # weights: uniform with variance 1/fan_in  (sqrt(3/fan_in) half-width)
    bound = np.float32(np.sqrt(3.0 / _fan_in(shape)))
    return uniform(name, shape, seed) * bound


def state_dict(cfg: UNetConfig, seed: int = 0, only_prefix: str | None = None) -> Dict[str, np.ndarray]:
    """Full (or prefix-filtered) recipe state_dict as float32 numpy arrays."""
    out = {}
    for name, shape in param_shapes(cfg).items():
        if only_prefix is not None and not name.startswith(only_prefix):
            continue
        out[name] = tensor(name, shape, seed)
    return out


def vae_state_dict(cfg: VAEConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Recipe weights of the VAE decode path (reference names), float32."""
    return {n: tensor("vae." + n, shp, seed) for n, shp in vae_decoder_param_shapes(cfg).items()}


def sd_first_conv(cfg: UNetConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Stand-in for GLIGEN/SD_input_conv_weight_bias.pth (openaimodel.py:397-402): the
    'SD' first conv that replaces input_blocks.0.0 on every fuser-scale-0 step.  Deliberately
    different from the recipe's GLIGEN first conv so the switch is observable in tests."""
    w = tensor("SD_first_conv.weight", (cfg.model_channels, cfg.in_channels, 3, 3), seed)
    b = tensor("SD_first_conv.bias", (cfg.model_channels,), seed)
    return {"weight": w, "bias": b}


# --------------------------------------------------------------------------- inputs

_GRID8 = [  # fixed 8-box layout (ltrb, normalised) used when no dataset layout is supplied
    (0.05, 0.05, 0.45, 0.40), (0.55, 0.05, 0.95, 0.45), (0.10, 0.50, 0.40, 0.95), (0.50, 0.55, 0.90, 0.90),
    (0.30, 0.30, 0.70, 0.70), (0.00, 0.70, 0.25, 1.00), (0.70, 0.00, 1.00, 0.30), (0.20, 0.10, 0.80, 0.25),
]


def synth_inputs(cfg: UNetConfig, batch: int, hw: int, n_boxes: int = 8, n_rel: int = 3,
                 max_relations: int = 10, seed: int = 1234, boxes=None) -> Dict[str, np.ndarray]:
    """Synthetic conditioning of the shapes interface.py:527-535 feeds the UNet.

    context/uc ~ N(0,1) [B,77,ctx]; relations: first n_rel rows N(0,1), rest 0, [B,R,ctx];
    boxes [B,30,4] ltrb with the first n_boxes valid; masks [B,30]; text embeddings N(0,1) on
    valid rows.  Latent x ~ N(0,1) [B,4,hw,hw].
    """
    mo = cfg.max_objs
    x = normal("in.x", (batch, cfg.in_channels, hw, hw), seed)
    context = normal("in.context", (batch, 77, cfg.context_dim), seed)
    uc = np.repeat(normal("in.uc", (1, 77, cfg.context_dim), seed), batch, axis=0)
    relations = np.zeros((batch, max_relations, cfg.context_dim), np.float32)
    relations[:, :n_rel] = normal("in.relations", (batch, n_rel, cfg.context_dim), seed)
    bx = np.zeros((batch, mo, 4), np.float32)
    masks = np.zeros((batch, mo), np.float32)
    emb = np.zeros((batch, mo, cfg.pos_in_dim), np.float32)
    for b in range(batch):
        for i in range(n_boxes):
            if boxes is not None:
                bx[b, i] = boxes[b][i]
            else:
                g = _GRID8[(i + 3 * b) % len(_GRID8)]
                # shrink slightly per (b, i) so samples differ but stay inside [0,1]
                s = 0.02 * ((b + i) % 3)
                bx[b, i] = (g[0] + s, g[1] + s, g[2] - s, g[3] - s)
        masks[b, :n_boxes] = 1.0
    emb[:, :n_boxes] = normal("in.text_embeddings", (batch, n_boxes, cfg.pos_in_dim), seed)
    return dict(x=x, context=context, uc=uc, relations=relations, boxes=bx, masks=masks,
                positive_embeddings=emb)
