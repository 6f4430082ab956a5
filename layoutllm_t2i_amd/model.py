"""Host-side mirrors of the reference objects that sit on the denoising path's boundary.

  UNetModel          <- ldm.modules.diffusionmodules.openaimodel.UNetModel     (openaimodel.py:234-459)
  LatentDiffusion    <- ldm.models.diffusion.ldm.LatentDiffusion / ddpm.DDPM   (schedule buffers only)
  GroundingNetInput  <- grounding_input.text_layout_tokinzer_input.GroundingNetInput

Same attribute names, call contracts and error behaviour; the arithmetic lives in the HIP engine.
"""
from __future__ import annotations

import os
from typing import Mapping, Optional

import numpy as np
import torch

from . import host
from .arch import UNetConfig
from .engine import UNetEngine
from .weights import pack_state_dict


class LatentDiffusion:
    """Schedule holder (ddpm.py:19-54; ldm.py:12-22).  Only the buffers the sampler reads."""

    def __init__(self, linear_start: float = 0.00085, linear_end: float = 0.012, timesteps: int = 1000,
                 beta_schedule: str = "linear", device="cpu"):
        if beta_schedule != "linear":
            raise NotImplementedError("only the 'linear' beta schedule is on the reference's path (coco2014.yaml:1-6)")
        acp = host.alphas_cumprod(timesteps, linear_start, linear_end)
        betas64 = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
        self.num_timesteps = int(timesteps)
        self.linear_start, self.linear_end = linear_start, linear_end
        self.betas = torch.tensor(betas64, dtype=torch.float32, device=device)
        self.alphas_cumprod = torch.tensor(acp, dtype=torch.float32, device=device)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, acp.astype(np.float64)[:-1]), dtype=torch.float32, device=device)
        self.clip_denoised = False

    def to(self, device):
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev"):
            setattr(self, k, getattr(self, k).to(device))
        return self

    def q_sample(self, x_start, t, noise=None):
        """ldm.py:19-22 (inpainting only; kept for API parity)."""
        noise = torch.randn_like(x_start) if noise is None else noise
        a = self.alphas_cumprod.to(x_start.device)[t].sqrt().reshape(-1, 1, 1, 1)
        s = (1.0 - self.alphas_cumprod.to(x_start.device)[t]).sqrt().reshape(-1, 1, 1, 1)
        return a * x_start + s * noise


class GroundingNetInput:
    """text_layout_tokinzer_input.py:6-62: pass-through of boxes/masks/text embeddings, zeros as null."""

    def __init__(self):
        self.set = False

    def prepare(self, batch, text_encoder=None):
        self.set = True
        boxes, masks = batch["boxes"], batch["masks"]
        self.batch, self.max_box, _ = boxes.shape
        self.device = boxes.device
        self.in_dim = 768
        if "text_embeddings" in batch:
            positive_embeddings = batch["text_embeddings"]
        else:
            labels = [s.split("|") for s in batch["labels"]]
            box_list = torch.sum(masks, dim=-1).tolist()
            positive_embeddings = torch.zeros((self.batch, self.max_box, self.in_dim)).to(self.device)
            for b in range(self.batch):
                for i in range(int(box_list[b])):
                    positive_embeddings[b, i] = text_encoder.encode_one_token(labels[b][i])
        self.dtype = positive_embeddings.dtype
        return {"boxes": boxes, "masks": masks, "positive_embeddings": positive_embeddings}

    def get_null_input(self, batch=None, device=None, dtype=None):
        assert self.set, "not set yet, cannot call this funcion"
        batch = self.batch if batch is None else batch
        device = self.device if device is None else device
        dtype = self.dtype if dtype is None else dtype
        boxes = torch.zeros(batch, self.max_box, 4).type(dtype).to(device)
        masks = torch.zeros(batch, self.max_box).type(dtype).to(device)
        positive_embeddings = torch.zeros(batch, self.max_box, self.in_dim).type(dtype).to(device)
        return {"boxes": boxes, "masks": masks, "positive_embeddings": positive_embeddings}


class UNetModel:
    """Callable with the reference's ``input`` dict (keys interface.py:527-535), returns eps [B,4,h,w].

    ``state_dict``: reference-named fp32 tensors (numpy or torch).  ``sd_first_conv``: the tensors of
    GLIGEN/SD_input_conv_weight_bias.pth (``{'weight','bias'}``) that restore_first_conv_from_SD()
    switches to, permanently, exactly like the reference (openaimodel.py:393-411).
    """

    def __init__(self, cfg: UNetConfig, state_dict: Mapping[str, object], device="cuda:0",
                 sd_first_conv: Optional[Mapping[str, object]] = None, allow_missing_sd_conv: bool = False):
        self.cfg = cfg
        self.device = torch.device(device)
        self.image_size = cfg.image_size
        self.in_channels = cfg.in_channels
        self.out_channels = cfg.out_channels
        self.model_channels = cfg.model_channels
        self.first_conv_restorable = sd_first_conv is not None
        self.allow_missing_sd_conv = allow_missing_sd_conv
        self.first_conv_type = "GLIGEN"
        self.grounding_tokenizer_input: Optional[GroundingNetInput] = None   # set externally (interface.py:370)
        self.fuser_scale = 1.0          # what set_alpha_scale writes (interface.py:34-38)
        self.training = False
        packed = pack_state_dict(state_dict, cfg, self.device, sd_first_conv)
        self.engine = UNetEngine(packed)
        self._cond_key = None
        self.strict = False

    def set_strict(self, on: bool = True):
        """STRICT mode of the engine (gl_set_handle_option 50; include/gligen_hip.h): every matrix product on split-fp16 operands, output
        within north_star's rtol 1e-3 / atol 1e-4 of the fp32 reference.  Needs weights packed in the split layout
        (``UNetConfig.split_weights``; ``interface.load_ckpt(..., strict=True)`` does both)."""
        if on and not getattr(self.cfg, "split_weights", False):
            raise RuntimeError("strict mode needs the split weight layout: build the model with dataclasses.replace(cfg, split_weights=True) "
                               "(interface.load_ckpt(..., strict=True) / GLIGEN_STRICT=1 do)")
        self.engine.set_option(50, 1 if on else 0)
        self.strict = bool(on)
        self._cond_key = None           # the conditioning hoists are option-dependent: recompute on the next call
        return self

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise NotImplementedError("weights are packed for one device at construction")
        return self

    def restore_first_conv_from_SD(self):
        """openaimodel.py:393-408.  The reference's non-restorable branch exists for inpainting models only; for this
        (non-inpainting) UNet a missing SD conv would silently change 35 of 50 steps, so it is an error unless the
        model was built with ``allow_missing_sd_conv=True`` (then the reference's message is printed)."""
        if self.first_conv_restorable:
            self.first_conv_type = "SD"
        elif getattr(self, "allow_missing_sd_conv", False):
            print("First conv layer is not restorable and skipped this process, probably because this is an inpainting model?")
        else:
            raise RuntimeError("restore_first_conv_from_SD: the SD first-conv weights (GLIGEN/SD_input_conv_weight_bias.pth) were "
                               "not given to UNetModel(sd_first_conv=...); the reference switches to them on every fuser-scale-0 "
                               "step (openaimodel.py:393-405)")

    @property
    def use_sd_conv(self) -> bool:
        return self.first_conv_type == "SD" and self.first_conv_restorable

    # -- conditioning plumbing shared with the sampler
    def grounding_of(self, input: dict) -> dict:
        if "grounding_input" in input:
            return input["grounding_input"]
        if self.grounding_tokenizer_input is None:
            raise RuntimeError("model.grounding_tokenizer_input is not set (interface.py:370)")
        return self.grounding_tokenizer_input.get_null_input()

    def set_conditioning(self, context, relations, grounding: dict, hw: int, key=None) -> None:
        if key is not None and key == self._cond_key:
            return
        self.engine.set_conditioning(context, relations, grounding["boxes"], grounding["masks"],
                                     grounding["positive_embeddings"], hw)
        self._cond_key = key

    @torch.no_grad()
    def __call__(self, input: dict) -> torch.Tensor:
        """UNetModel.forward (openaimodel.py:413-459): one B-sized evaluation."""
        x = input["x"]
        g = self.grounding_of(input)
        self.set_conditioning(input["context"], input["relations"], g, x.shape[-1], key=None)
        t = input["timesteps"]
        return self.engine.forward(x.to(self.device, torch.float32).contiguous(), t, self.fuser_scale, self.use_sd_conv, 1).clone()

    forward = __call__


def load_sd_first_conv(path: Optional[str] = None) -> Optional[dict]:
    """Reads GLIGEN/SD_input_conv_weight_bias.pth once (the reference re-reads it from disk on every
    scale-0 step, openaimodel.py:397-398)."""
    if path is None or not os.path.exists(path):
        return None
    sd = torch.load(path, map_location="cpu")
    return {"weight": sd["weight"].float(), "bias": sd["bias"].float()}
