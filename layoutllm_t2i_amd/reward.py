"""Reward scoring stage of the train_rl.py rollout on the MI355X (BASELINE.json configs[4]).

Mirrors the GPU part of ``Reward_Model.forward`` (models/policy.py:106-135): CLIP text-image / image-image cosine
similarities and the AestheticMLP score (tools/aesthetic.py:15-31) run in ONE HIP kernel (``gl_reward_score``) on the
CLIP features of the rollout batch; ``reward = clip + 0.1 * aes + 10 * mIoU + 10 * DocSim`` (policy.py:135) is finished
on the host with the layout terms the reference computes in CPU python (``compute_maximum_iou`` / ``compute_docsim``),
which are passed in.  ``RewardModel`` adds the CLIP towers (``clip.ClipTowers``: get_text_features / get_image_features of
the reference's transformers.CLIPModel on the HIP kernels), i.e. the whole GPU part of ``Reward_Model.forward``; the
tokenizer and the image processor (PIL resize / crop / normalise) stay the caller's CPU objects, as in the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import torch

from . import _lib
from ._lib import RewardArgs, check

F32 = torch.float32
# nn.Sequential indices of the five Linear layers of AestheticMLP (tools/aesthetic.py:21-34; Dropouts sit in between)
_LINEARS = (0, 2, 4, 6, 7)
_SHAPES = ((1024, None), (128, 1024), (64, 128), (16, 64), (1, 16))


class RewardScorer:
    def __init__(self, aesthetic_state_dict: Mapping[str, torch.Tensor], device="cuda:0", input_size: int = 768):
        """``aesthetic_state_dict``: the AestheticMLP checkpoint (keys ``layers.{0,2,4,6,7}.{weight,bias}``,
        models/policy.py:44-46 loads ``sac+logos+ava1-l14-linearMSE.pth`` into it)."""
        if not torch.cuda.is_available():
            raise RuntimeError("RewardScorer needs a GPU: the scoring stage has no CPU fallback")
        self.device = torch.device(device)
        self.input_size = input_size
        self.w = []
        for li, (n, k) in zip(_LINEARS, _SHAPES):
            k = input_size if k is None else k
            w = torch.as_tensor(aesthetic_state_dict[f"layers.{li}.weight"], dtype=F32)
            b = torch.as_tensor(aesthetic_state_dict[f"layers.{li}.bias"], dtype=F32)
            if tuple(w.shape) != (n, k) or tuple(b.shape) != (n,):
                raise ValueError(f"layers.{li}: weight {tuple(w.shape)} / bias {tuple(b.shape)}, expected ({n}, {k}) / ({n},)")
            self.w.append((w.to(self.device).contiguous(), b.to(self.device).contiguous()))

    @torch.no_grad()
    def score(self, txt_features: torch.Tensor, img_pred_features: torch.Tensor, img_gt_features: torch.Tensor,
              miou: Optional[torch.Tensor] = None, laysim: Optional[torch.Tensor] = None) -> dict:
        """features: fp32 [B, D] CLIP embeddings (``get_text_features`` / ``get_image_features``, unnormalised).
        Returns dict(sims_ti, sims_ii, clip_reward, aes_reward, reward) of fp32 [B] device tensors; ``reward`` includes
        10 * miou + 10 * laysim when those (CPU-side layout terms, policy.py:126-133) are given."""
        f = lambda t: torch.as_tensor(t, dtype=F32).to(self.device).contiguous()
        t, p, g = f(txt_features), f(img_pred_features), f(img_gt_features)
        if not (t.shape == p.shape == g.shape) or t.dim() != 2 or t.shape[1] != self.input_size:
            raise ValueError(f"features must all be [B, {self.input_size}]")
        B = t.shape[0]
        out = torch.empty(4, B, dtype=F32, device=self.device)
        a = RewardArgs()
        a.txt, a.img_pred, a.img_gt, a.B, a.D = t.data_ptr(), p.data_ptr(), g.data_ptr(), B, self.input_size
        for i, (w, b) in enumerate(self.w, start=1):
            setattr(a, f"w{i}", w.data_ptr())
            setattr(a, f"b{i}", b.data_ptr())
        a.sims_ti, a.sims_ii, a.aesthetic, a.partial_reward = (out[i].data_ptr() for i in range(4))
        with torch.cuda.device(self.device):
            check(_lib.lib().gl_reward_score(C.byref(a), torch.cuda.current_stream(self.device).cuda_stream), "gl_reward_score")
        reward = out[3]
        if miou is not None:
            reward = reward + f(miou) * 10
        if laysim is not None:
            reward = reward + f(laysim) * 10
        return dict(sims_ti=out[0], sims_ii=out[1], clip_reward=out[0] + out[1], aes_reward=out[2], reward=reward)


class RewardModel:
    """GPU part of ``Reward_Model.forward`` (models/policy.py:106-124,135): CLIP text / image features -> cosine
    similarities + aesthetic score -> reward.  ``clip_state_dict`` = ``transformers.CLIPModel.state_dict()`` of the model
    policy.py:40 loads, ``aesthetic_state_dict`` = the AestheticMLP checkpoint (policy.py:44-46)."""

    def __init__(self, clip_state_dict, aesthetic_state_dict, device="cuda:0", vision_heads: int = 16, text_heads: int = 12, image_size: int = 224):
        from .clip import ClipTowers
        self.towers = ClipTowers(clip_state_dict, vision_heads, text_heads, device)
        self.scorer = RewardScorer(aesthetic_state_dict, device, input_size=int(self.towers.vproj.shape[0]))
        self.device = self.towers.device
        from .preprocess import ClipImagePreprocessor
        # self.processor(images=...) of policy.py:109,111, on the GPU (image_size = the checkpoint's vision_config.image_size)
        self.preprocess = ClipImagePreprocessor(size=image_size, crop_size=image_size, device=self.device)

    @torch.no_grad()
    def forward(self, input_ids, pixel_values_pred, pixel_values_gt, attention_mask=None, miou=None, laysim=None) -> dict:
        """``input_ids`` [B, T] (tokenizer output), ``pixel_values_*`` [B, 3, S, S] (processor output); returns the dict of
        RewardScorer.score plus the three feature tensors."""
        txt = self.towers.get_text_features(input_ids, attention_mask)
        pp_, pg_ = torch.as_tensor(pixel_values_pred), torch.as_tensor(pixel_values_gt)
        if pp_.shape[1:] == pg_.shape[1:]:
            # one pass of the vision tower over predictions + ground truth (rows are independent; twice the rows per GEMM)
            both = self.towers.get_image_features(torch.cat([pp_.to(self.device), pg_.to(self.device)], 0))
            pred, gt = both[:pp_.shape[0]], both[pp_.shape[0]:]
        else:
            pred = self.towers.get_image_features(pp_)
            gt = self.towers.get_image_features(pg_)
        out = self.scorer.score(txt, pred, gt, miou, laysim)
        out.update(txt_features=txt, img_pred_features=pred, img_gt_features=gt)
        return out

    @torch.no_grad()
    def forward_images(self, input_ids, images_pred, images_gt, attention_mask=None, miou=None, laysim=None,
                       pred_is_pixel_values: bool = False, gt_is_pixel_values: bool = False) -> dict:
        """``forward`` from IMAGES, as the reference's takes them (policy.py:106-111): ``images_pred`` is the VAE decoder's fp32
        output [B, 3, H, W] in [-1, 1] still on the GPU (converted to the uint8 pixels of interface.py:543-547 there), a uint8
        [B, H, W, 3] GPU tensor, or a list of PIL images; ``images_gt`` a list of PIL images / uint8 arrays of any sizes.
        The CLIP feature extractor's resize / crop / normalise runs on the GPU (preprocess.py), bit-identical to the PIL path.
        Ready ``pixel_values`` (processor output) are passed through ONLY when the caller says so with ``*_is_pixel_values``:
        an fp32 [B, 3, S, S] tensor is otherwise always taken as decoded images (a VAE output of the crop size is not
        pixel_values)."""
        def px(im, ready):
            if ready:
                if not (torch.is_tensor(im) and im.dtype == torch.float32 and im.dim() == 4 and im.shape[1] == 3):
                    raise TypeError("pixel_values must be an fp32 [B, 3, S, S] tensor")
                return im
            if torch.is_tensor(im) and im.dtype == torch.uint8:
                return self.preprocess(im.to(self.device))
            if torch.is_tensor(im):
                return self.preprocess.from_decoded(im)
            return self.preprocess.from_pil(im)
        return self.forward(input_ids, px(images_pred, pred_is_pixel_values), px(images_gt, gt_is_pixel_values), attention_mask, miou, laysim)

    __call__ = forward
