"""Counterpart of ``GLIGEN/gligen_inference.py`` for the text_layout modality.

The reference file is an upstream GLIGEN script that is *stale* relative to the modified UNet (it passes
no ``relations`` and calls ``prepare(batch)`` with one argument, gligen_inference.py:411-424; SURVEY
App-B#9), so only its entry-point name and argument meaning are kept; semantics follow
``GLIGEN/interface.py``:

    run(meta, config, starting_noise=None) -> list[PIL.Image]

``meta``: ``ckpt`` (path), ``prompt``, ``phrases``, ``locations`` (ltrb, normalised), optional
``alpha_type``, ``save_folder_name``; ``config``: object/dict with ``batch_size``, ``guidance_scale``,
``no_plms`` (must be False), optional ``folder``.  Images are saved like the reference does
(gligen_inference.py:437-446) when ``config.folder`` is given.
"""
from __future__ import annotations

import os

import torch

from . import interface

_MODELS = {}


def _get(cfg, key, default=None):
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


@torch.no_grad()
def run(meta, config, starting_noise=None, clip_model=None, clip_processor=None):
    """``clip_model`` / ``clip_processor``: the HF CLIP objects the reference creates inline (gligen_inference.py:96-98,
    ``openai/clip-vit-large-patch14``); pass them in to run offline -- they are only downloaded when omitted."""
    device = _get(config, "device", "cuda")
    ckpt = meta["ckpt"]
    if ckpt not in _MODELS:
        _MODELS[ckpt] = interface.load_all_models(ckpt, device)
    all_models = _MODELS[ckpt]
    # The reference reloads the checkpoint on every run() (gligen_inference.py:346), so each run starts from the GLIGEN
    # first conv; the cached model here must be put back (restore_first_conv_from_SD is permanent, openaimodel.py:393-411).
    all_models[0].first_conv_type = "GLIGEN"
    bs = _get(config, "batch_size", 1)
    args = dict(batch_size=bs, no_plms=bool(_get(config, "no_plms", False)), guidance_scale=_get(config, "guidance_scale", 7.5))
    m = dict(prompt=meta["prompt"], phrases=meta.get("phrases"), locations=meta["locations"],
             alpha_type=meta.get("alpha_type", [0.3, 0.0, 0.7]))
    if starting_noise is None:
        starting_noise = torch.randn(bs, 4, 64, 64).to(device)
    if clip_model is None or clip_processor is None:
        from transformers import CLIPModel, CLIPProcessor
        version = "openai/clip-vit-large-patch14"
        clip_model = CLIPModel.from_pretrained(version).to(device)
        clip_processor = CLIPProcessor.from_pretrained(version)
    if "steps" in (config if isinstance(config, dict) else vars(config)):
        args["steps"] = _get(config, "steps")
    images = interface.run_one_image(all_models, args, m, starting_noise, clip_model, clip_processor, device=device)
    folder = _get(config, "folder")
    if folder:
        out = os.path.join(folder, meta.get("save_folder_name", "out"))
        os.makedirs(out, exist_ok=True)
        start = len(os.listdir(out))
        for i, im in enumerate(images):
            im.save(os.path.join(out, f"{start + i}.png"))
    return images
