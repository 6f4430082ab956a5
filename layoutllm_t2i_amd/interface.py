"""Drop-in counterpart of ``GLIGEN/interface.py``: same function names, signatures, argument meaning
and quirks, with the denoiser (UNet + PLMS) running on the MI355X HIP engine.

    load_all_models(ckpt, device)                     interface.py:366-373
    load_ckpt(ckpt_path, device)                      interface.py:78-101
    generate_batch_images(all_models, captions, labels, bboxes, clip_model, clip_processor, device)  :551-570
    generate_one_image(all_models, caption, label, bbox, clip_model, clip_processor, device)          :376-395
    run_batch_images / run_one_image(all_models, args, meta, starting_noise, clip_model, clip_processor, device)
    set_alpha_scale, alpha_generator, prepare_batch, prepare_batch_multiple, prepare_relation_phrases,
    convert_xywh_to_ltrb, convert_xcycwh_to_ltrb

``all_models`` stays the reference's 5-tuple ``(model, autoencoder, text_encoder, diffusion, config)``.
``model`` is this package's UNetModel; ``autoencoder`` / ``text_encoder`` are whatever the caller
provides with the reference's ``decode(z)`` / ``encode(list[str], return_pooler_output=...)`` methods
(the reference's own CPU modules work unchanged -- VAE and CLIP are SURVEY 8f "next" rows, and the
LLM / policy orchestration stays on the reference path by design).

Preserved behaviour: alpha_type [0.3, 0, 0.7]; 50 PLMS steps; CFG 7.5; noise from the
global CPU RNG ``torch.randn(bs, 4, 64, 64)``; ``generate_batch_images`` passes boxes through
UNconverted while ``generate_one_image`` converts xywh -> ltrb; ``config.update(args)`` mutates the
caller's dict; relation phrases are "PAD" + every relation twice, truncated to max_relations;
clamp -> *0.5+0.5 -> *255 -> astype(uint8) truncation.
"""
from __future__ import annotations

import os
import warnings
from functools import partial
from typing import Union

import numpy as np
import torch

from . import host
from .arch import UNetConfig
from .arch import VAEConfig
from .model import GroundingNetInput, LatentDiffusion, UNetModel, load_sd_first_conv
from .vae import VAEDecoder
from .sampler import PLMSSampler

MAX_OBJS = 30
PLMS_STEPS = 50


def set_alpha_scale(model, alpha_scale):
    """interface.py:34-38: only the gated self-attention fuser is scaled; rela_fuse.scale stays 1."""
    model.fuser_scale = alpha_scale


alpha_generator = host.alpha_generator


class _AttrDict(dict):
    """Enough of OmegaConf.create(dict) for run_*_images: attribute access on a dict."""
    __getattr__ = dict.__getitem__


def _instantiate_reference(config_node):
    """instantiate_from_config (ldm/util.py:71-85) for the non-hot-path modules (VAE, CLIP text encoder):
    resolved from whatever ``ldm`` package is importable (the reference's, on its CPU path)."""
    import importlib
    module, cls = config_node["target"].rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)(**config_node.get("params", dict()))


def _load_text_encoder(node, state_dict, device):
    """The conditioning encoder (interface.py:85-88).  A checkpoint whose ``text_encoder`` entry holds a CLIP text tower
    (``FrozenCLIPEmbedder.state_dict()``: ``transformer.text_model.*``) runs on the HIP kernels (text_encoder.HipCLIPTextEncoder,
    SURVEY 8f-2); its tokenizer is the config's ``params.tokenizer`` node when present, else HuggingFace's CLIPTokenizer of
    ``params.version`` exactly as FrozenCLIPEmbedder loads it (encoders/modules.py:148).  Anything else -- or
    GLIGEN_REFERENCE_TEXT_ENCODER=1 -- is instantiated from whatever ``ldm`` package is importable, like the reference does."""
    from .text_encoder import HipCLIPTextEncoder
    params = node.get("params", {}) or {}
    if not os.environ.get("GLIGEN_REFERENCE_TEXT_ENCODER") and HipCLIPTextEncoder.accepts(state_dict):
        if params.get("tokenizer") is not None:
            tokenizer = _instantiate_reference(params["tokenizer"])
        else:
            from transformers import CLIPTokenizer
            tokenizer = CLIPTokenizer.from_pretrained(params.get("version", "openai/clip-vit-large-patch14"))
        return HipCLIPTextEncoder(state_dict, tokenizer, device, heads=params.get("num_attention_heads"), max_length=params.get("max_length", 77))
    text_encoder = _instantiate_reference(node).to(device).eval()
    text_encoder.load_state_dict(state_dict)
    return text_encoder


def hip_phrase_encoder(clip_model, device, heads=None):
    """The ``clip_model`` argument of generate_batch_images / generate_one_image (a HuggingFace ``CLIPModel``, used only for the
    grounding phrases' ``text_model_output.pooler_output``, interface.py:114-141) moved onto the HIP text tower: pass the result
    as ``clip_model`` instead; ``clip_processor`` keeps tokenising.  ``heads`` defaults to hidden / 64."""
    from .text_encoder import HipCLIPTextEncoder
    sd = clip_model if isinstance(clip_model, dict) else clip_model.state_dict()
    if heads is None and hasattr(clip_model, "config"):
        heads = getattr(getattr(clip_model.config, "text_config", None), "num_attention_heads", None)
    return HipCLIPTextEncoder(sd, None, device, heads=heads)


def find_sd_first_conv(ckpt_path=None):
    """Locates ``SD_input_conv_weight_bias.pth``.  The reference reads it from the GLIGEN code directory on every
    scale-0 step (openaimodel.py:396-398) and fails hard when it is missing.  Search order: $GLIGEN_SD_FIRST_CONV,
    next to the checkpoint, $GLIGEN_HOME, and the directory the reference itself computes -- four levels above
    ``ldm/modules/diffusionmodules/openaimodel.py`` of whatever ``ldm`` package is importable.  Returns None if absent."""
    name = "SD_input_conv_weight_bias.pth"
    cands = []
    if os.environ.get("GLIGEN_SD_FIRST_CONV"):
        cands.append(os.environ["GLIGEN_SD_FIRST_CONV"])
    if ckpt_path is not None:
        cands.append(os.path.join(os.path.dirname(os.path.abspath(ckpt_path)), name))
    if os.environ.get("GLIGEN_HOME"):
        cands.append(os.path.join(os.environ["GLIGEN_HOME"], name))
    try:
        import importlib.util
        spec = importlib.util.find_spec("ldm")
        if spec is not None and spec.submodule_search_locations:
            cands.append(os.path.join(os.path.dirname(list(spec.submodule_search_locations)[0]), name))
    except (ImportError, ValueError):
        pass
    for c in cands:
        if os.path.exists(c):
            return c
    return None


def load_ckpt(ckpt_path, device="cuda", strict=None):
    """interface.py:78-101.  The UNet ('model') is built by this package from saved_ckpt['model'];
    autoencoder / text_encoder / grounding tokenizer are instantiated from the checkpoint's config.

    ``strict`` (default: the environment variable ``GLIGEN_STRICT=1``): pack the UNet's weights in the split layout ([Whi | Wlo] for
    every matrix, twice the bytes) and run the engine in STRICT mode (gl_set_handle_option 50: split-fp16 operands for every matrix
    product): the UNet output is then within rtol 1e-3 / atol 1e-4 of the fp32 reference (measured rel-L2 < 1e-5) at ~0.47 x the default
    mode's images/s.  ``model.set_strict(False)`` switches the same handle back to the default (fast) arithmetic.

    Like the reference (openaimodel.py:393-405) the SD first-conv file is needed as soon as a fuser-scale-0 step runs:
    a missing file warns here and raises in ``restore_first_conv_from_SD`` (where the reference fails), never silently
    keeps the GLIGEN conv.  ``GLIGEN_ALLOW_NO_SD_CONV=1`` opts out explicitly (first_conv_restorable = False)."""
    saved_ckpt = torch.load(ckpt_path, map_location="cpu")
    config = saved_ckpt["config_dict"]["_content"]
    cfg = UNetConfig.from_dict(config["model"]["params"])
    if strict is None:
        strict = os.environ.get("GLIGEN_STRICT") == "1"
    if strict:
        import dataclasses
        cfg = dataclasses.replace(cfg, split_weights=True)
    sd_path = find_sd_first_conv(ckpt_path)
    allow_missing = os.environ.get("GLIGEN_ALLOW_NO_SD_CONV") == "1"
    if sd_path is None and not allow_missing:
        warnings.warn(
            "SD_input_conv_weight_bias.pth not found (looked at $GLIGEN_SD_FIRST_CONV, next to the checkpoint, $GLIGEN_HOME and "
            "the importable ldm package's GLIGEN directory): the first fuser-scale-0 step will raise, exactly where the reference "
            "fails (openaimodel.py:393-405); schedules without a scale-0 stage run.  GLIGEN_ALLOW_NO_SD_CONV=1 keeps the GLIGEN conv "
            "instead (results then differ from the reference).")
    model = UNetModel(cfg, saved_ckpt["model"], device=device, sd_first_conv=load_sd_first_conv(sd_path),
                      allow_missing_sd_conv=sd_path is None and allow_missing)
    if strict:
        model.set_strict(True)
    dparams = config["diffusion"].get("params", {})
    diffusion = LatentDiffusion(linear_start=dparams.get("linear_start", 0.00085), linear_end=dparams.get("linear_end", 0.012),
                                timesteps=dparams.get("timesteps", 1000), device=device)
    if os.environ.get("GLIGEN_REFERENCE_VAE"):
        autoencoder = _instantiate_reference(config["autoencoder"]).to(device).eval()
        autoencoder.load_state_dict(saved_ckpt["autoencoder"])
    else:
        # decode stage on the HIP kernels; only `.decode(z)` is used on this path (interface.py:541)
        ap = config["autoencoder"].get("params", {})
        dd = ap.get("ddconfig", {})
        vcfg = VAEConfig(ch=dd.get("ch", 128), ch_mult=tuple(dd.get("ch_mult", (1, 2, 4, 4))),
                         num_res_blocks=dd.get("num_res_blocks", 2), z_channels=dd.get("z_channels", 4),
                         out_ch=dd.get("out_ch", 3), embed_dim=ap.get("embed_dim", 4), scale_factor=ap.get("scale_factor", 0.18215))
        autoencoder = VAEDecoder(saved_ckpt["autoencoder"], vcfg, device)
    text_encoder = _load_text_encoder(config["text_encoder"], saved_ckpt["text_encoder"], device)
    for m in (autoencoder, text_encoder):
        # (reference modules keep the device string they .to()'d; the HIP stages own a resolved torch.device and keep it)
        if not isinstance(m, VAEDecoder) and not _is_hip_encoder(m) and "device" in vars(m):
            m.device = device
    return model, autoencoder, text_encoder, diffusion, config


def load_all_models(ckpt, device, strict=None):
    """interface.py:366-373 (``strict``: see load_ckpt)."""
    model, autoencoder, text_encoder, diffusion, config = load_ckpt(ckpt, device, strict=strict)
    model.grounding_tokenizer_input = GroundingNetInput()
    return model, autoencoder, text_encoder, diffusion, config


def convert_xcycwh_to_ltrb(bbox):
    xc, yc, w, h = bbox
    return [xc - w / 2, yc - h / 2, xc + w / 2, yc + h / 2]


def convert_xywh_to_ltrb(bbox):
    x1, y1, w, h = bbox
    return [x1, y1, x1 + w, y1 + h]


def complete_mask(has_mask, max_objs):
    mask = torch.ones(1, max_objs)
    if has_mask is None:
        return mask
    if type(has_mask) == int or type(has_mask) == float:
        return mask * has_mask
    for idx, value in enumerate(has_mask):
        mask[0, idx] = value
    return mask


def _hf_text_pooler_output(model, input_ids, attention_mask, device):
    """``outputs.text_model_output.pooler_output`` of interface.py:136-139 under the reference's pinned transformers 4.19.2: the
    text tower's pooled hidden state BEFORE ``text_projection`` (``which_layer_text = 'before'``, :115).  transformers 5.x
    overwrites that field with the PROJECTED feature inside ``CLIPModel.forward``; calling the text tower itself gives the
    4.19.2 value under every version (and skips the placeholder vision pass)."""
    if hasattr(model, "text_model"):
        return model.text_model(input_ids=input_ids, attention_mask=attention_mask).pooler_output
    return model(input_ids=input_ids, attention_mask=attention_mask, pixel_values=torch.ones(1, 3, 224, 224).to(device)).text_model_output.pooler_output


def get_clip_feature(model, processor, input, device, is_image=False):
    """interface.py:114-141, text branch ('before' projection = pooler_output).  Image grounding is
    not on the text_layout path."""
    if input is None:
        return None
    if is_image:
        raise NotImplementedError("image grounding tokens are not on the text_layout path")
    inputs = processor(text=input, return_tensors="pt", padding=True)
    return _hf_text_pooler_output(model, inputs["input_ids"].to(device), inputs["attention_mask"].to(device), device)


def get_clip_features_batched(model, processor, phrases, device):
    """Conditioning-prep batching.  The reference calls ``get_clip_feature`` once per phrase,
    sequentially (interface.py:446-448: B x n_boxes CLIP forwards, each also running the vision tower on a
    placeholder image).  Here every DISTINCT phrase of the whole batch goes through ONE padded forward; CLIP's
    text tower is causal and the pooled feature is read at the EOS position, so padding after EOS cannot change
    it (checked against the per-phrase path in tests/test_host_cpu.py).  Returns {phrase: [1, 768] feature}."""
    uniq = [p for p in dict.fromkeys(phrases) if p is not None]
    if not uniq:
        return {}
    inputs = processor(text=uniq, return_tensors="pt", padding=True)
    if hasattr(model, "pooler_output") and callable(model.pooler_output):
        # the HIP text tower (hip_phrase_encoder / text_encoder.HipCLIPTextEncoder): token rows in, pooled rows out; no vision pass
        pooled = model.pooler_output(inputs["input_ids"])
        return {p: pooled[i:i + 1] for i, p in enumerate(uniq)}
    pooled = _hf_text_pooler_output(model, inputs["input_ids"].to(device), inputs["attention_mask"].to(device), device)
    return {p: pooled[i:i + 1] for i, p in enumerate(uniq)}


def _one_sample_grounding(phrases, locations, model, processor, max_objs, device, feature_cache=None):
    boxes = torch.zeros(max_objs, 4)
    masks = torch.zeros(max_objs)
    text_masks = torch.zeros(max_objs)
    image_masks = torch.zeros(max_objs)
    text_embeddings = torch.zeros(max_objs, 768)
    image_embeddings = torch.zeros(max_objs, 768)
    if phrases is None:
        phrases = [None] * len(locations)           # interface.py:166: zero embeddings, boxes still grounded
    if feature_cache is None:
        feature_cache = get_clip_features_batched(model, processor, phrases, device)
    feats = [None if ph is None else feature_cache[ph] for ph in phrases]
    for idx, (box, feat) in enumerate(zip(locations, feats)):
        boxes[idx] = torch.tensor(box)
        masks[idx] = 1
        if feat is not None:
            text_embeddings[idx] = feat
            text_masks[idx] = 1
    return boxes, masks, text_masks, image_masks, text_embeddings, image_embeddings


@torch.no_grad()
def prepare_batch(meta, model, processor, batch=1, max_objs=MAX_OBJS, device=None):
    """interface.py:156-193: one layout repeated `batch` times."""
    boxes, masks, tm, im, te, ie = _one_sample_grounding(meta.get("phrases"), meta["locations"], model, processor, max_objs, device)
    out = {
        "boxes": boxes.unsqueeze(0).repeat(batch, 1, 1),
        "masks": masks.unsqueeze(0).repeat(batch, 1),
        "text_masks": tm.unsqueeze(0).repeat(batch, 1) * complete_mask(meta.get("text_mask"), max_objs),
        "image_masks": im.unsqueeze(0).repeat(batch, 1) * complete_mask(meta.get("image_mask"), max_objs),
        "text_embeddings": te.unsqueeze(0).repeat(batch, 1, 1),
        "image_embeddings": ie.unsqueeze(0).repeat(batch, 1, 1),
    }
    return {k: v.to(device) for k, v in out.items()}


@torch.no_grad()
def prepare_batch_multiple(meta, model, processor, batch=1, max_objs=MAX_OBJS, device=None):
    """interface.py:424-475: one layout per prompt."""
    phrases_batch = meta.get("phrases")
    if phrases_batch is None:
        phrases_batch = [None] * len(meta["locations"])
    phrases_batch = [[None] * len(loc) if ph is None else ph for ph, loc in zip(phrases_batch, meta["locations"])]
    assert batch == len(phrases_batch)
    cols = [[] for _ in range(6)]
    cache = get_clip_features_batched(model, processor, [ph for phrases in phrases_batch for ph in phrases], device)
    for i, phrases in enumerate(phrases_batch):
        parts = _one_sample_grounding(phrases, meta["locations"][i], model, processor, max_objs, device, cache)
        parts = list(parts)
        parts[2] = parts[2].unsqueeze(0) * complete_mask(meta.get("text_mask"), max_objs)
        parts[3] = parts[3].unsqueeze(0) * complete_mask(meta.get("image_mask"), max_objs)
        for j in (0, 1, 4, 5):
            parts[j] = parts[j].unsqueeze(0)
        for j in range(6):
            cols[j].append(parts[j])
    names = ("boxes", "masks", "text_masks", "image_masks", "text_embeddings", "image_embeddings")
    return {n: torch.cat(c, dim=0).to(device) for n, c in zip(names, cols)}


def _relation_phrases(prompt, max_relas):
    """interface.py:221-240: scene-graph triplets "subject relation object"; 'PAD' first, each relation listed
    twice, truncated to max_relas.  [] when the prompt has no relation."""
    import sng_parser   # same optional dependency as the reference (interface.py:8)
    graph = sng_parser.parse(prompt)
    entities = graph["entities"]
    triplets = []
    for r in graph.get("relations", []):
        triplets.append(" ".join([entities[r["subject"]]["lemma_head"], r["relation"], entities[r["object"]]["lemma_head"]]))
    return (["PAD"] + triplets + triplets)[:max_relas] if triplets else []


@torch.no_grad()
def prepare_relation_phrases_batch(prompts, max_relas=5, text_encoder=None, device=None, parse=_relation_phrases):
    """One ``text_encoder.encode`` call for the relation phrases of ALL prompts (the reference encodes them prompt
    by prompt, interface.py:490-496).  FrozenCLIPEmbedder pads every row to 77 tokens (modules.py:159-161), so
    batching cannot change a row's pooled embedding.  Returns [len(prompts), max_relas, 768], zero-padded."""
    lists = [parse(p, max_relas) for p in prompts]
    flat = [ph for l in lists for ph in l]
    emb = torch.zeros(len(prompts), max_relas, 768)
    if flat:
        _, pooled = text_encoder.encode(flat, return_pooler_output=True)
        pooled = pooled.to(emb.dtype).cpu()
        k = 0
        for i, l in enumerate(lists):
            emb[i, :len(l), :] = pooled[k:k + len(l)]
            k += len(l)
    return emb.to(device)


@torch.no_grad()
def prepare_relation_phrases(prompt, batch_size=1, max_relas=5, text_encoder=None, device=None):
    """interface.py:221-252: one prompt's relation embeddings repeated batch_size times."""
    emb = prepare_relation_phrases_batch([prompt], max_relas, text_encoder, device)
    return emb.repeat(batch_size, 1, 1)


def _postprocess(samples):
    """interface.py:543-547: clamp -> [0,1] -> *255 -> uint8 truncation -> PIL."""
    from PIL import Image
    out = []
    for sample in samples:
        sample = torch.clamp(sample, min=-1, max=1) * 0.5 + 0.5
        sample = sample.cpu().numpy().transpose(1, 2, 0) * 255
        out.append(Image.fromarray(sample.astype(np.uint8)))
    return out


@torch.no_grad()
def denoise(all_models, context, uc, relations, grounding_batch, starting_noise, alpha_type=None, guidance_scale=7.5,
            steps=PLMS_STEPS):
    """The denoising hot path proper, from conditioning tensors to the final latent
    (run_batch_images lines interface.py:505-539 without text/VAE stages)."""
    model, autoencoder, text_encoder, diffusion, config = all_models
    sampler = PLMSSampler(diffusion, model, alpha_generator_func=partial(alpha_generator, type=alpha_type),
                          set_alpha_scale=set_alpha_scale)
    grounding_input = model.grounding_tokenizer_input.prepare(grounding_batch, text_encoder)
    input = dict(x=starting_noise, timesteps=None, context=context, relations=relations, grounding_input=grounding_input,
                 inpainting_extra_input=None, grounding_extra_input=None)
    shape = tuple(starting_noise.shape)
    return sampler.sample(S=steps, shape=shape, input=input, uc=uc, guidance_scale=guidance_scale, mask=None, x0=None)


def _run(all_models, args, meta, starting_noise, clip_model, clip_processor, device, multiple):
    model, autoencoder, text_encoder, diffusion, config = all_models
    config.update(args)                      # mutates the caller's dict, like interface.py:297/484
    cfg = _AttrDict(config)
    if cfg.get("no_plms", False):
        raise NotImplementedError("the DDIM path is broken in the reference with this UNet (SURVEY App-B#8); PLMS only")
    bs = cfg.batch_size
    max_rel = cfg.get("max_relations", 10)
    if multiple:
        batch = prepare_batch_multiple(meta, clip_model, clip_processor, bs, device=device)
        context = text_encoder.encode(meta["prompts"])
        relations = prepare_relation_phrases_batch(meta["prompts"], max_rel, text_encoder, device=device)
    else:
        batch = prepare_batch(meta, clip_model, clip_processor, bs, device=device)
        context = text_encoder.encode([meta["prompt"]] * bs)
        relations = prepare_relation_phrases(meta["prompt"], bs, max_rel, text_encoder, device=device)
    uc = text_encoder.encode([""]).repeat(bs, 1, 1)          # the reference encodes bs copies of "" (interface.py:496)
    # S is a harness parameter: the reference hard-codes 50 (interface.py:507); ``args["steps"]`` overrides it
    samples = denoise(all_models, context, uc, relations, batch, starting_noise, meta.get("alpha_type"), cfg.guidance_scale,
                      steps=int(args.get("steps", PLMS_STEPS)))        # per call: not sticky through the cached config dict
    return _postprocess(autoencoder.decode(samples))


@torch.no_grad()
def run_one_image(all_models, args, meta, starting_noise=None, clip_model=None, clip_processor=None, device=None):
    """interface.py:292-357."""
    return _run(all_models, args, meta, starting_noise, clip_model, clip_processor, device, multiple=False)


@torch.no_grad()
def run_batch_images(all_models, args, meta, starting_noise=None, clip_model=None, clip_processor=None, device=None):
    """interface.py:478-549."""
    return _run(all_models, args, meta, starting_noise, clip_model, clip_processor, device, multiple=True)


def generate_one_image(all_models, caption, label, bbox, clip_model=None, clip_processor=None, device=None):
    """interface.py:376-395 (converts xywh -> ltrb, :383)."""
    args = dict(batch_size=1, no_plms=False, guidance_scale=7.5)
    bbox = [convert_xywh_to_ltrb(b) for b in bbox]
    meta = dict(prompt=caption, phrases=label, locations=bbox, alpha_type=[0.3, 0.0, 0.7])
    starting_noise = torch.randn(args["batch_size"], 4, 64, 64).to(device)
    warnings.filterwarnings("ignore", category=DeprecationWarning)
    return run_one_image(all_models, args, meta, starting_noise, clip_model, clip_processor, device=device)


def generate_batch_images(all_models, captions, labels, bboxes, clip_model=None, clip_processor=None, device=None):
    """interface.py:551-570 (boxes are NOT converted here, App-B#10)."""
    bs = len(captions)
    args = dict(batch_size=bs, no_plms=False, guidance_scale=7.5)
    meta = dict(prompts=captions, phrases=labels, locations=bboxes, alpha_type=[0.3, 0.0, 0.7])
    starting_noise = torch.randn(bs, 4, 64, 64).to(device)
    warnings.filterwarnings("ignore", category=DeprecationWarning)
    return run_batch_images(all_models, args, meta, starting_noise, clip_model, clip_processor, device=device)


# ------------------------------------------------------------------------------------------------------------------
# Multi-GPU product entry.  The reference's inference path is single-GPU (txt2img.py:535-536,
# train_rl.py:321); prompts are independent units, so the 8-GPU mode is: rank 0 reads the checkpoint, ONE broadcast
# carries the packed UNet + VAE weights, every call shards the prompts round-robin (rank r takes prompts r, r + world, ...),
# per-PROMPT seeds make a sample's result independent of the rank that ran it, and the uint8 images are gathered on rank 0.
# One process per GPU under torch.distributed ("nccl" = RCCL over xGMI on ROCm; "gloo" in the tests).
# ------------------------------------------------------------------------------------------------------------------
def _unet_facade(packed, cfg, device, allow_missing_sd_conv=False):
    from .engine import UNetEngine
    m = UNetModel.__new__(UNetModel)
    m.cfg, m.device = cfg, torch.device(device)
    m.image_size, m.in_channels, m.out_channels, m.model_channels = cfg.image_size, cfg.in_channels, cfg.out_channels, cfg.model_channels
    m.first_conv_restorable, m.allow_missing_sd_conv, m.first_conv_type = bool(packed.has_sd_conv), bool(allow_missing_sd_conv), "GLIGEN"
    m.grounding_tokenizer_input = GroundingNetInput()
    m.fuser_scale, m.training, m._cond_key = 1.0, False, None
    m.engine = UNetEngine(packed)
    m.strict = False
    return m


def load_all_models_sharded(ckpt, device, src=0, strict=None):
    """``load_all_models`` for a torch.distributed job: only rank ``src`` touches the checkpoint file; the others receive the
    packed UNet (the C engine's flat weight buffer), the packed VAE decoder and -- when the checkpoint's text encoder runs on the HIP
    tower -- its weights in dist.broadcast_bundle's single collective, plus the config dict.  With a reference torch text encoder
    the receiving ranks get ``text_encoder = None``: conditioning is then prepared on ``src`` and scattered per call
    (generate_batch_images_sharded).  Returns the usual 5-tuple on every rank."""
    import torch.distributed as dist
    from .dist import broadcast_bundle
    if not (dist.is_available() and dist.is_initialized()):
        return load_all_models(ckpt, device, strict=strict)
    rank = dist.get_rank()
    if rank == src:
        am = load_all_models(ckpt, device, strict=strict)
        model, autoencoder, text_encoder, diffusion, config = am
        if not isinstance(autoencoder, VAEDecoder):
            raise NotImplementedError("the sharded entry broadcasts the HIP VAE decoder's packed weights (unset GLIGEN_REFERENCE_VAE)")
        dcfg = dict(linear_start=diffusion.linear_start, linear_end=diffusion.linear_end, timesteps=diffusion.num_timesteps)
        # the HIP text tower travels too (fp32 state dict, ~0.5 GB for ViT-L/14's text model, same single broadcast) together with its
        # (picklable, host-side) tokenizer, so that every rank encodes its own prompts; a reference torch encoder stays on src
        hip_te = _is_hip_encoder(text_encoder)
        te_extra = dict(tokenizer=text_encoder.tokenizer, heads=text_encoder.heads, max_length=text_encoder.max_length) if hip_te else None
        broadcast_bundle(model.engine.P, autoencoder.W, model.cfg, device, src,
                         extra=dict(config=config, vcfg=autoencoder.cfg, diffusion=dcfg, text_encoder=te_extra,
                                    allow_missing_sd_conv=bool(getattr(model, "allow_missing_sd_conv", False)),
                                    strict=bool(getattr(model, "strict", False))),
                         aux=dict(text_encoder=text_encoder.towers_state_dict()) if hip_te else None)
        return am
    P, vw, extra, aux = broadcast_bundle(None, None, None, device, src, want_aux=True)
    model = _unet_facade(P, P.cfg, device, extra.get("allow_missing_sd_conv", False))     # src's GLIGEN_ALLOW_NO_SD_CONV applies to every rank
    if extra.get("strict"):                 # ... and so does its strict mode (the split weight layout travelled in the broadcast)
        model.set_strict(True)
    autoencoder = VAEDecoder.from_packed(vw, extra["vcfg"], device)
    d = extra["diffusion"]
    diffusion = LatentDiffusion(linear_start=d["linear_start"], linear_end=d["linear_end"], timesteps=d["timesteps"], device=device)
    text_encoder = None
    if extra.get("text_encoder") is not None:
        from .text_encoder import HipCLIPTextEncoder
        te = extra["text_encoder"]
        text_encoder = HipCLIPTextEncoder(aux["text_encoder"], te["tokenizer"], device, heads=te["heads"], max_length=te["max_length"])
    return model, autoencoder, text_encoder, diffusion, extra["config"]


def _collective_device(device):
    """Object collectives of the RCCL backend stage their bytes on the CURRENT device: make that this rank's GPU even when the
    caller never called torch.cuda.set_device (all ranks staging on GPU 0 is an RCCL 'duplicate GPU' error)."""
    import contextlib
    d = torch.device(device)
    return torch.cuda.device(d) if d.type == "cuda" else contextlib.nullcontext()


@torch.no_grad()
def prepare_conditioning(all_models, captions, labels, bboxes, clip_model, clip_processor, device):
    """Everything ``run_batch_images`` feeds the sampler (interface.py:486-535), per prompt row, on the CPU: context / uc
    [n, 77, 768], relations [n, R, 768], boxes / masks / text_embeddings [n, 30, ...].  Rows are independent, so any subset of
    rows is a valid batch."""
    model, autoencoder, text_encoder, diffusion, config = all_models
    n = len(captions)
    meta = dict(prompts=captions, phrases=labels, locations=bboxes)
    batch = prepare_batch_multiple(meta, clip_model, clip_processor, n, device=device)
    cond = dict(context=text_encoder.encode(captions), uc=text_encoder.encode([""]).repeat(n, 1, 1),
                relations=prepare_relation_phrases_batch(captions, config.get("max_relations", 10), text_encoder, device=device),
                boxes=batch["boxes"], masks=batch["masks"], text_embeddings=batch["text_embeddings"])
    return {k: v.detach().float().cpu().contiguous() for k, v in cond.items()}


def _is_hip_encoder(m) -> bool:
    from .text_encoder import HipCLIPTextEncoder
    return isinstance(m, HipCLIPTextEncoder)


def tokenize_conditioning(all_models, captions, labels, bboxes, clip_processor, max_objs=MAX_OBJS):
    """Host-side half of the conditioning prep for the HIP encoders: every string of ``run_batch_images``'s prep
    (interface.py:424-475, :486-496) becomes token rows -- no GPU work, a few KB per prompt, so rank ``src`` does it for the
    whole call and the ranks encode their own rows (encode_conditioning).  Returns a dict of CPU tensors:
    cap_ids [n, 77], uc_ids [1, 77], rel_ids [m, 77] + rel_owner / rel_slot [m] (prompt, slot) of each relation phrase,
    phrase_ids [k, L] (the distinct grounding phrases, ``clip_processor`` padding) + phrase_index [n, max_objs] (-1 = none),
    boxes [n, max_objs, 4], masks [n, max_objs]."""
    model, autoencoder, text_encoder, diffusion, config = all_models
    n = len(captions)
    R = config.get("max_relations", 10)
    rel_lists = [_relation_phrases(p_, R) for p_ in captions]
    flat = [ph for l in rel_lists for ph in l]
    T = text_encoder.max_length
    phrases_batch = labels if labels is not None else [None] * n
    phrases_batch = [[None] * len(loc) if ph is None else ph for ph, loc in zip(phrases_batch, bboxes)]
    uniq = [p_ for p_ in dict.fromkeys(ph for phrases in phrases_batch for ph in phrases) if p_ is not None]
    where = {p_: i for i, p_ in enumerate(uniq)}
    boxes = torch.zeros(n, max_objs, 4)
    masks = torch.zeros(n, max_objs)
    pidx = torch.full((n, max_objs), -1, dtype=torch.long)
    for i, (phrases, locs) in enumerate(zip(phrases_batch, bboxes)):
        for j, (box, ph) in enumerate(zip(locs, phrases)):
            boxes[i, j] = torch.tensor(box)
            masks[i, j] = 1
            if ph is not None:
                pidx[i, j] = where[ph]
    return dict(cap_ids=text_encoder.tokenize(captions), uc_ids=text_encoder.tokenize([""]),
                rel_ids=text_encoder.tokenize(flat) if flat else torch.zeros(0, T, dtype=torch.long),
                rel_owner=torch.tensor([i for i, l in enumerate(rel_lists) for _ in l], dtype=torch.long),
                rel_slot=torch.tensor([j for l in rel_lists for j in range(len(l))], dtype=torch.long), max_relations=R,
                phrase_ids=clip_processor(text=uniq, return_tensors="pt", padding=True)["input_ids"] if uniq else torch.zeros(0, 1, dtype=torch.long),
                phrase_index=pidx, boxes=boxes, masks=masks)


@torch.no_grad()
def encode_conditioning(all_models, tok, rows, phrase_encoder, device):
    """GPU half: the conditioning tensors of prompt rows ``rows`` from their token rows, on THIS rank's HIP text tower
    (all_models' HipCLIPTextEncoder for prompts / empty prompt / relation phrases; ``phrase_encoder`` for the grounding phrases'
    pooled features, interface.py:114-141).  Same dict as prepare_conditioning, on ``device``."""
    te = all_models[2]
    rows = [int(r_) for r_ in rows]
    nr, R, D = len(rows), int(tok["max_relations"]), te.hidden
    context = te.encode_ids(tok["cap_ids"][rows]) if nr else torch.zeros(0, te.max_length, D, device=device)
    uc = te.encode_ids(tok["uc_ids"]).repeat(nr, 1, 1)
    relations = torch.zeros(nr, R, D, device=device)
    pos = {r_: i for i, r_ in enumerate(rows)}
    sel = [k for k, o in enumerate(tok["rel_owner"].tolist()) if o in pos]
    if sel:
        pooled = te.encode_ids(tok["rel_ids"][sel], return_pooler_output=True)[1]
        relations[[pos[int(tok["rel_owner"][k])] for k in sel], [int(tok["rel_slot"][k]) for k in sel]] = pooled
    pidx = tok["phrase_index"][rows]
    emb = torch.zeros(nr, pidx.shape[1], D, device=device)
    used = sorted(set(int(v) for v in pidx.flatten().tolist() if v >= 0))
    if used:
        pooled = phrase_encoder.pooler_output(tok["phrase_ids"][used])
        lut = {u: i for i, u in enumerate(used)}
        ii, jj = (pidx >= 0).nonzero(as_tuple=True)
        emb[ii.to(device), jj.to(device)] = pooled[[lut[int(pidx[a, b])] for a, b in zip(ii.tolist(), jj.tolist())]]
    return dict(context=context, uc=uc, relations=relations, boxes=tok["boxes"][rows].to(device), masks=tok["masks"][rows].to(device),
                text_embeddings=emb)


def prompt_noise(seeds, latent=64):
    """Starting noise [n, 4, latent, latent], row i drawn from its OWN CPU generator seeded with ``seeds[i]``: a prompt's
    noise does not depend on the batch it is in or the rank it runs on."""
    rows = []
    for sd_ in seeds:
        g = torch.Generator(device="cpu")
        g.manual_seed(int(sd_))
        rows.append(torch.randn(1, 4, latent, latent, generator=g))
    return torch.cat(rows, 0) if rows else torch.zeros(0, 4, latent, latent)


@torch.no_grad()
def run_shard(all_models, cond, noise, device, alpha_type=(0.3, 0.0, 0.7), guidance_scale=7.5, steps=PLMS_STEPS):
    """Denoise + decode the rows of ``cond`` (a prepare_conditioning dict, or a row subset of one) as ONE batch;
    returns uint8 [n, H, W, 3] (the arithmetic of _postprocess, interface.py:543-547)."""
    model, autoencoder = all_models[0], all_models[1]
    n = noise.shape[0]
    if n == 0:
        return np.zeros((0, 0, 0, 3), dtype=np.uint8)
    model.first_conv_type = "GLIGEN"      # every sharded call starts from the checkpoint's state, whatever ran before on this rank
    dv = lambda t: t.to(device)
    batch = dict(boxes=dv(cond["boxes"]), masks=dv(cond["masks"]), text_embeddings=dv(cond["text_embeddings"]))
    lat = denoise(all_models, dv(cond["context"]), dv(cond["uc"]), dv(cond["relations"]), batch, dv(noise), list(alpha_type),
                  guidance_scale, steps=steps)
    img = autoencoder.decode(lat)
    img = torch.clamp(img, min=-1, max=1) * 0.5 + 0.5
    return (img.cpu().numpy().transpose(0, 2, 3, 1) * 255).astype(np.uint8)


@torch.no_grad()
def generate_batch_images_sharded(all_models, captions=None, labels=None, bboxes=None, clip_model=None, clip_processor=None,
                                  device=None, seeds=None, src=0, steps=PLMS_STEPS, latent=64):
    """``generate_batch_images`` (interface.py:551-570) across the ranks of a torch.distributed job.  Rank ``src`` passes the
    prompts (the others pass None) and gets the list of PIL images in prompt order; the other ranks get None.
    Per call: one object broadcast of the conditioning rows (~0.4 MB per prompt), no collective during the 51 sampling
    steps, one gather of uint8 images.  ``seeds[i]`` seeds prompt i's starting noise (default: i)."""
    import torch.distributed as dist
    from .dist import shard_indices
    from PIL import Image
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    box = [None]
    # With the HIP text tower in all_models on every rank (load_all_models_sharded ships its weights in the bundle) only the
    # host-side STRING work happens on src: the token rows travel (a few KB per prompt) and every rank encodes its own shard --
    # no rank-0 serial encoder stage.  Reference torch encoders (only present on src) keep the older flow: src encodes all rows.
    # Nothing between two collectives may raise on one rank only (the others would block forever in the next collective): every
    # failure travels as the payload of the collective that follows it and is raised on every rank after it.
    hip_flow = _is_hip_encoder(all_models[2])
    if rank == src or not multi:
        try:
            n = len(captions)
            seeds = list(range(n)) if seeds is None else [int(s_) for s_ in seeds]
            assert len(seeds) == n
            if hip_flow:
                if clip_model is not None and not _is_hip_encoder(clip_model):
                    # a reference torch CLIPModel for the grounding phrases (interface.py:114-141): never substituted silently by the
                    # checkpoint's text tower.  One process: moved onto the HIP tower once (cached on the object); several ranks: the
                    # other ranks do not hold its weights, so the caller has to pass an encoder every rank can build
                    if multi:
                        raise ValueError("clip_model is a torch CLIPModel that only rank %d holds: pass hip_phrase_encoder(clip_model, device) "
                                         "built from the same weights on EVERY rank, or None on every rank to encode the grounding phrases "
                                         "with the checkpoint's text tower" % src)
                    # cached on the object together with a fingerprint of what it was built from (target device, storage address and in-place
                    # version counter of the first parameter): a CLIPModel that is fine-tuned, reloaded or moved afterwards gets a fresh
                    # fp16 copy instead of silently encoding with the stale one
                    p0 = next(iter(clip_model.parameters()), None) if hasattr(clip_model, "parameters") else None
                    fp = (str(device), None if p0 is None else (p0.data_ptr(), int(getattr(p0, "_version", 0)), str(p0.device)))
                    entry = getattr(clip_model, "_gligen_hip_phrase_encoder", None)
                    if entry is None or entry[0] != fp:
                        entry = (fp, hip_phrase_encoder(clip_model, device))
                        try:
                            clip_model._gligen_hip_phrase_encoder = entry
                        except AttributeError:      # objects without attribute storage (__slots__): rebuilt per call
                            pass
                    clip_model = entry[1]
                box = [dict(tok=tokenize_conditioning(all_models, captions, labels, bboxes, clip_processor), seeds=seeds,
                            own_phrase_encoder=_is_hip_encoder(clip_model))]
            else:
                # prepare_conditioning returns HOST tensors: object collectives pickle tensors with their device, and rank r must not
                # receive rank 0's "cuda:0" tensors
                box = [dict(cond=prepare_conditioning(all_models, captions, labels, bboxes, clip_model, clip_processor, device), seeds=seeds)]
        except Exception as ex:          # noqa: BLE001
            if not multi:
                raise
            box = [dict(error=f"rank {rank}: {type(ex).__name__}: {ex}")]
    if multi:
        with _collective_device(device):
            dist.broadcast_object_list(box, src=src)
        if "error" in box[0]:
            raise RuntimeError("generate_batch_images_sharded failed on " + box[0]["error"])      # on every rank alike
        if "tok" in box[0] and box[0]["own_phrase_encoder"]:
            # src encodes the grounding phrases with its own HIP phrase encoder (clip_model): every rank must hold one -- agreed on
            # by ALL ranks before any of them diverges
            flags = [None] * world
            with _collective_device(device):
                dist.all_gather_object(flags, bool(_is_hip_encoder(clip_model)))
            missing = [r_ for r_, f_ in enumerate(flags) if not f_]
            if missing:
                raise ValueError("ranks %s passed no HIP phrase encoder: src encodes the grounding phrases with its own (clip_model); every "
                                 "rank must pass one built from the same CLIP weights, or all ranks pass None to use the checkpoint's text "
                                 "tower" % missing)
    seeds = box[0]["seeds"]
    n = len(seeds)
    mine = shard_indices(n, rank, world)
    # a rank whose shard fails (encoding included: sequence too long, out of memory, graph capture) still takes part in the gather,
    # with the error as its payload: the others must not block forever in gather_object, and src re-raises
    err, imgs = None, None
    try:
        if "tok" in box[0]:
            # grounding phrases: this rank's own phrase encoder when src has one (a HipCLIPTextEncoder, e.g.
            # hip_phrase_encoder(CLIPModel)), else the conditioning tower itself -- the GLIGEN checkpoint's text encoder and the
            # reference's default CLIPModel are both openai/clip-vit-large-patch14 (encoders/modules.py:146, train_rl.py:282)
            pe = clip_model if box[0]["own_phrase_encoder"] else all_models[2]
            sub = encode_conditioning(all_models, box[0]["tok"], mine, pe, device)
        else:
            sub = {k: v[mine] for k, v in box[0]["cond"].items()}
        imgs = run_shard(all_models, sub, prompt_noise([seeds[i] for i in mine], latent), device, steps=steps)
    except Exception as ex:          # noqa: BLE001
        if not multi:
            raise
        imgs, err = None, f"rank {rank}: {type(ex).__name__}: {ex}"
    if not multi:
        return [Image.fromarray(a) for a in imgs]
    parts = [None] * world if rank == src else None
    with _collective_device(device):
        dist.gather_object((mine, imgs, err), parts, dst=src)
    if rank != src:
        if err is not None:
            raise RuntimeError(err)
        return None
    errors = [e_ for _, _, e_ in parts if e_ is not None]
    if errors:
        raise RuntimeError("generate_batch_images_sharded failed on " + "; ".join(errors))
    out = [None] * n
    for idx, arr, _ in parts:
        for j, i in enumerate(idx):
            out[i] = Image.fromarray(arr[j])
    return out
