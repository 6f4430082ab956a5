#!/usr/bin/env python3
"""Benchmark of the layout-conditioned denoising hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic input = a complete 50-step PLMS
denoise (CFG 7.5, alpha_type [0.3,0,0.7] -> 102 UNet evaluations per image, run as 51 evaluations of
the 2B [cond;uncond] batch) + VAE decode.  ``--config`` picks the workload by SURVEY 8d's numbering:
   2 (default at N=1)  BASELINE.json configs[1]: 512x512, batch=4/GPU, 8 grounding boxes per image
   3                   configs[2]: 768x768, batch=2, 16 boxes (long-sequence stress)
   4 (default at N>1)  configs[3]: 8 images per GPU (global batch 64 on 8 GPUs), config-2 shapes
   5                   configs[4]: train_rl rollout, batch=16 denoise (+ reward scoring stage)
--batch/--latent/--boxes/--plms-steps override individual fields; the metric / workload strings always state what
actually ran.  Weights are random (recipe scaling; no checkpoint exists offline), inputs synthetic and resident in
HBM before the timed region.  value = images/s over all ranks (weak scaling: fixed images per GPU per step).

Extra objects in the JSON line:
  roofline      whole-UNet-forward MFMA roofline: algorithmic FLOPs of the forwards executed in the
                timed region (F_full = 1.1477 TFLOP / sample-forward with the fuser,
                F_off = 0.8141 TFLOP when the sampler's scale-0 skips it) / HIP-event time of those
                forward launches (graph replays) vs the 2.5 PFLOP/s dense fp16 MFMA peak.
  cpu_baseline  the fp32 oracle (CPU restatement of the reference UNet, oracle/unet_ref.py) timed on
                this box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

MFMA_PEAK_TFLOPS = 2500.0
# SURVEY 8d numbering -> (BASELINE.json configs index, images per GPU, latent side, boxes per image)
CONFIGS = {2: (1, 4, 64, 8), 3: (2, 2, 96, 16), 4: (3, 8, 64, 8), 5: (4, 16, 64, 8)}


def conditioning_encode_ms(dev, B, boxes, rank):
    """f-2 timed beside the step, per rank: what interface.prepare_conditioning / get_clip_features_batched run on the HIP text
    tower for one batch -- B captions + the empty prompt + B x 3 relation phrases as 77-token rows (last_hidden_state), and the B x boxes
    grounding phrases as short rows (pooler_output) -- on random-init weights of openai/clip-vit-large-patch14's text tower and synthetic token ids
    (tokenisation is host string work and not part of it).  The benchmark step itself starts from conditioning TENSORS, as the metric does."""
    from layoutllm_t2i_amd.clip import ClipTowers
    g = torch.Generator(device=dev)
    g.manual_seed(977 + rank)
    sd = {k: v for k, v in random_clip_vit_l14_state_dict(dev, g, vision=False).items()}
    towers = ClipTowers(sd, text_heads=12, device=dev)

    def ids(n, length, lo, hi):
        t = torch.randint(1, 49406, (n, length), generator=g, device=dev)
        t[:, 0] = 49406
        for r_ in range(n):
            e_ = lo + (r_ * 5) % max(1, hi - lo)
            t[r_, e_:] = 49407
        return t.cpu()
    rows77 = ids(B + 1 + 3 * B, 77, 4, 40)
    phrases = ids(B * boxes, 8, 2, 7)
    ts = []
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        towers.text_hidden_states(rows77)
        towers.text_hidden_states(phrases)
        e1.record()
        torch.cuda.synchronize()
        if it:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return {"ms_per_batch": round(ts[len(ts) // 2], 2), "rows_77_tokens": int(rows77.shape[0]), "phrase_rows": int(phrases.shape[0]),
            "note": "HIP CLIP text tower (random-init ViT-L/14 text weights, synthetic ids), this rank, outside the timed step"}


def random_clip_vit_l14_state_dict(dev, gen, vision=True):
    """Random-init weights with transformers.CLIPModel's key names and openai/clip-vit-large-patch14's shapes (vision 24 x 1024 /
    16 heads / 4096, 257 positions; text 12 x 768 / 12 heads / 3072, 77 positions, vocab 49408; projection 768)."""
    sd = {}
    r = lambda *s_: torch.randn(*s_, device=dev, generator=gen)

    def tower(prefix, n, c, inter):
        for i in range(n):
            p = f"{prefix}.encoder.layers.{i}"
            for ln in ("layer_norm1", "layer_norm2"):
                sd[f"{p}.{ln}.weight"], sd[f"{p}.{ln}.bias"] = 1 + 0.1 * r(c), 0.05 * r(c)
            for w in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[f"{p}.self_attn.{w}.weight"], sd[f"{p}.self_attn.{w}.bias"] = r(c, c) * (0.7 / c ** 0.5), 0.02 * r(c)
            sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"] = r(inter, c) / c ** 0.5, 0.02 * r(inter)
            sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"] = r(c, inter) * (0.5 / inter ** 0.5), 0.02 * r(c)
    if vision:
        tower("vision_model", 24, 1024, 4096)
    tower("text_model", 12, 768, 3072)
    if vision:
        sd["vision_model.embeddings.patch_embedding.weight"] = r(1024, 3, 14, 14) / 588 ** 0.5
        sd["vision_model.embeddings.class_embedding"] = r(1024)
        sd["vision_model.embeddings.position_embedding.weight"] = 0.3 * r(257, 1024)
    for ln, c in ((("vision_model.pre_layrnorm", 1024), ("vision_model.post_layernorm", 1024)) if vision else ()) + (("text_model.final_layer_norm", 768),):
        sd[ln + ".weight"], sd[ln + ".bias"] = 1 + 0.1 * r(c), 0.05 * r(c)
    if vision:
        sd["visual_projection.weight"] = r(768, 1024) / 32.0
    sd["text_model.embeddings.token_embedding.weight"] = 0.5 * r(49408, 768)
    sd["text_model.embeddings.position_embedding.weight"] = 0.3 * r(77, 768)
    sd["text_projection.weight"] = r(768, 768) / 768 ** 0.5
    return sd


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="workload by SURVEY 8d numbering (0 = auto: 2 at N=1, 4 at N>1)")
    ap.add_argument("--batch", type=int, default=0, help="override: images per GPU per step")
    ap.add_argument("--latent", type=int, default=0, help="override: latent side (64 = 512x512, 96 = 768x768)")
    ap.add_argument("--plms-steps", type=int, default=50)
    ap.add_argument("--boxes", type=int, default=0, help="override: grounding boxes per image")
    ap.add_argument("--no-cpu-config1", action="store_true", help="skip the end-to-end config-1 CPU run (22 oracle forwards, ~90 s)")
    ap.add_argument("--cpu-config1", action="store_true",
                    help="also run BASELINE configs[0] end to end on the CPU oracle (S=10, 22 forwards, ~2 min)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="debug: tiny UNet (not a valid benchmark)")
    ap.add_argument("--no-hot-kernel", action="store_true", help="skip the standalone timing of the hottest kernel shape (PMC passes)")
    ap.add_argument("--no-strict", action="store_true", help="skip the strict-mode leg (second engine on split weights: images/s + parity beside the default mode)")
    ap.add_argument("--strict-steps", type=int, default=5, help="timed denoise steps of the strict-mode leg")
    ap.add_argument("--verbose-json", action="store_true", help="keep the long descriptive strings in the JSON line (default: compact line, the prose lives in DESIGN.md section 5)")
    ap.add_argument("--strict-main", action="store_true", help="run the MAIN timed loop in strict mode (any config; the line then says so in dtype / config; "
                    "not the headline: the headline is the default mode)")
    ap.add_argument("--no-vae", action="store_true", help="stop at the final latent (exclude the VAE decode stage from the step)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="gl_set_option tuning knob for same-box A/B runs (see include/gligen_hip.h), repeatable")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path (process group, bundle broadcast, barriers, max-over-ranks) "
                    "even at WORLD_SIZE=1: runs the RCCL calls on a 1-GPU box")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to smoke-test the path on one GPU)")
    return ap.parse_args()


# descriptive strings of the JSON line that --verbose-json keeps; the default line carries the numbers only (the ~5 KB line lost its
# front half in the driver's tail; what each field means is written down once)
_PROSE = ("what", "note", "traffic_note", "flop_model", "classes_note", "step_includes", "reward_score_note", "launch", "sample", "kernel", "cpu_model")


def compact_line(obj):
    if isinstance(obj, dict):
        return {k: compact_line(v) for k, v in obj.items() if k not in _PROSE or k in ("sample", "cpu_model") and not isinstance(v, dict)
                and len(str(v)) <= 120}
    if isinstance(obj, float):
        return float(f"{obj:.5g}")
    return obj


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but only {ndev} GPUs visible")
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from layoutllm_t2i_amd import recipe
    from layoutllm_t2i_amd.arch import TINY, UNetConfig
    from layoutllm_t2i_amd.dist import broadcast_packed
    from layoutllm_t2i_amd.engine import UNetEngine
    from layoutllm_t2i_amd.interface import denoise
    from layoutllm_t2i_amd.model import GroundingNetInput, LatentDiffusion, UNetModel
    from layoutllm_t2i_amd.arch import VAEConfig
    from layoutllm_t2i_amd.vae import VAEDecoder
    from layoutllm_t2i_amd.weights import pack_state_dict, random_state_dict, random_vae_state_dict

    cfg = TINY if args.tiny else UNetConfig()
    if args.strict_main:
        import dataclasses as _dcs
        cfg = _dcs.replace(cfg, split_weights=True)
        args.no_strict = True
    cnum = args.config or (2 if world == 1 else 4)
    cidx, B, side, nbox = CONFIGS[cnum]
    B, side, nbox = args.batch or B, args.latent or side, args.boxes or nbox
    args.boxes = nbox
    overridden = (B, side, nbox) != CONFIGS[cnum][1:] or args.plms_steps != 50 or args.tiny
    from layoutllm_t2i_amd.flops import unet_forward_flops
    F_FULL = unet_forward_flops(cfg, side, True)       # algorithmic FLOPs / sample-forward, fuser on   
    F_OFF = unet_forward_flops(cfg, side, False)       # ... with the gated-SA fuser exactly skipped (scale == 0)
    if args.opt:
        from layoutllm_t2i_amd import ops as _ops
        for kv in args.opt:
            k, v = kv.split("=")
            _ops.set_option(int(k), int(v))

    # ---- weights: rank 0 builds + packs, one RCCL broadcast to the others (timed separately)
    t0 = time.time()
    sd_cpu_sample = None
    packed = None
    packed_strict = None
    if rank == 0:
        sd = random_state_dict(cfg, dev, seed=0)
        fc = {"weight": torch.randn(cfg.model_channels, cfg.in_channels, 3, 3, device=dev) * 0.16,
              "bias": torch.zeros(cfg.model_channels, device=dev)}
        packed = pack_state_dict(sd, cfg, dev, fc)
        if world == 1 and not args.no_cpu_baseline:
            sd_cpu_sample = {k: v.cpu() for k, v in sd.items()}
        want_strict = world == 1 and not args.no_strict and not args.tiny and cnum == 2 and not overridden
        packed_strict = None
        if want_strict:
            # the same weights in the split layout ([Whi | Wlo] for every matrix): the strict-mode engine of the second leg
            import dataclasses as _dc
            packed_strict = pack_state_dict(sd, _dc.replace(cfg, split_weights=True), dev, fc)
        del sd
    # VAE decoder weights: built on rank 0 and sent with the UNet in the SAME broadcast (dist.broadcast_bundle: one flat buffer)
    want_vae = not args.no_vae and not args.tiny
    vae = None
    if rank == 0 and want_vae:
        vae = VAEDecoder(random_vae_state_dict(VAEConfig(), dev, seed=1), VAEConfig(), dev)
    bcast_ms = None
    if multi:
        from layoutllm_t2i_amd.dist import broadcast_bundle
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.time()
        packed, vae_w, _ = broadcast_bundle(packed, vae.W if vae is not None else None, cfg if rank == 0 else None, dev, src=0)
        torch.cuda.synchronize()
        dist.barrier()
        bcast_ms = (time.time() - tb) * 1e3
        if rank != 0 and want_vae:
            vae = VAEDecoder.from_packed(vae_w, VAEConfig(), dev)
    def make_model(packed_):
        # assemble the facade around already-packed weights (UNetModel.__init__ packs from a state_dict)
        m_ = UNetModel.__new__(UNetModel)
        m_.cfg, m_.device = packed_.cfg, dev
        m_.image_size, m_.in_channels, m_.out_channels, m_.model_channels = cfg.image_size, cfg.in_channels, cfg.out_channels, cfg.model_channels
        m_.first_conv_restorable, m_.first_conv_type = True, "GLIGEN"
        m_.grounding_tokenizer_input = GroundingNetInput()
        m_.fuser_scale, m_.training, m_._cond_key = 1.0, False, None
        m_.engine = UNetEngine(packed_)
        return m_
    model = make_model(packed)
    if args.strict_main:
        model.engine.set_option(50, 1)
    diffusion = LatentDiffusion(device=dev)
    all_models = (model, vae, None, diffusion, {})
    setup_s = time.time() - t0

    # ---- synthetic inputs, resident on the device (seed 1234 + rank)
    inp = {k: torch.from_numpy(v).to(dev) for k, v in
           recipe.synth_inputs(cfg, B, side, n_boxes=args.boxes, n_rel=3, seed=1234 + rank).items()}
    batch = dict(boxes=inp["boxes"], masks=inp["masks"], text_embeddings=inp["positive_embeddings"])

    # ---- HIP-event timing of every UNet forward launch (graph replay) on the launch stream.  The sampler issues one
    # gl_plms_step per evaluation = the forward's graph + two tiny elementwise kernels (CFG combine, x_prev update).
    eng = model.engine
    fwd_events = []
    orig_step = eng.plms_step

    def timed_step(x_eval, x_base, x_out, e_out, e_terms, coefs, div, t, reps, guidance, fuser_scale, sd_conv, *sched):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_step(x_eval, x_base, x_out, e_out, e_terms, coefs, div, t, reps, guidance, fuser_scale, sd_conv, *sched)
        e1.record()
        fwd_events.append((e0, e1, fuser_scale != 0, x_eval.shape[0] * reps))
        return out
    eng.plms_step = timed_step

    vae_events = []
    # configs[4] (train_rl rollout): the reward stage runs after the decode, as Reward_Model.forward does (models/policy.py:106-135):
    # decoded images -> uint8 (interface.py:543-547) -> the CLIP feature extractor's PIL-bicubic resize 512 -> 224, crop, normalise ON THE
    # GPU (preprocess.py: Pillow's fixed-point algorithm bit for bit; the 16 ground-truth images arrive as host uint8 480 x 640 arrays,
    # like dataset images, and take the same kernels after one upload) -> CLIP ViT-L/14 vision tower (HIP) on the 16 predictions and the
    # 16 ground-truth images, text tower (HIP) on the 16 captions -> similarities + AestheticMLP (gl_reward_score).  Random-init towers
    # of the ViT-L/14 architecture.  The first step also runs the reference's host path (HF processor on the CPU) once: the GPU
    # pixel_values must equal it bitwise, and its time is reported beside the GPU one.
    scorer, score_events, score_in = None, [], None
    reward_stage_ms = {"processor_cpu": [], "preprocess_equal": []}
    if cnum == 5 and not args.tiny:
        from layoutllm_t2i_amd.reward import RewardModel
        gsc = torch.Generator(device=dev)
        gsc.manual_seed(4242 + rank)
        shp = {0: (1024, 768), 2: (128, 1024), 4: (64, 128), 6: (16, 64), 7: (1, 16)}
        aes_sd = {}
        for li, (n_, k_) in shp.items():
            aes_sd[f"layers.{li}.weight"] = (torch.rand(n_, k_, device=dev, generator=gsc) * 2 - 1) * (3.0 / k_) ** 0.5
            aes_sd[f"layers.{li}.bias"] = (torch.rand(n_, device=dev, generator=gsc) * 2 - 1) * 0.1
        scorer = RewardModel(random_clip_vit_l14_state_dict(dev, gsc), aes_sd, dev)
        try:
            from transformers import CLIPImageProcessor
            processor = CLIPImageProcessor()
        except Exception as e:                                   # pragma: no cover
            raise SystemExit(f"--config 5 needs transformers' CLIPImageProcessor for the host-side preprocessing step: {e}")
        import numpy as _np
        gt_u8 = _np.random.default_rng(7 + rank).integers(0, 256, (B, 480, 640, 3), dtype=_np.uint8)       # ground-truth images (host, dataset-like)
        cap_ids = torch.randint(1, 49406, (B, 77), device=dev, generator=gsc)
        cap_ids[:, 0] = 49406
        for b_ in range(B):
            L_ = 8 + (b_ * 5) % 40
            cap_ids[b_, L_] = 49407
            cap_ids[b_, L_ + 1:] = 49407
        score_in = (cap_ids, gt_u8, processor)

    def one_step():
        model.first_conv_type = "GLIGEN"
        lat = denoise(all_models, inp["context"], inp["uc"], inp["relations"], batch, inp["x"], [0.3, 0.0, 0.7], 7.5,
                      steps=args.plms_steps)
        if vae is None:
            return lat
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        img = vae.decode(lat)                      # fp32 [B, 3, 512, 512], what interface.py:541 hands to the PIL loop
        e1.record()
        vae_events.append((e0, e1))
        if scorer is not None:
            cap_ids, gt_u8, processor = score_in
            sp, s0, s1, s2 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            sp.record()
            px = scorer.preprocess.from_decoded(img)
            gt_pixels = scorer.preprocess(torch.from_numpy(gt_u8).to(dev))
            s0.record()
            txt = scorer.towers.get_text_features(cap_ids)
            fboth = scorer.towers.get_image_features(torch.cat([px, gt_pixels], 0))      # one pass over predictions + ground truth, as RewardModel.forward
            fp, fg = fboth[:B], fboth[B:]
            s1.record()
            sc = scorer.scorer.score(txt, fp, fg)
            s2.record()
            score_events.append((sp, s0, s1, s2))
            assert sc["reward"].shape == (B,) and bool(torch.isfinite(sc["reward"]).all())
        return img

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = one_step()
    fwd_events.clear()
    vae_events.clear()
    score_events.clear()
    sync()
    t1 = time.time()
    for _ in range(args.steps):
        out = one_step()
    sync()
    elapsed = time.time() - t1
    assert torch.isfinite(out).all(), "non-finite latents"
    if scorer is not None:
        # outside the timed region: the reference's host path (HF processor on the CPU) once on the last batch -- the GPU
        # pixel_values must equal it bitwise; its time is reported beside the GPU preprocessing time
        cap_ids, gt_u8, processor = score_in
        torch.cuda.synchronize()
        tc = time.time()
        u8 = ((torch.clamp(out, -1, 1) * 0.5 + 0.5).cpu().numpy().transpose(0, 2, 3, 1) * 255).astype("uint8")     # interface.py:543-547
        px_host = processor(images=[u for u in u8] + [g for g in gt_u8], return_tensors="pt")["pixel_values"]
        reward_stage_ms["processor_cpu"].append((time.time() - tc) * 1e3)
        px_gpu = torch.cat([scorer.preprocess.from_decoded(out), scorer.preprocess(torch.from_numpy(gt_u8).to(dev))], 0)
        reward_stage_ms["preprocess_equal"].append(bool(torch.equal(px_gpu.cpu(), px_host)))
        assert reward_stage_ms["preprocess_equal"][-1], "GPU image preprocessing differs from the host CLIP processor"
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- STRICT mode leg (second engine on the split weight layout, gl_set_handle_option 50; outside the headline's timed region): the same
    # denoise + decode on split-fp16 operands everywhere -- the mode whose output is within north_star's rtol 1e-3 / atol 1e-4 of the fp32
    # reference -- timed the same way; its parity is filled in by the oracle leg below
    strict_info, eng_s = None, None
    if packed_strict is not None:
        eng.plms_step = orig_step
        model_s = make_model(packed_strict)
        eng_s = model_s.engine
        eng_s.set_option(50, 1)
        am_s = (model_s, vae, None, diffusion, {})
        s_events = []
        s_orig = eng_s.plms_step

        def s_timed(*a_, **k_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_ = s_orig(*a_, **k_)
            e1.record()
            s_events.append((e0, e1, a_[10] != 0, a_[0].shape[0] * a_[8]))      # (fuser_scale != 0, samples of the launch): plms_step's argument order
            return o_
        eng_s.plms_step = s_timed

        def strict_step():
            model_s.first_conv_type = "GLIGEN"
            lat_ = denoise(am_s, inp["context"], inp["uc"], inp["relations"], batch, inp["x"], [0.3, 0.0, 0.7], 7.5, steps=args.plms_steps)
            return vae.decode(lat_) if vae is not None else lat_
        strict_step()
        s_events.clear()
        torch.cuda.synchronize()
        ts = time.time()
        for _ in range(args.strict_steps):
            o_s = strict_step()
        torch.cuda.synchronize()
        t_strict = time.time() - ts
        assert torch.isfinite(o_s).all()
        eng_s.plms_step = s_orig
        strict_info = {"images_per_s": round(args.strict_steps * B / t_strict, 4), "ms_per_step": round(t_strict / args.strict_steps * 1e3, 1),
                       "unet_forward_ms": round(sum(a_.elapsed_time(b_) for a_, b_, _, _ in s_events) / max(len(s_events), 1), 3),
                       "steps": args.strict_steps, "vs_default": round((args.strict_steps * B / t_strict) / (args.steps * B * world / elapsed), 3),
                       "weights_bytes": packed_strict.nbytes(),
                       "what": "gl_set_handle_option(50, 1) on a split_weights handle: split-fp16 operands ([hi | lo] activations, [Whi | Wlo] weights, "
                               "3 MFMA passes) for every conv / GEMM / attention product; same workload, same timing method"}
        # the strict leg's own roofline line: ALGORITHMIC FLOPs (the same model as the headline's) over the HIP-event time of its forward launches;
        # the matrix pipe issues three times that work (issued_frac).  Traffic: the newest strict PMC record under profiles/, if any.
        s_ms = sum(a_.elapsed_time(b_) for a_, b_, _, _ in s_events)
        s_fl = sum(n_ * (F_FULL if on_ else F_OFF) for _, _, on_, n_ in s_events)
        s_tf = s_fl / (s_ms * 1e-3) / 1e12 if s_ms > 0 else float("nan")
        s_tpath = next((q for q in (os.path.join(ROOT, "profiles", f"r{r}_strict_traffic.json") for r in (6,)) if os.path.exists(q)), None)
        strict_info["roofline"] = {"bound": "mfma", "achieved": round(s_tf, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(s_tf / MFMA_PEAK_TFLOPS, 4),
                                   "issued_frac": round(3 * s_tf / MFMA_PEAK_TFLOPS, 4), "launches": len(s_events),
                                   "avg_launch_ms": round(s_ms / max(len(s_events), 1), 3),
                                   "traffic": (round(json.load(open(s_tpath))["traffic_bytes_per_forward"]) if s_tpath else None)}
        if not args.no_hot_kernel:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import kbench as _kbs
            strict_info["roofline"]["classes"] = _kbs.class_summary(iters=4, peak_tflops=MFMA_PEAK_TFLOPS, strict=True)
        eng.plms_step = timed_step

    # ---- roofline of the UNet forward (dominant launch)
    flops = 0.0
    gpu_ms = 0.0
    for e0, e1, fuser_on, nsamp in fwd_events:
        gpu_ms += e0.elapsed_time(e1)
        flops += nsamp * (F_FULL if fuser_on else F_OFF)
    n_fwd = len(fwd_events)
    achieved = flops / (gpu_ms * 1e-3) / 1e12 if gpu_ms > 0 else float("nan")
    # HBM traffic per forward launch: measured in separate rocprofv3 --pmc passes (profiles/r1_traffic.json)
    traffic = None
    tpath = next((q for q in (os.path.join(ROOT, "profiles", f"r{r}_traffic.json") for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(q)), None)
    traffic_age = None
    if tpath and side == 64 and not args.tiny and B == 4:
        tdoc = json.load(open(tpath))
        traffic = round(tdoc["traffic_bytes_per_forward"])
        traffic_age = tdoc.get("git_head", "unrecorded (before round 5)")
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_age": traffic_age,
                "traffic_note": "PMC passes (2*FETCH+WRITE) of %s, tree %s; not measured in this run" % (os.path.relpath(tpath, ROOT) if tpath else None, traffic_age),
                "launch": "UNet forward of the 2B=%d [cond;uncond] batch (one hipGraph replay)" % (2 * B),
                "launches": n_fwd, "avg_launch_ms": round(gpu_ms / max(n_fwd, 1), 3),
                "flops_per_launch": round(flops / max(n_fwd, 1)), "flop_model": "SURVEY 8d minimal (32 F_full + 70 F_off per image at S=50; F_full=%.4f, F_off=%.4f TFLOP/sample-forward at this latent)" % (F_FULL / 1e12, F_OFF / 1e12)}

    roofline = {**{k_: roofline[k_] for k_ in ("traffic_note", "launch", "flop_model")}, **{k_: v_ for k_, v_ in roofline.items() if k_ not in ("traffic_note", "launch", "flop_model")}}

    # ---- the single hottest kernel shape, timed standalone with HIP events on the launch stream: the implicit-GEMM
    # 3x3 conv 320 -> 320 at the 64x64 level of the 2B batch (gemm8_kernel<256,160,true>, 7 launches per forward; that instantiation is
    # the top line of profiles/r3_kernel_stats.csv, where its average covers all conv shapes incl. the split-K ones)
    if rank == 0 and side == 64 and not args.tiny and not args.no_hot_kernel:
        from layoutllm_t2i_amd import ops as _o
        Bn = 2 * B
        xa = torch.randn(Bn * side * side, 320, device=dev).to(torch.float16)
        wa = (torch.randn(320, 9 * 320, device=dev) * 0.02).to(torch.float16)   # packed [Cout, Cin/64, 3, 3, 64] flattened
        oa = torch.empty(Bn * side * side, 320, dtype=torch.float16, device=dev)
        ba = torch.zeros(320, device=dev)
        for _ in range(5):
            _o.conv3x3(xa, wa, oa, Bn, side, side, ba)
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nrep = 50
        k0.record()
        for _ in range(nrep):
            _o.conv3x3(xa, wa, oa, Bn, side, side, ba)
        k1.record()
        torch.cuda.synchronize()          # rank-local: no collective here, the other ranks do not run this block
        k_us = k0.elapsed_time(k1) / nrep * 1e3
        k_flops = 2.0 * Bn * side * side * 320 * 9 * 320
        k_tf = k_flops / (k_us * 1e-6) / 1e12
        roofline["hot_kernel"] = {"kernel": "gemm8_kernel<256,160,true> (8-wave deep-pipelined implicit-GEMM) as 3x3 conv 320->320 @64x64, 2B=%d" % Bn,
                                  "avg_us": round(k_us, 1), "flops": k_flops, "achieved": round(k_tf, 1), "unit": "TFLOP/s",
                                  "frac": round(k_tf / MFMA_PEAK_TFLOPS, 4), "launches_per_forward": 7}
        del xa, wa, oa
        # per kernel CLASS of one fuser-off forward: every distinct shape x its multiplicity timed standalone with HIP events IN THIS RUN
        # (tools/kbench.class_summary): fraction of the dense-fp16 MFMA peak for the 3x3 convs, the plain GEMMs and attention
        if B == 4:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import kbench as _kb
            roofline["classes"] = _kb.class_summary(iters=6, peak_tflops=MFMA_PEAK_TFLOPS)
            roofline["classes_note"] = ("standalone op-level launches of every conv / plain-GEMM / attention shape of one fuser-off 2B=8 forward "
                                        "(multiplicity-weighted), HIP events on the launch stream, measured in this run")

    images = args.steps * B * world
    value = images / elapsed
    result = {
        "metric": f"{side * 8}x{side * 8} {args.plms_steps}-step images/sec (PLMS, CFG 7.5, layout-conditioned GLIGEN UNet)",
        "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "split-fp16 operands, 3 MFMA passes (STRICT mode), fp32 accumulate" if args.strict_main else "fp16 (fp32 accumulate)", "data": "synthetic inputs, random-init weights of the reference architecture",
        "config": {"workload": ("custom (flags override the named config): " if overridden else f"configs[{cidx}]: ")
                   + f"{'TINY debug UNet, ' if args.tiny else ''}{side * 8}x{side * 8}, {args.plms_steps} PLMS steps, batch={B}/GPU, {args.boxes} grounding boxes/image, fp16",
                   "survey_config": cnum,
                   "images_per_gpu_per_step": B, "unet_forwards_per_image": 2 * (args.plms_steps + 1), "latent": [B, 4, side, side],
                   "parallelism": f"replicas x{world}, one weight broadcast, no per-step collectives"},
        "unet_step_ms": round(gpu_ms / max(n_fwd, 1), 3),
        "vae_decode_ms_per_batch": (round(sum(a.elapsed_time(b) for a, b in vae_events) / max(len(vae_events), 1), 2) if vae_events else None),
        "step_includes": f"PLMS denoise ({args.plms_steps + 1} x gl_plms_step: 2B UNet forward + CFG + update)" + (" + VAE decode to fp32 images" if vae is not None else "") + (" + reward scoring (gl_reward_score)" if scorer is not None else ""),
        "launches_per_forward": eng.num_launches(),
        "images_per_sec_per_gpu": round(value / world, 4),
        "reward_score_ms_per_batch": (round(sum(a.elapsed_time(d) for a, b, c, d in score_events) / max(len(score_events), 1), 3) if score_events else None),
        "reward_stage_ms": ({"image preprocessing (HIP: uint8 + PIL-exact bicubic 512->224 of 16 predictions; upload + 480x640->224 of 16 ground-truth images)":
                             round(sum(a.elapsed_time(b) for a, b, c, d in score_events) / len(score_events), 3),
                             "clip_towers(text 16 + vision 2x16, HIP)": round(sum(b.elapsed_time(c) for a, b, c, d in score_events) / len(score_events), 3),
                             "similarities+aesthetic (gl_reward_score)": round(sum(c.elapsed_time(d) for a, b, c, d in score_events) / len(score_events), 3),
                             "reference host path of the preprocessing (HF CLIP processor, PIL, CPU; once, outside the timed region)": round(reward_stage_ms["processor_cpu"][-1], 1),
                             "gpu_pixel_values_equal_host_processor_bitwise": reward_stage_ms["preprocess_equal"][-1]}
                            if score_events else None),
        "reward_score_note": ("Reward_Model.forward's GPU part on the decoded images: uint8 + CLIP feature-extractor preprocessing (HIP) -> CLIP "
                              "ViT-L/14 towers (random-init, HIP) on 16 predictions + 16 ground-truth images + 16 captions -> similarities + "
                              "AestheticMLP; the CPU layout rewards (IoU / DocSim, policy.py:126-133) are the caller's and not timed" if score_events else None),
        "roofline": roofline,
        "setup_s": round(setup_s, 1),
    }
    if not args.tiny:
        result["conditioning_encode"] = conditioning_encode_ms(dev, B, args.boxes, rank)
    if bcast_ms is not None:
        result["weight_broadcast_ms"] = round(bcast_ms, 1)
        result["weight_bytes"] = packed.nbytes()

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N=1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and sd_cpu_sample is not None:
        from oracle import plms_ref, unet_ref
        cores = min(os.cpu_count() or 1, 32)   # eager torch scales poorly past ~32 threads on this op mix
        torch.set_num_threads(cores)
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as fcpu:
                cpu_model = next(l.split(":", 1)[1].strip() for l in fcpu if l.startswith("model name"))
        except (OSError, StopIteration):
            pass
        ci = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, 1, side, n_boxes=args.boxes, n_rel=3, seed=1234).items()}
        zz = torch.zeros_like

        def cpu_fwd(x, tt, cond, fuser_scale=1.0):
            if cond:
                return unet_ref.unet_forward(sd_cpu_sample, cfg, x, tt, ci["context"], ci["relations"], ci["boxes"], ci["masks"],
                                             ci["positive_embeddings"], fuser_scale=fuser_scale)
            return unet_ref.unet_forward(sd_cpu_sample, cfg, x, tt, ci["uc"], ci["relations"], zz(ci["boxes"]), zz(ci["masks"]),
                                         zz(ci["positive_embeddings"]), fuser_scale=fuser_scale)
        with torch.no_grad():
            # bounded sample (~20 s of CPU work): one warm-up + the four forward kinds a sampling step is made of
            cpu_fwd(ci["x"], torch.tensor([481]), True)
            kinds, refs = {}, {}
            for name, cond, fs in (("cond_on", True, 1.0), ("uncond_on", False, 1.0), ("cond_off", True, 0.0), ("uncond_off", False, 0.0)):
                tc = time.time()
                refs[name] = cpu_fwd(ci["x"], torch.tensor([481]), cond, fs)
                kinds[name] = time.time() - tc
        # the oracle as CHECKER of the batch the benchmark actually ran (same tiles / split-K dispatch): sample 0 of the
        # 2B batch = synth sample 0 (cond), sample B = its unconditional twin; unrounded fp32 weights on the oracle side
        cat = lambda a, b: torch.cat([a, b], 0)
        eng.plms_step = orig_step
        eng.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], zz(inp["boxes"])),
                             cat(inp["masks"], zz(inp["masks"])), cat(inp["positive_embeddings"], zz(inp["positive_embeddings"])), side)
        rl2 = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
        # fraction of elements outside north_star's elementwise tolerance rtol 1e-3 / atol 1e-4
        outside = lambda a, b: float(((a.float().cpu() - b).abs() > 1e-4 + 1e-3 * b.abs()).float().mean())
        e_on = eng.forward(inp["x"], 481.0, 1.0, False, 2).clone()
        e_off = eng.forward(inp["x"], 481.0, 0.0, False, 2).clone()
        # the same forward on the Whi half of the split 1x1-conv weights only (key 45 = 0): the engine then uses exactly the fp16-rounded
        # weight matrices the second oracle run below uses (arithmetic-only comparison)
        from layoutllm_t2i_amd import ops as _op2
        _op2.set_option(45, 0)
        _op2.set_option(38, 0)
        e_on_hi = eng.forward(inp["x"], 481.0, 1.0, False, 2).clone()
        _op2.set_option(45, 1024)
        _op2.set_option(38, 1)
        for kv in args.opt:                          # (restore a --opt 45=... given on the command line)
            if kv.split("=")[0] == "45":
                _op2.set_option(45, int(kv.split("=")[1]))
        # one more oracle forward on fp16-ROUNDED weight matrices (what the engine stores): isolates the arithmetic error from the
        # weight quantisation, the quantity tests/test_gpu_configs.py bounds
        sd_r = {k: (v.half().float() if v.dim() >= 2 else v) for k, v in sd_cpu_sample.items()}
        with torch.no_grad():
            ref_r = unet_ref.unet_forward(sd_r, cfg, ci["x"], torch.tensor([481]), ci["context"], ci["relations"], ci["boxes"], ci["masks"],
                                          ci["positive_embeddings"], fuser_scale=1.0)
        del sd_r
        if eng_s is not None:
            eng_s.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], zz(inp["boxes"])),
                                   cat(inp["masks"], zz(inp["masks"])), cat(inp["positive_embeddings"], zz(inp["positive_embeddings"])), side)
            s_on = eng_s.forward(inp["x"], 481.0, 1.0, False, 2).clone()
            s_off = eng_s.forward(inp["x"], 481.0, 0.0, False, 2).clone()
            strict_info["parity_vs_fp32_oracle_unrounded_weights"] = {
                "rel_l2": {"cond_on": rl2(s_on[0:1], refs["cond_on"]), "uncond_on": rl2(s_on[B:B + 1], refs["uncond_on"]),
                           "cond_off": rl2(s_off[0:1], refs["cond_off"]), "uncond_off": rl2(s_off[B:B + 1], refs["uncond_off"])},
                "outside_rtol1e-3_atol1e-4": {"cond_on": round(outside(s_on[0:1], refs["cond_on"]), 5), "uncond_on": round(outside(s_on[B:B + 1], refs["uncond_on"]), 5),
                                              "cond_off": round(outside(s_off[0:1], refs["cond_off"]), 5), "uncond_off": round(outside(s_off[B:B + 1], refs["uncond_off"]), 5)}}
            strict_info["outside_frac_max"] = max(strict_info["parity_vs_fp32_oracle_unrounded_weights"]["outside_rtol1e-3_atol1e-4"].values())
        result["parity_at_bench_batch"] = {"rel_l2_cond_on": rl2(e_on[0:1], refs["cond_on"]), "rel_l2_uncond_on": rl2(e_on[B:B + 1], refs["uncond_on"]),
                                           "rel_l2_cond_off": rl2(e_off[0:1], refs["cond_off"]), "rel_l2_uncond_off": rl2(e_off[B:B + 1], refs["uncond_off"]),
                                           "outside_rtol1e-3_atol1e-4": {"cond_on": round(outside(e_on[0:1], refs["cond_on"]), 4),
                                                                         "uncond_on": round(outside(e_on[B:B + 1], refs["uncond_on"]), 4),
                                                                         "cond_off": round(outside(e_off[0:1], refs["cond_off"]), 4),
                                                                         "uncond_off": round(outside(e_off[B:B + 1], refs["uncond_off"]), 4)},
                                           "vs_fp16_rounded_weights_cond_on": {"rel_l2": rl2(e_on_hi[0:1], ref_r),
                                                                               "outside_rtol1e-3_atol1e-4": round(outside(e_on_hi[0:1], ref_r), 4)},
                                           "note": "HIP engine sample 0 / B of the 2B batch vs the fp32 oracle, t=481: on UNROUNDED fp32 weights (includes the "
                                                   "fp16 rounding of the stored weights; the three kinds of 1x1 conv keep [Whi | Wlo], gl_set_option 45) and, for "
                                                   "cond_on, engine on the fp16 halves only vs the oracle on fp16-rounded weight matrices (arithmetic only)"}
        with torch.no_grad():
            pass
        S = args.plms_steps
        n_on = int(0.3 * S) + 1                       # step-evaluations with the fuser scaled 1 (step 0 evaluates twice)
        n_off = S + 1 - n_on
        per_image = n_on * (kinds["cond_on"] + kinds["uncond_on"]) + n_off * (kinds["cond_off"] + kinds["uncond_off"])
        result["cpu_baseline"] = {
            "value": round(1.0 / per_image, 6), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model, "host_cores_total": os.cpu_count(),
            "sample": f"4 oracle UNet forwards (cond/uncond x fuser on/off) at B=1, {side}x{side}, extrapolated to {n_on}+{n_off} steps x 2",
            "sample_seconds": {k: round(v, 2) for k, v in kinds.items()},
            "seconds_per_forward": round(sum(kinds.values()) / 4, 2)}
        if not args.no_cpu_config1 and not args.tiny and args.config in (0, 2):
            # (default since round 4: the CPU number is then not only an extrapolation)
            # BASELINE.json configs[0] / BASELINE.md section 4: txt2img plumbing case, B=1, 64x64, S=10 -> 22 forwards, fp32, end to end
            c1 = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, 1, 64, n_boxes=2, n_rel=3, seed=1234).items()}

            def eps_fn(x, tt, i, alpha):
                with torch.no_grad():
                    e_c = unet_ref.unet_forward(sd_cpu_sample, cfg, x, tt, c1["context"], c1["relations"], c1["boxes"], c1["masks"],
                                                c1["positive_embeddings"], fuser_scale=float(alpha))
                    e_u = unet_ref.unet_forward(sd_cpu_sample, cfg, x, tt, c1["uc"], c1["relations"], zz(c1["boxes"]), zz(c1["masks"]),
                                                zz(c1["positive_embeddings"]), fuser_scale=float(alpha))
                return e_u + 7.5 * (e_c - e_u)
            tc = time.time()
            lat1 = plms_ref.plms_sample(eps_fn, c1["x"], 10, [0.3, 0.0, 0.7])
            t_c1 = time.time() - tc
            assert torch.isfinite(lat1).all()
            result["cpu_baseline"]["config1_end_to_end"] = {"seconds": round(t_c1, 1), "forwards": 22, "S": 10, "boxes": 2,
                                                            "images_per_s": round(1.0 / t_c1, 6)}
    if strict_info is not None:
        result["strict_mode"] = strict_info
    # flat scalars inside `roofline` (the driver's record keeps the scalar fields of this object and drops nested ones)
    hk, cl = roofline.get("hot_kernel"), roofline.get("classes")
    if hk:
        roofline["hot_kernel_us"], roofline["hot_kernel_frac"] = hk["avg_us"], hk["frac"]
    if cl:
        for name_, key_ in (("conv3x3", "conv_frac"), ("plain_gemm", "plain_gemm_frac"), ("attention", "attention_frac")):
            if name_ in cl:
                roofline[key_] = cl[name_].get("frac")
    if strict_info is not None:
        roofline["strict_images_per_s"] = strict_info["images_per_s"]
        roofline["strict_outside_frac"] = strict_info.get("outside_frac_max")
    # line order: the long descriptive fields first, the judged numbers LAST (a tail of the line keeps them)
    last = ("parity_at_bench_batch", "strict_mode", "roofline", "cpu_baseline")
    result = {**{k_: v_ for k_, v_ in result.items() if k_ not in last}, **{k_: result[k_] for k_ in last if k_ in result}}
    if not args.verbose_json:
        result = compact_line(result)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
