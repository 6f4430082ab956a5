"""Parity at the sizes BASELINE.json's configs actually run (the tile / split-K / 256-row dispatch depends on the
batch): the 2B = 8 batch of configs[1], the 2B = 32 batch of configs[4], configs[2] at 96x96 latents with B = 2 and
16 boxes.  Samples of a batch are independent (GroupNorm is per sample, LayerNorm per token), so the oracle is
evaluated for ONE sample of each batch on the host CPU and compared with the engine's row for that sample.

Bounds are <= 1.1x the values measured on MI355X (printed by ``report``).  north_star's elementwise rtol 1e-3 / atol 1e-4 is NOT
met by every element: 27-30 % lie outside it (rel-L2 6.7e-4 .. 7.5e-4).  With fp16 operands for the 3x3 convs and the FF / attention
projections the floor is rel-L2 6.5e-4 / 26 % outside (tools/precision_sim.py); the engine sits at it.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(__file__))
from layoutllm_t2i_amd import ops, recipe
from layoutllm_t2i_amd.arch import UNetConfig
from layoutllm_t2i_amd.interface import denoise
from layoutllm_t2i_amd.model import GroundingNetInput, LatentDiffusion, UNetModel
from layoutllm_t2i_amd.weights import random_state_dict
from oracle import unet_ref

DEV = "cuda:0"
T = torch.from_numpy


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def report(name, out, ref, frac_bound=None):
    """rel-L2, and the fraction of elements outside north_star's elementwise rtol 1e-3 / atol 1e-4; ``frac_bound`` asserts that
    fraction (<= 1.1x the measured one) instead of only printing it."""
    out, ref = out.float().cpu(), ref.float().cpu()
    r = rel_l2(out, ref)
    d = (out - ref).abs()
    frac = float((d > 1e-4 + 1e-3 * ref.abs()).float().mean())
    print(f"[{name}] rel_l2={r:.3e} max|err|={float(d.max()):.3e} |ref|max={float(ref.abs().max()):.3f} "
          f"outside rtol1e-3/atol1e-4: {100 * frac:.1f}%")
    assert torch.isfinite(out).all(), name
    if frac_bound is not None:
        assert frac <= frac_bound, f"{name}: {100 * frac:.1f}% of elements outside rtol 1e-3 / atol 1e-4 (bound {100 * frac_bound:.1f}%)"
    return r


_full = {}


def full_model():
    """The 1.26 B-parameter config-2 UNet with random weights (recipe scaling) + an SD first conv; one per session."""
    if "m" not in _full:
        cfg = UNetConfig()
        dev = torch.device(DEV)
        sd = random_state_dict(cfg, dev, seed=3)
        g = torch.Generator(device=dev)
        g.manual_seed(11)
        fc = {"weight": torch.randn(cfg.model_channels, cfg.in_channels, 3, 3, device=dev, generator=g) * 0.16,
              "bias": torch.zeros(cfg.model_channels, device=dev)}
        # ARITHMETIC parity: both sides get weight matrices that are fp16-representable (the engine's split 1x1-conv weights then have Wlo = 0:
        # their third pass adds exact zeros).  The comparison against UNROUNDED fp32 weights -- where the third pass counts -- is
        # test_unrounded_reference_weights_at_bench_batch below.
        sd = {k: (v.half().float() if v.dim() >= 2 else v) for k, v in sd.items()}
        m = UNetModel(cfg, sd, device=DEV, sd_first_conv={k: v.cpu().numpy() for k, v in fc.items()})
        m.grounding_tokenizer_input = GroundingNetInput()
        # oracle weights: the same fp16-representable matrices
        sd_cpu = {k: (v.detach().float().cpu().half().float() if v.dim() >= 2 else v.detach().float().cpu()) for k, v in sd.items()}
        fc_cpu = {k: (v.float().cpu().half().float() if v.dim() >= 2 else v.float().cpu()) for k, v in fc.items()}
        del sd
        torch.cuda.empty_cache()
        _full.update(m=m, sd=sd_cpu, fc=fc_cpu, cfg=cfg)
    return _full["m"], _full["sd"], _full["fc"], _full["cfg"]


def cfg_batch(cfg, B, hw, n_boxes, seed):
    """[cond ; uncond] conditioning exactly as the sampler builds it (sampler.py / plms.py:115-124)."""
    inp = {k: T(v) for k, v in recipe.synth_inputs(cfg, B, hw, n_boxes=n_boxes, n_rel=3, seed=seed).items()}
    z = torch.zeros_like
    cat = lambda a, b: torch.cat([a, b], 0)
    two = dict(context=cat(inp["context"], inp["uc"]), relations=cat(inp["relations"], inp["relations"]),
               boxes=cat(inp["boxes"], z(inp["boxes"])), masks=cat(inp["masks"], z(inp["masks"])),
               positive_embeddings=cat(inp["positive_embeddings"], z(inp["positive_embeddings"])))
    return inp, two


class _NoRound:
    """x whose .half().float() is the identity (oracle_one rounds the latent to fp16 unless told otherwise)"""

    def __init__(self, t):
        self.t = t

    def __getitem__(self, i):
        return _NoRound(self.t[i])

    def half(self):
        return self

    def float(self):
        return self.t


def oracle_one(sd, cfg, inp, k, cond, tval, fuser_scale=1.0, first_conv=None, round_x=True, round_ctx=True):
    """fp32 oracle for sample k of the batch: conditional, or its null-grounding / empty-prompt twin.  ``round_ctx``: the text context / relation
    tokens are rounded to fp16 first (what the default engine's hoisted K / V projections read); False = the reference's own fp32 tensors (strict mode)."""
    s = lambda a: a[k:k + 1]
    z = torch.zeros_like
    if not round_x:          # the reference's own input: the fp32 latent (the engine's first conv takes it as [hi | lo], gl_set_option 38)
        inp = dict(inp, x=_NoRound(inp["x"]))
    if not round_ctx:
        inp = dict(inp, context=_NoRound(inp["context"]), uc=_NoRound(inp["uc"]), relations=_NoRound(inp["relations"]))
    torch.set_num_threads(min(32, max(1, os.cpu_count() or 1)))
    t = torch.full((1,), int(tval), dtype=torch.long)
    with torch.no_grad():
        if cond:
            return unet_ref.unet_forward(sd, cfg, s(inp["x"]).half().float(), t, s(inp["context"]).half().float(),
                                         s(inp["relations"]).half().float(), s(inp["boxes"]), s(inp["masks"]), s(inp["positive_embeddings"]),
                                         fuser_scale=fuser_scale, first_conv=first_conv)
        return unet_ref.unet_forward(sd, cfg, s(inp["x"]).half().float(), t, s(inp["uc"]).half().float(),
                                     s(inp["relations"]).half().float(), z(s(inp["boxes"])), z(s(inp["masks"])),
                                     z(s(inp["positive_embeddings"])), fuser_scale=fuser_scale, first_conv=first_conv)


# measured on MI355X (round 2): see DESIGN.md section 4; asserts are <= 1.5x these
# measured on MI355X (round 4: split-fp16 1x1 convs, GroupNorm on the fp32 stream, fp32 emb rows / fuser objs / LN3 pooling): rel-L2 6.54e-4 .. 7.34e-4,
# 26.0 .. 28.9 % outside over all (batch, mode) cases and both 768px rows; round 3 (fp16 copies): 1.11e-3 .. 1.23e-3, 44 .. 48 %.  Bounds = 1.1x the worst case.
BOUND_FULL = 8.1e-4
FRAC_FULL = 0.32             # elements outside north_star's rtol 1e-3 / atol 1e-4


@pytest.mark.parametrize("B", [4, 8, 16], ids=["configs1_2B8", "configs3_2B16", "configs4_2B32"])
def test_default_mode_config2_shapes_at_bench_batch_fp16_operand_bound_vs_oracle(B):
    """configs[1] (B = 4 -> 2B = 8), configs[3]'s per-GPU batch (64 / 8 GPUs = 8 -> 2B = 16: other tile / split-K / 8-wave
    dispatch decisions than either neighbour) and configs[4] (B = 16 -> 2B = 32): the 2B batch the sampler launches, checked
    against the oracle on one conditional sample (k = 1) and its unconditional twin (row B + 1); fuser on, then the
    scale-0 / SD-first-conv form the last 35 sampling steps use."""
    model, sd, fc, cfg = full_model()
    hw, k = 64, 1
    inp, two = cfg_batch(cfg, B, hw, 8, seed=2024)
    eng = model.engine
    eng.set_conditioning(two["context"], two["relations"], two["boxes"], two["masks"], two["positive_embeddings"], hw)
    x = inp["x"].half().float().to(DEV)          # fp16-representable latent on both sides (arithmetic parity; the first conv's lo channels are 0)
    e_on = eng.forward(x, 481.0, 1.0, False, 2).clone()
    e_off = eng.forward(x, 201.0, 0.0, True, 2).clone()
    assert e_on.shape == (2 * B, 4, hw, hw)
    r = [report(f"2B={2 * B} cond  fuser on ", e_on[k:k + 1], oracle_one(sd, cfg, inp, k, True, 481), FRAC_FULL),
         report(f"2B={2 * B} uncond fuser on ", e_on[B + k:B + k + 1], oracle_one(sd, cfg, inp, k, False, 481), FRAC_FULL),
         report(f"2B={2 * B} cond  fuser off", e_off[k:k + 1], oracle_one(sd, cfg, inp, k, True, 201, 0.0, fc), FRAC_FULL)]
    assert max(r) < BOUND_FULL, r
    # samples are independent: every row of the batch is a different image, and a replay is deterministic
    assert rel_l2(e_on[0:1], e_on[1:2]) > 1e-2 and rel_l2(e_on[0:1], e_on[B:B + 1]) > 1e-3
    assert torch.equal(e_on, eng.forward(x, 481.0, 1.0, False, 2))


def test_unrounded_reference_weights_at_bench_batch():
    """The reference's weights are fp32; the engine stores fp16 (+ the Wlo halves of the three kinds of 1x1 conv, gl_set_option 45).  Engine
    packed from UNROUNDED random fp32 weights vs the fp32 oracle on those same weights at configs[1]'s batch: the error now includes the weight
    storage.  Measured (round 4, bench.py's sample): 9.5e-4 / 39 % outside with the third pass, 1.20e-3 / 47 % on the fp16 weights alone (key 45 = 0)."""
    cfg = UNetConfig()
    dev = torch.device(DEV)
    sd = random_state_dict(cfg, dev, seed=3)
    m = UNetModel(cfg, sd, device=DEV, allow_missing_sd_conv=True)
    sd_cpu = {k: v.detach().float().cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    B, hw, k = 4, 64, 1
    inp, two = cfg_batch(cfg, B, hw, 8, seed=2024)
    eng = m.engine
    eng.set_conditioning(two["context"], two["relations"], two["boxes"], two["masks"], two["positive_embeddings"], hw)
    x = inp["x"].to(DEV)
    ref = oracle_one(sd_cpu, cfg, inp, k, True, 481, round_x=False)
    r3 = report("2B=8 cond fuser on, fp32 reference weights, three-pass 1x1 convs", eng.forward(x, 481.0, 1.0, False, 2)[k:k + 1], ref, 0.47)
    ops.set_option(45, 0)
    ops.set_option(38, 0)
    try:
        r2 = report("2B=8 cond fuser on, fp32 reference weights, fp16 weights only  ", eng.forward(x, 481.0, 1.0, False, 2)[k:k + 1], ref)
    finally:
        ops.set_option(45, 1024)
        ops.set_option(38, 1)
    assert r3 < 1.3e-3 and r3 < 0.93 * r2, (r3, r2)
    del m, eng
    torch.cuda.empty_cache()


_strict = {}


def strict_model():
    """The config-2 UNet on the SPLIT weight layout, packed from unrounded fp32 random weights, engine in strict mode; one per session."""
    if "m" not in _strict:
        import dataclasses
        cfg = dataclasses.replace(UNetConfig(), split_weights=True)
        dev = torch.device(DEV)
        sd = random_state_dict(cfg, dev, seed=3)
        g = torch.Generator(device=dev)
        g.manual_seed(11)
        fc = {"weight": torch.randn(cfg.model_channels, cfg.in_channels, 3, 3, device=dev, generator=g) * 0.16,
              "bias": torch.zeros(cfg.model_channels, device=dev)}
        m = UNetModel(cfg, sd, device=DEV, sd_first_conv={k: v.cpu().numpy() for k, v in fc.items()})
        m.grounding_tokenizer_input = GroundingNetInput()
        sd_cpu = {k: v.detach().float().cpu() for k, v in sd.items()}
        fc_cpu = {k: v.float().cpu() for k, v in fc.items()}
        del sd
        torch.cuda.empty_cache()
        _strict.update(m=m, sd=sd_cpu, fc=fc_cpu, cfg=cfg)
    return _strict["m"], _strict["sd"], _strict["fc"], _strict["cfg"]


def test_strict_mode_meets_north_star_tolerance_at_bench_batch():
    """STRICT mode (gl_set_handle_option 50 on a split_weights handle: every matrix product on split-fp16 operands) at
    configs[1]'s batch against the fp32 oracle: north_star's elementwise rtol 1e-3 / atol 1e-4 must hold for >= 99 % of the output
    elements -- (i) on the reference's own UNROUNDED fp32 weights and fp32 latent (three passes: + x.Wlo), cond / uncond / scale-0 + SD
    conv; (ii) with the third pass off (key 51 = 0) on fp16-representable weights, where only the q projections' folded softmax scale is
    left unsplit.  The default mode of the same handle is printed beside it (26-29 % outside on representable weights, 33-38 % on fp32 ones)."""
    import time
    m, sd_cpu, fc_cpu, cfg = strict_model()
    m.engine.clear_options()
    B, hw, k = 4, 64, 1
    inp, two = cfg_batch(cfg, B, hw, 8, seed=2024)
    eng = m.engine
    eng.set_conditioning(two["context"], two["relations"], two["boxes"], two["masks"], two["positive_embeddings"], hw)
    x = inp["x"].to(DEV)                          # the fp32 latent itself
    # the reference's own tensors everywhere: fp32 weights, fp32 latent, fp32 context / relation tokens
    refs = [("cond  fuser on ", oracle_one(sd_cpu, cfg, inp, k, True, 481, round_x=False, round_ctx=False), k, 481.0, 1.0, False),
            ("uncond fuser on ", oracle_one(sd_cpu, cfg, inp, k, False, 481, round_x=False, round_ctx=False), B + k, 481.0, 1.0, False),
            ("cond  fuser off", oracle_one(sd_cpu, cfg, inp, k, True, 201, 0.0, fc_cpu, round_x=False, round_ctx=False), k, 201.0, 0.0, True)]
    for name, ref, row, t, fs, sdc in refs:
        report(f"default mode, 2B=8 {name}, fp32 reference weights", eng.forward(x, t, fs, sdc, 2)[row:row + 1], ref)
    eng.set_option(50, 1)
    worst = 0.0
    for name, ref, row, t, fs, sdc in refs:
        out = eng.forward(x, t, fs, sdc, 2)
        worst = max(worst, report(f"STRICT (3 passes), 2B=8 {name}, fp32 reference weights", out[row:row + 1], ref, 0.001))
    assert worst < 3e-5, worst
    # round 6: the same forward on the round-5 kernel forms (K-walk three-pass products, key 52 = 0; attn_split_kernel, key 53 = 2): same products,
    # other fp32 summation order -- the two builds of the strict forward must agree far inside the tolerance, and both with the oracle
    name, ref, row, t, fs, sdc = refs[0]
    new_forms = eng.forward(x, t, fs, sdc, 2).clone()
    eng.set_option(52, 0)
    eng.set_option(53, 2)
    old_forms = eng.forward(x, t, fs, sdc, 2).clone()
    eng.set_option(52, 1)
    eng.set_option(53, 0)
    d_forms = rel_l2(new_forms, old_forms.cpu())
    print(f"[strict] round-6 kernel forms vs round-5 forms, whole 2B=8 forward: rel_l2 {d_forms:.2e}; round-5 forms vs oracle {rel_l2(old_forms[row:row + 1], ref):.2e}")
    assert d_forms < 1e-5 and rel_l2(old_forms[row:row + 1], ref) < 3e-5
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        eng.forward(x, 481.0, 1.0, False, 2)
    torch.cuda.synchronize()
    t_on = (time.time() - t0) / 5 * 1e3
    t0 = time.time()
    for _ in range(5):
        eng.forward(x, 201.0, 0.0, True, 2)
    torch.cuda.synchronize()
    print(f"[strict] forward of the 2B=8 batch: {t_on:.1f} ms fuser on, {(time.time() - t0) / 5 * 1e3:.1f} ms fuser off (three passes)")
    # for information: activations split only (key 51 = 0, two passes) against the oracle on fp16-ROUNDED weight matrices.  Not a clean pairing --
    # the first conv keeps its [Whi | Whi | Wlo] packing and the q projections their folded softmax scale, whose residuals only the third
    # pass carries (measured 3.5e-4 / 9.7 % outside: first-conv weight rounding 2.9e-4 of it, profiles/r4_weight_rounding_attribution.txt)
    eng.set_option(51, 0)
    eng.set_conditioning(two["context"], two["relations"], two["boxes"], two["masks"], two["positive_embeddings"], hw)     # (not needed any more: the strict hoists follow key 51 lazily)
    sd_r = {k_: (v.half().float() if v.dim() >= 2 else v) for k_, v in sd_cpu.items()}
    ref_r = oracle_one(sd_r, cfg, inp, k, True, 481, round_x=False, round_ctx=False)
    report("STRICT (2 passes: activations split), 2B=8 cond fuser on, fp16-ROUNDED oracle weights", eng.forward(x, 481.0, 1.0, False, 2)[k:k + 1], ref_r)
    eng.clear_options()


def test_strict_mode_config3_768px():
    """configs[2] in strict mode: 96x96 latents (9216 queries x 9246 fuser keys on the 8-wave split attention kernel, 12x12 maps ragged against
    the tiles), B = 2, 16 boxes: cond and uncond rows of the 2B batch against the fp32 oracle on the reference's own fp32 tensors."""
    m, sd_cpu, fc_cpu, cfg = strict_model()
    eng = m.engine
    eng.clear_options()
    eng.set_option(50, 1)
    B, hw, k = 2, 96, 1
    inp, two = cfg_batch(cfg, B, hw, 16, seed=31)
    eng.set_conditioning(two["context"], two["relations"], two["boxes"], two["masks"], two["positive_embeddings"], hw)
    x = inp["x"].to(DEV)
    e = eng.forward(x, 481.0, 1.0, False, 2).clone()
    r = [report("768px STRICT cond", e[k:k + 1], oracle_one(sd_cpu, cfg, inp, k, True, 481, round_x=False, round_ctx=False), 0.001),
         report("768px STRICT uncond", e[B + k:B + k + 1], oracle_one(sd_cpu, cfg, inp, k, False, 481, round_x=False, round_ctx=False), 0.001)]
    assert max(r) < 3e-5, r
    assert torch.equal(e, eng.forward(x, 481.0, 1.0, False, 2))
    eng.clear_options()


def test_config4_rollout_batch16_plms_runs():
    """configs[4]'s denoise stage: 16 prompts through ``denoise`` (5 PLMS steps, CFG 7.5, alpha_type [0.3, 0, 0.7])."""
    model, sd, fc, cfg = full_model()
    model.first_conv_type = "GLIGEN"
    B, hw = 16, 64
    inp, _ = cfg_batch(cfg, B, hw, 8, seed=7)
    batch = dict(boxes=inp["boxes"], masks=inp["masks"], text_embeddings=inp["positive_embeddings"])
    am = (model, None, None, LatentDiffusion(device=DEV), {})
    lat = denoise(am, inp["context"], inp["uc"], inp["relations"], batch, inp["x"].to(DEV), [0.3, 0.0, 0.7], 7.5, steps=5)
    assert lat.shape == (B, 4, hw, hw) and torch.isfinite(lat).all()
    assert model.first_conv_type == "SD"
    # a sample's latent does not depend on what else is in the batch (up to the fp16 pipeline's dispatch-dependent rounding)
    model.first_conv_type = "GLIGEN"
    sub = {k: v[:4] for k, v in inp.items()}
    lat4 = denoise(am, sub["context"], sub["uc"], sub["relations"], {k: v[:4] for k, v in batch.items()}, sub["x"].to(DEV),
                   [0.3, 0.0, 0.7], 7.5, steps=5)
    r = report("B=16 vs B=4 rows", lat[:4], lat4)
    assert r < 4e-3, r              # measured 2.7e-3: other tile / split-K choices at another batch change fp32 summation order


def test_default_mode_config3_768px_fp16_operand_bound_vs_oracle_and_50_steps():
    """configs[2]: 768x768 -> 96x96 latents (9216 / 2304 / 576 / 144 tokens per level, 12x12 convs with M = 2B*144
    ragged against the 128-row tiles), B = 2, 16 grounding boxes: whole UNet vs the oracle on sample 1, null-grounding
    and determinism properties, then the full 50-step sampling run."""
    model, sd, fc, cfg = full_model()
    model.first_conv_type = "GLIGEN"
    B, hw, k = 2, 96, 1
    inp, two = cfg_batch(cfg, B, hw, 16, seed=31)
    assert float(inp["masks"].sum()) == 32.0
    eng = model.engine
    eng.set_conditioning(two["context"], two["relations"], two["boxes"], two["masks"], two["positive_embeddings"], hw)
    x = inp["x"].to(DEV)
    e = eng.forward(x, 481.0, 1.0, False, 2).clone()
    r = [report("768px cond", e[k:k + 1], oracle_one(sd, cfg, inp, k, True, 481), FRAC_FULL),
         report("768px uncond", e[B + k:B + k + 1], oracle_one(sd, cfg, inp, k, False, 481), FRAC_FULL)]
    assert max(r) < BOUND_FULL, r
    assert torch.equal(e, eng.forward(x, 481.0, 1.0, False, 2))
    # the unconditional half ignores boxes / phrase embeddings entirely (null tokens; rela_fuse == LN3)
    two2 = dict(two)
    two2["boxes"] = torch.cat([inp["boxes"], torch.rand_like(inp["boxes"])], 0)
    two2["positive_embeddings"] = torch.cat([inp["positive_embeddings"], torch.randn_like(inp["positive_embeddings"])], 0)
    eng.set_conditioning(two2["context"], two2["relations"], two2["boxes"], two2["masks"], two2["positive_embeddings"], hw)
    assert torch.equal(e, eng.forward(x, 481.0, 1.0, False, 2))
    batch = dict(boxes=inp["boxes"], masks=inp["masks"], text_embeddings=inp["positive_embeddings"])
    am = (model, None, None, LatentDiffusion(device=DEV), {})
    lat = denoise(am, inp["context"], inp["uc"], inp["relations"], batch, x, [0.3, 0.0, 0.7], 7.5, steps=50)
    assert lat.shape == (B, 4, hw, hw) and torch.isfinite(lat).all() and float(lat.abs().max()) < 1e3
    model.first_conv_type = "GLIGEN"
    lat2 = denoise(am, inp["context"], inp["uc"], inp["relations"], batch, x, [0.3, 0.0, 0.7], 7.5, steps=50)
    assert torch.equal(lat, lat2), "the sampling run is deterministic (fixed reduction orders, graph replay)"


def test_default_mode_config0_10_steps_fp16_operand_bound_vs_oracle_plms():
    """configs[0] (the txt2img plumbing case -- 1 image, 64x64 latent, S = 10 PLMS steps,
    2 grounding boxes, CFG 7.5, alpha_type [0.3, 0, 0.7] -> 3 fuser-on and 7 fuser-off steps with the SD first conv,
    22 UNet evaluations) through ``denoise`` on the FULL-SIZE model, against the oracle's PLMS loop driving the oracle UNet
    (plms.py:58-163 restated in oracle/plms_ref.py).  The B = 1 batch takes the small-grid dispatch paths (2B = 2) that no
    other whole-model test reaches."""
    from oracle import plms_ref
    model, sd, fc, cfg = full_model()
    model.first_conv_type = "GLIGEN"
    B, hw, S = 1, 64, 10
    inp, _ = cfg_batch(cfg, B, hw, 2, seed=99)
    assert float(inp["masks"].sum()) == 2.0
    batch = dict(boxes=inp["boxes"], masks=inp["masks"], text_embeddings=inp["positive_embeddings"])
    am = (model, None, None, LatentDiffusion(device=DEV), {})
    torch.manual_seed(5)
    lat = denoise(am, inp["context"], inp["uc"], inp["relations"], batch, inp["x"].to(DEV), [0.3, 0.0, 0.7], 7.5, steps=S)
    assert lat.shape == (B, 4, hw, hw) and model.first_conv_type == "SD"
    torch.set_num_threads(min(32, max(1, os.cpu_count() or 1)))
    z = torch.zeros_like
    state = dict(sd=False)

    def eps_fn(x, t, i, alpha):
        if alpha == 0:
            state["sd"] = True                       # permanent, like restore_first_conv_from_SD (openaimodel.py:393-411)
        first = fc if state["sd"] else None
        with torch.no_grad():
            e_c = unet_ref.unet_forward(sd, cfg, x, t, inp["context"], inp["relations"], inp["boxes"], inp["masks"],
                                        inp["positive_embeddings"], fuser_scale=float(alpha), first_conv=first)
            e_u = unet_ref.unet_forward(sd, cfg, x, t, inp["uc"], inp["relations"], z(inp["boxes"]), z(inp["masks"]),
                                        z(inp["positive_embeddings"]), fuser_scale=float(alpha), first_conv=first)
        return e_u + 7.5 * (e_c - e_u)
    ref = plms_ref.plms_sample(eps_fn, inp["x"], S, [0.3, 0.0, 0.7])
    r = report("configs[0] B=1 S=10 final latent", lat.cpu(), ref)
    assert r < 1.7e-3, r           # 22 chained evaluations; round 4: 1.10e-3 (round 3: 1.87e-3)


def test_config0_strict_mode_10_steps_vs_oracle_plms():
    """The strict twin of the test above: north_star's tolerance is claimed per forward, the sampler compounds 22 chained
    evaluations (plms.py:63-163: step-0 double evaluation, AB-2/3/4 history, 3 fuser-on + 7 fuser-off steps with the permanent SD first conv).
    Full-size UNet on the SPLIT weight layout packed from UNROUNDED fp32 weights, engine in strict mode, B = 1, S = 10, against the oracle's PLMS
    loop driving the fp32 oracle UNet on the same fp32 weights / latent / context: the FINAL LATENT must sit inside rtol 1e-3 / atol 1e-4
    elementwise (<= 1 % outside asserted, the fraction is printed) and within 1e-4 rel-L2."""
    from oracle import plms_ref
    model, sd, fc, cfg = strict_model()
    eng = model.engine
    eng.clear_options()
    eng.set_option(50, 1)
    model.first_conv_type = "GLIGEN"
    B, hw, S = 1, 64, 10
    inp, _ = cfg_batch(cfg, B, hw, 2, seed=99)
    batch = dict(boxes=inp["boxes"], masks=inp["masks"], text_embeddings=inp["positive_embeddings"])
    am = (model, None, None, LatentDiffusion(device=DEV), {})
    try:
        torch.manual_seed(5)
        lat = denoise(am, inp["context"], inp["uc"], inp["relations"], batch, inp["x"].to(DEV), [0.3, 0.0, 0.7], 7.5, steps=S)
    finally:
        eng.clear_options()
    assert lat.shape == (B, 4, hw, hw) and model.first_conv_type == "SD"
    torch.set_num_threads(min(32, max(1, os.cpu_count() or 1)))
    z = torch.zeros_like
    state = dict(sd=False)

    def eps_fn(x, t, i, alpha):
        if alpha == 0:
            state["sd"] = True
        first = fc if state["sd"] else None
        with torch.no_grad():
            e_c = unet_ref.unet_forward(sd, cfg, x, t, inp["context"], inp["relations"], inp["boxes"], inp["masks"],
                                        inp["positive_embeddings"], fuser_scale=float(alpha), first_conv=first)
            e_u = unet_ref.unet_forward(sd, cfg, x, t, inp["uc"], inp["relations"], z(inp["boxes"]), z(inp["masks"]),
                                        z(inp["positive_embeddings"]), fuser_scale=float(alpha), first_conv=first)
        return e_u + 7.5 * (e_c - e_u)
    ref = plms_ref.plms_sample(eps_fn, inp["x"], S, [0.3, 0.0, 0.7])
    r = report("configs[0] STRICT B=1 S=10 final latent (22 chained evaluations)", lat.cpu(), ref, 0.01)
    assert r < 1e-4, r
