"""The C++ engine behind the forward-level C ABI (gl_create / gl_load_weights / gl_set_conditioning / gl_unet_forward /
gl_plms_step, csrc/engine.hip) against an independent op-by-op restatement of the same launch sequence written in
Python over the op-level ABI (tests/engine_pyref.py).  Both drive the same kernels, so the outputs must be BITWISE
equal: this pins the C++ orchestration -- plan builder, weight table, buffer pool and aliasing, pointer / stride
arithmetic, the device-side box rectangles (vs host.box_rects), graph capture / replay -- independently of precision.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(__file__))
import golden_cases as gc
from engine_pyref import PyRefEngine
from layoutllm_t2i_amd import host, ops, recipe
from layoutllm_t2i_amd.arch import TINY, UNetConfig
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.weights import pack_state_dict, random_state_dict

DEV = "cuda:0"
T = torch.from_numpy


def same(a, b):
    """bitwise equality, NaNs included (a poisoned sample is NaN everywhere in both)"""
    return a.shape == b.shape and torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))


def engines(cfg, seed=0, random=False):
    dev = torch.device(DEV)
    sd = random_state_dict(cfg, dev, seed=seed) if random else recipe.state_dict(cfg, seed)
    fc = recipe.sd_first_conv(cfg, seed)
    P = pack_state_dict(sd, cfg, dev, fc)
    return UNetEngine(P), PyRefEngine(P)


@pytest.mark.parametrize("variant", ["normal", "degenerate", "null", "clamp", "maskgap", "empty_slice", "negative"])
def test_cpp_engine_equals_python_launch_sequence_tiny(variant):
    """tiny UNet, B = 3, every box-rectangle edge case of attention.py:321-346 (the C++ side derives the rectangles
    on the DEVICE, the Python side with host.box_rects); fuser on / skipped + SD conv; reps = 1 and the 2B CFG batch"""
    eng, ref = engines(TINY)
    B, hw = 3, 16
    inp = {k: T(v) for k, v in recipe.synth_inputs(TINY, B, hw, n_boxes=4, n_rel=3, seed=5).items()}
    if variant == "negative":
        boxes = np.zeros((B, 30, 4), np.float32)
        masks = np.zeros((B, 30), np.float32)
        boxes[0, :2] = [(-0.5, 0.0, 0.5, 0.5), (0.2, 0.2, 0.8, 0.8)]        # slice(-8, 8) on 16 -> [8, 8): empty -> NaN sample
        boxes[1, :3] = [(0.999, 0.0, 1.0, 1.0), (0.0, -0.3, 0.7, 0.6), (0.1, 0.1, 0.9, 2.5)]
        boxes[2, :1] = [(0.3, 0.3, 0.30001, 0.9)]                           # zero width at every level -> break at box 0
        masks[0, :2] = 1
        masks[1, :3] = 1
        masks[2, :1] = 1
    else:
        boxes, masks = gc.rela_boxes(variant, B)
    inp["boxes"], inp["masks"] = T(boxes), T(masks)
    x = inp["x"].to(DEV)
    for e in (eng, ref):
        e.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], hw)
    for fs, sdc in ((1.0, False), (0.0, True), (0.5, False)):      # (0.5 x gate is exact in fp32 and in python's double alike)
        a = eng.forward(x, 481.0, fs, sdc, 1).clone()
        b = ref.forward(x, 481.0, fs, sdc, 1).clone()
        assert same(a, b), (variant, fs, float((torch.nan_to_num(a) - torch.nan_to_num(b)).abs().max()))
        if variant in ("empty_slice", "negative"):
            assert torch.isnan(a[0]).all() and torch.isfinite(a[2]).all()
        elif variant != "null":
            assert torch.isfinite(a).all()
    # per-sample timesteps on the device
    tt = torch.tensor([981.0, 481.0, 1.0])
    assert same(eng.forward(x, tt, 1.0, False, 1).clone(), ref.forward(x, tt, 1.0, False, 1).clone())
    # the 2B [cond ; uncond] batch sharing one latent
    z = torch.zeros_like
    cat = lambda p, q: torch.cat([p, q], 0)
    for e in (eng, ref):
        e.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                           cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), hw)
    assert same(eng.forward(x, 21.0, 1.0, False, 2).clone(), ref.forward(x, 21.0, 1.0, False, 2).clone())


def test_cpp_engine_graph_replay_eager_and_reconditioning():
    eng, ref = engines(TINY)
    B, hw = 2, 16
    inp = {k: T(v) for k, v in recipe.synth_inputs(TINY, B, hw, n_boxes=4, n_rel=3, seed=9).items()}
    x = inp["x"].to(DEV)
    eng.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], hw)
    eng.use_graphs = False
    a = eng.forward(x, 500.0, 1.0, False, 1).clone()
    eng.use_graphs = True
    b = eng.forward(x, 500.0, 1.0, False, 1).clone()
    c = eng.forward(x, 500.0, 1.0, False, 1).clone()
    assert torch.equal(a, b) and torch.equal(b, c), "graph replay must be bit-identical to eager launches"
    assert 100 < eng.num_launches() < 2000
    # new conditioning of the same shape re-uses the captured graph (the hoisted buffers are updated in place)
    inp2 = {k: T(v) for k, v in recipe.synth_inputs(TINY, B, hw, n_boxes=2, n_rel=1, seed=10).items()}
    eng.set_conditioning(inp2["context"], inp2["relations"], inp2["boxes"], inp2["masks"], inp2["positive_embeddings"], hw)
    ref.set_conditioning(inp2["context"], inp2["relations"], inp2["boxes"], inp2["masks"], inp2["positive_embeddings"], hw)
    d = eng.forward(x, 500.0, 1.0, False, 1).clone()
    assert same(d, ref.forward(x, 500.0, 1.0, False, 1).clone()) and not torch.equal(d, c)
    # a bigger batch / other latent size grows the pool: graphs are re-captured, results still equal
    inp3 = {k: T(v) for k, v in recipe.synth_inputs(TINY, 5, 24, n_boxes=3, n_rel=2, seed=11).items()}
    for e in (eng, ref):
        e.set_conditioning(inp3["context"], inp3["relations"], inp3["boxes"], inp3["masks"], inp3["positive_embeddings"], 24)
    x3 = inp3["x"].to(DEV)
    assert same(eng.forward(x3, 77.0, 1.0, False, 1).clone(), ref.forward(x3, 77.0, 1.0, False, 1).clone())
    # ... and going back to the first shape still matches
    for e in (eng, ref):
        e.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], hw)
    assert torch.equal(eng.forward(x, 500.0, 1.0, False, 1), a)
    # scale-0 skip is an exact identity of the fuser: compare with the fuser EXECUTED under zero gates (test knob 20)
    e0 = eng.forward(x, 500.0, 0.0, False, 1).clone()
    ops.set_option(20, 1)
    try:
        e0_exec = eng.forward(x, 500.0, 0.0, False, 1).clone()
    finally:
        ops.set_option(20, 0)
    assert torch.equal(e0, e0_exec), "skipping the fuser at scale 0 must equal executing it with zero gates"
    with pytest.raises(Exception):
        eng.forward(inp3["x"].to(DEV), 1.0)                     # latent does not match the conditioning batch / size


def test_cpp_engine_equals_python_launch_sequence_full_model():
    """the real 1.26 B-parameter config at 64x64, 2B = 4 (split-K, K-split, 64-row tiles all in play)"""
    cfg = UNetConfig()
    eng, ref = engines(cfg, seed=3, random=True)
    B, hw = 2, 64
    inp = {k: T(v) for k, v in recipe.synth_inputs(cfg, B, hw, n_boxes=8, n_rel=3, seed=77).items()}
    z = torch.zeros_like
    cat = lambda p, q: torch.cat([p, q], 0)
    for e in (eng, ref):
        e.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                           cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), hw)
    x = inp["x"].to(DEV)
    for fs, sdc in ((1.0, False), (0.0, True)):
        a = eng.forward(x, 481.0, fs, sdc, 2).clone()
        b = ref.forward(x, 481.0, fs, sdc, 2).clone()
        assert torch.isfinite(a).all() and torch.equal(a, b), float((a - b).abs().max())
    n = eng.num_launches()
    print(f"[engine] kernel launches per forward (fuser off): {n}; engine pool {eng.pool_bytes() / 2**20:.0f} MiB")
    del eng, ref
    torch.cuda.empty_cache()


def test_plms_step_composite_equals_separate_calls():
    """gl_plms_step == gl_unet_forward + gl_cfg_combine + gl_plms_update issued one by one, bitwise"""
    eng, _ = engines(TINY)
    B, hw = 2, 16
    inp = {k: T(v) for k, v in recipe.synth_inputs(TINY, B, hw, n_boxes=4, n_rel=3, seed=21).items()}
    z = torch.zeros_like
    cat = lambda p, q: torch.cat([p, q], 0)
    eng.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                         cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), hw)
    x = inp["x"].to(DEV).contiguous()
    old = [torch.randn_like(x) for _ in range(3)]
    sched = host.make_schedule(50, host.alphas_cumprod())
    sq_at, s1m, sq_ap, dirc = host.step_coefs(sched, 30)
    coefs, div = host.PLMS_COEFS[3]
    e_out, x_out = torch.empty_like(x), torch.empty_like(x)
    eng.plms_step(x, x, x_out, e_out, [e_out] + old, coefs, div, 601.0, 2, 7.5, 1.0, False, sq_at, s1m, sq_ap, dirc)
    eps = eng.forward(x, 601.0, 1.0, False, 2).clone()
    e2 = ops.cfg_combine(eps, 7.5, torch.empty_like(x))
    x2 = ops.plms_update(x, e2, old, coefs, div, sq_at, s1m, sq_ap, dirc, torch.empty_like(x))
    assert torch.equal(e_out, e2) and torch.equal(x_out, x2)
    # in-place form the sampler uses (x_out aliases x_base / x_eval)
    xa = x.clone()
    eng.plms_step(xa, xa, xa, e_out, [e_out] + old, coefs, div, 601.0, 2, 7.5, 1.0, False, sq_at, s1m, sq_ap, dirc)
    assert torch.equal(xa, x2)


def test_two_engines_interleaved_and_option_toggles():
    """Two engines of DIFFERENT architectures alive in one process, forwards interleaved, a process-global A/B knob flipped
    in between (gl_set_option bumps an epoch that invalidates captured graphs): every output must equal the engine's own
    solo result bit for bit -- handles share no activation state, and the knob state never leaks into a replayed graph."""
    cfg_b = UNetConfig(image_size=16, model_channels=128, num_heads=8, channel_mult=(1, 2), attention_resolutions=(1, 2), num_res_blocks=1,
                       context_dim=128, pos_in_dim=64, pos_out_dim=128)
    ea, _ = engines(TINY, seed=1)
    eb, _ = engines(cfg_b, seed=2)
    ia = {k: T(v) for k, v in recipe.synth_inputs(TINY, 2, 16, n_boxes=5, n_rel=2, seed=21).items()}
    ib = {k: T(v) for k, v in recipe.synth_inputs(cfg_b, 3, 24, n_boxes=9, n_rel=4, seed=22).items()}
    xa, xb = ia["x"].to(DEV), ib["x"].to(DEV)
    ea.set_conditioning(ia["context"], ia["relations"], ia["boxes"], ia["masks"], ia["positive_embeddings"], 16)
    solo_a = [ea.forward(xa, 481.0, 1.0, False, 1).clone(), ea.forward(xa, 201.0, 0.0, True, 1).clone()]
    eb.set_conditioning(ib["context"], ib["relations"], ib["boxes"], ib["masks"], ib["positive_embeddings"], 24)
    solo_b = [eb.forward(xb, 481.0, 1.0, False, 1).clone(), eb.forward(xb, 201.0, 0.0, True, 1).clone()]
    for rnd_ in range(3):
        assert torch.equal(ea.forward(xa, 481.0, 1.0, False, 1), solo_a[0])
        assert torch.equal(eb.forward(xb, 201.0, 0.0, True, 1), solo_b[1])
        assert torch.equal(ea.forward(xa, 201.0, 0.0, True, 1), solo_a[1])
        assert torch.equal(eb.forward(xb, 481.0, 1.0, False, 1), solo_b[0])
        if rnd_ == 0:
            # a knob that changes the launch sequence (LayerNorm inside rela_merge or as its own launch): results are
            # bit-identical by construction, the graphs must be re-captured, and flipping it back restores the launch count
            n1 = ea.num_launches()
            ops.set_option(25, 0)
            try:
                assert torch.equal(ea.forward(xa, 481.0, 1.0, False, 1), solo_a[0])
                assert ea.num_launches() > n1
                assert torch.equal(eb.forward(xb, 481.0, 1.0, False, 1), solo_b[0])
            finally:
                ops.set_option(25, 1)
    # destroying one engine leaves the other intact
    del ea
    torch.cuda.synchronize()
    assert torch.equal(eb.forward(xb, 481.0, 1.0, False, 1), solo_b[0])


def test_handle_option_overrides_are_per_engine():
    """gl_set_handle_option: a knob overridden on ONE engine is in effect only while that engine's entry points run -- the
    other engine of the process, and op-level calls, keep the process defaults ('two hosts in one process still
    share tuning state').  Checked with the knob that moves GEMMs / convs to the 8-wave kernel (a different fp32 summation
    order, so outputs differ in the last bits and the launch counter moves) and with one that changes the launch sequence."""
    cfg = UNetConfig(image_size=16, model_channels=128, num_heads=8, channel_mult=(1, 2), attention_resolutions=(1, 2), num_res_blocks=1,
                     context_dim=128, pos_in_dim=64, pos_out_dim=128)
    dev = torch.device(DEV)
    P = pack_state_dict(recipe.state_dict(cfg, 3), cfg, dev, recipe.sd_first_conv(cfg, 3))
    e1, e2 = UNetEngine(P), UNetEngine(P)
    inp = {k: T(v) for k, v in recipe.synth_inputs(cfg, 3, 24, n_boxes=9, n_rel=4, seed=5).items()}
    x = inp["x"].to(DEV)
    for e in (e1, e2):
        e.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], 24)
    base = e1.forward(x, 481.0, 1.0, False, 1).clone()
    assert torch.equal(e2.forward(x, 481.0, 1.0, False, 1), base)
    # reference for "8-wave kernel wherever it applies": the PROCESS default flipped, then restored
    ops.set_option(30, 2)
    try:
        c0 = ops.gemm8_launch_count()
        forced = e1.forward(x, 481.0, 1.0, False, 1).clone()
        n_forced = ops.gemm8_launch_count() - c0
    finally:
        ops.set_option(30, 1)
    assert n_forced > 0 and not torch.equal(forced, base)
    assert torch.equal(e1.forward(x, 481.0, 1.0, False, 1), base)
    # the same knob as an override of e2 alone
    e2.set_option(30, 2)
    c0 = ops.gemm8_launch_count()
    e2.use_graphs = False                      # eager: every forward launches (and counts) its kernels
    assert torch.equal(e2.forward(x, 481.0, 1.0, False, 1), forced)
    assert ops.gemm8_launch_count() - c0 > 0             # (n_forced counted a warm-up pass + a capture pass)
    c1 = ops.gemm8_launch_count()
    e1.use_graphs = False
    assert torch.equal(e1.forward(x, 481.0, 1.0, False, 1), base)            # e1 still sees the defaults ...
    a = torch.randn(512, 256, device=DEV).half()
    w = torch.randn(320, 256, device=DEV).half()
    ops.gemm(a, w, torch.empty(512, 320, dtype=torch.float16, device=DEV), None)     # ... and so does an op-level call
    assert ops.gemm8_launch_count() == c1
    e1.use_graphs = e2.use_graphs = True
    assert torch.equal(e2.forward(x, 481.0, 1.0, False, 1), forced)           # graph captured under the override
    assert torch.equal(e1.forward(x, 481.0, 1.0, False, 1), base)
    # a knob that changes the launch sequence, on e2 only
    n1 = e1.num_launches()
    e2.clear_options()
    assert torch.equal(e2.forward(x, 481.0, 1.0, False, 1), base)
    assert e2.num_launches() == n1
    e2.set_option(25, 0)
    assert torch.equal(e2.forward(x, 481.0, 1.0, False, 1), base) and e2.num_launches() > n1
    assert torch.equal(e1.forward(x, 481.0, 1.0, False, 1), base) and e1.num_launches() == n1
    with pytest.raises(Exception):
        e2.set_option(99, 1)
    with pytest.raises(Exception):
        e2.set_option(1, 1)                     # not a knob


def test_precision_modes_match_python_sequence_and_precise_is_closer_to_the_oracle():
    """gl_set_option 41 / 42: the default mode (split-fp16 activations for the three kinds of 1x1 conv, GroupNorm on
    the fp32 stream, fp32 first-conv output inside a ResBlock), the mode with fp16 h1, and round 3's fp16-copy mode are each
    BITWISE equal to the Python launch sequence in the same mode; against the fp32 oracle the default mode is the closest."""
    from oracle import unet_ref
    cfg = UNetConfig(image_size=16, model_channels=128, num_heads=8, channel_mult=(1, 2), attention_resolutions=(1, 2), num_res_blocks=1,
                     context_dim=128, pos_in_dim=64, pos_out_dim=128)
    dev = torch.device(DEV)
    sd = recipe.state_dict(cfg, 3)
    sd = {k: (np.asarray(v).astype(np.float16).astype(np.float32) if np.asarray(v).ndim >= 2 else np.asarray(v)) for k, v in sd.items()}     # fp16-representable matrices on both sides
    P = pack_state_dict(sd, cfg, dev, recipe.sd_first_conv(cfg, 3))
    eng, ref = UNetEngine(P), PyRefEngine(P)
    B, hw = 2, 16
    inp = {k: T(v) for k, v in recipe.synth_inputs(cfg, B, hw, n_boxes=5, n_rel=3, seed=8).items()}
    x = inp["x"].to(DEV)
    for e in (eng, ref):
        e.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], hw)
    osd = {k: (T(np.asarray(v)).float().half().float() if np.asarray(v).ndim >= 2 else T(np.asarray(v)).float()) for k, v in sd.items()}
    with torch.no_grad():
        want = unet_ref.unet_forward(osd, cfg, inp["x"].half().float(), torch.full((B,), 481), inp["context"].half().float(),
                                     inp["relations"].half().float(), inp["boxes"], inp["masks"], inp["positive_embeddings"])
    errs = {}
    try:
        for name, k41, k42 in (("precise", 1, 1), ("precise_h1_fp16", 1, 0), ("fp16_copies", 0, 0)):
            ops.set_option(41, k41)
            ops.set_option(42, k42)
            ref.precise, ref.h1_f32 = bool(k41), bool(k42)
            a = eng.forward(x, 481.0, 1.0, False, 1).clone()
            b = ref.forward(x, 481.0, 1.0, False, 1).clone()
            assert same(a, b), (name, float((a - b).abs().max()))
            assert same(eng.forward(x, 481.0, 0.0, True, 1).clone(), ref.forward(x, 481.0, 0.0, True, 1).clone()), name
            errs[name] = float((a.cpu() - want).norm() / want.norm())
    finally:
        ops.set_option(41, 1)
        ops.set_option(42, 1)
    print("[precision modes] rel-L2 vs fp32 oracle (fp16-rounded weights):", {k: f"{v:.3e}" for k, v in errs.items()})
    assert errs["precise"] < errs["precise_h1_fp16"] < errs["fp16_copies"], errs
    assert errs["precise"] < 0.75 * errs["fp16_copies"], errs


def test_shared_cond_uncond_prefix_matches_python_sequence_and_the_unshared_forward():
    """gl_set_option 44 (default on): a reps = 2 forward computes conv_in, the first ResBlock and proj_in .. attn1 of the first
    transformer ONCE on the shared latents.  Bitwise equal to the Python launch sequence doing the same; within fp32-summation-order
    rounding of the forward that computes both halves (other tile / split-K choices at half the rows); per-sample timesteps (a device
    tensor) and reps = 1 take the unshared path."""
    eng, ref = engines(TINY)
    B, hw = 2, 16
    inp = {k: T(v) for k, v in recipe.synth_inputs(TINY, B, hw, n_boxes=4, n_rel=3, seed=21).items()}
    z = torch.zeros_like
    cat = lambda p, q: torch.cat([p, q], 0)
    for e in (eng, ref):
        e.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                           cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), hw)
    x = inp["x"].to(DEV)
    n_shared = None
    for fs, sdc in ((1.0, False), (0.0, True)):
        a = eng.forward(x, 481.0, fs, sdc, 2).clone()
        assert same(a, ref.forward(x, 481.0, fs, sdc, 2).clone()), fs
        n_shared = eng.num_launches()
        ops.set_option(44, 0)
        try:
            ref.share_prefix = False
            b = eng.forward(x, 481.0, fs, sdc, 2).clone()
            assert same(b, ref.forward(x, 481.0, fs, sdc, 2).clone()), fs
        finally:
            ops.set_option(44, 1)
            ref.share_prefix = True
        r = float((a - b).norm() / b.norm())
        assert r < 1e-3 and not torch.equal(a[:B], a[B:]), r           # cond != uncond after the shared prefix
    # a per-sample timestep tensor: not shared (the halves may differ in t)
    tt = torch.tensor([481.0, 481.0, 481.0, 481.0])
    c = eng.forward(x, tt, 0.0, True, 2).clone()
    assert same(c, ref.forward(x, tt, 0.0, True, 2).clone()) and float((c - b).norm() / b.norm()) < 1e-6


def test_relation_chain_on_used_rows_matches_python_sequence_and_the_30_row_chain():
    """gl_set_option 43 (default on): the relation chain (norm1 / cross-attention / FeedForward of rela_fuse) runs on max nvalid (rounded
    up to 8) rows per sample instead of all 30 (attention.py:348-351 computes 30, uses nvalid).  Bitwise equal to the Python launch sequence
    doing the same; equal to the 30-row chain up to the tile / split-K choice of GEMMs with fewer rows; 26 boxes fall back to 30 rows."""
    eng, ref = engines(TINY)
    B, hw = 2, 16
    z = torch.zeros_like
    cat = lambda p, q: torch.cat([p, q], 0)
    try:
        for n_boxes in (3, 11, 26):
            inp = {k: T(v) for k, v in recipe.synth_inputs(TINY, B, hw, n_boxes=n_boxes, n_rel=3, seed=40 + n_boxes).items()}
            x = inp["x"].to(DEV)
            outs = []
            for opt in (1, 0):
                ops.set_option(43, opt)
                ref.rela_compact = bool(opt)
                for e in (eng, ref):
                    e.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                                       cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), hw)
                a = eng.forward(x, 481.0, 1.0, False, 2).clone()
                assert same(a, ref.forward(x, 481.0, 1.0, False, 2).clone()), (n_boxes, opt)
                outs.append(a)
            r = float((outs[0] - outs[1]).norm() / outs[1].norm())
            assert r < 2e-4, (n_boxes, r)
    finally:
        ops.set_option(43, 1)
        ref.rela_compact = True


def test_third_pass_of_the_1x1_convs_matches_python_sequence_and_moves_towards_fp32_weights():
    """gl_set_option 45 (default on, with key 41): skip_connection / proj_in / proj_out store [Whi | Wlo] rows and launches of more than 1024
    rows add the pass xhi.Wlo.  Bitwise equal to the Python launch sequence either way; with fp16-representable weights (Wlo = 0) the third
    pass adds exact zeros, so the outputs with and without it are bitwise equal."""
    eng, ref = engines(TINY)
    B, hw = 2, 32                      # 2B * 32 * 32 = 4096 rows at the first level
    inp = {k: T(v) for k, v in recipe.synth_inputs(TINY, B, hw, n_boxes=4, n_rel=3, seed=77).items()}
    z = torch.zeros_like
    cat = lambda p, q: torch.cat([p, q], 0)
    for e in (eng, ref):
        e.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                           cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), hw)
    x = inp["x"].to(DEV)
    outs = []
    try:
        for opt in (1024, 0):
            ops.set_option(45, opt)
            ops.set_option(38, 1 if opt else 0)          # ... and the split first conv (key 38) with it
            ref.w3 = opt
            ref.in_split = bool(opt)
            a = eng.forward(x, 481.0, 1.0, False, 2).clone()
            assert same(a, ref.forward(x, 481.0, 1.0, False, 2).clone()), opt
            outs.append(a)
    finally:
        ops.set_option(45, 1024)
        ops.set_option(38, 1)
        ref.w3 = 1024
        ref.in_split = True
    r = float((outs[0] - outs[1]).norm() / outs[1].norm())
    assert 1e-6 < r < 3e-3, r              # the recipe weights are fp32: the third pass moves the result at the weight-rounding level
