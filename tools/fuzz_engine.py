"""Randomised ARCHITECTURE sweep of the C++ engine (GPU box): small UNets with random widths, level counts, attention
levels, head counts, residual-block counts, latent sizes, batch sizes, box / relation counts and random layouts (including
degenerate and out-of-range boxes), each forward compared with the fp32 oracle on the host CPU (fp16-rounded weights) in
three modes: grounded, null grounding, fuser scale 0 with the SD first conv.  usage: python tools/fuzz_engine.py [seconds] [seed]
FUZZ_STRICT=1: the engine's STRICT mode (split weight layout, gl_set_handle_option 50) against the oracle on the UNROUNDED fp32 weights,
latent and conditioning tensors; a forward fails above rel-L2 1e-4 (default mode: 4e-3)."""
import os
import random
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from layoutllm_t2i_amd import recipe  # noqa: E402
from layoutllm_t2i_amd.arch import UNetConfig  # noqa: E402
from layoutllm_t2i_amd.model import GroundingNetInput, UNetModel  # noqa: E402
from oracle import unet_ref  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = "cuda:0"
T = torch.from_numpy
# FUZZ_OPTS="30=2": gl_set_option knobs for the sweep (e.g. force the 8-wave GEMM / conv kernel wherever it applies)
from layoutllm_t2i_amd import ops as _fz_ops  # noqa: E402
from layoutllm_t2i_amd._lib import init_device as _fz_init  # noqa: E402
_fz_init(0)
for _kv in filter(None, os.environ.get("FUZZ_OPTS", "").split(",")):
    _fz_ops.set_option(int(_kv.split("=")[0]), int(_kv.split("=")[1]))
torch.set_num_threads(16)
STRICT = os.environ.get("FUZZ_STRICT") == "1"
BOUND = 1e-4 if STRICT else 4e-3


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


t0, n, fails, worst = time.time(), 0, 0, 0.0
while time.time() - t0 < budget:
    nlev = rng.choice([2, 3, 4])
    mult = tuple([1] + sorted(rng.choice([1, 2, 3, 4]) for _ in range(nlev - 1)))
    att = tuple(sorted(set(rng.sample([1, 2, 4, 8][:nlev], rng.randint(1, nlev)))))
    mc = rng.choice([64, 128])
    # contract of the engine (gl_create / gl_attention reject the rest loudly): head dim a multiple of 8 and <= 160 at every
    # level, PositionNet output width == context width, 64-aligned feature widths
    heads = rng.choice([h for h in (2, 4, 8) if all((mc * m) % (8 * h) == 0 and mc * m // h <= 160 for m in mult)])
    ctx = rng.choice([64, 128])
    cfg = UNetConfig(image_size=16, model_channels=mc, num_heads=heads, channel_mult=mult, attention_resolutions=att,
                     num_res_blocks=rng.choice([1, 2]), context_dim=ctx, pos_in_dim=rng.choice([64, 128]), pos_out_dim=ctx)
    # deepest map at least 2 x 2: a 1 x 1 map with 2-channel groups turns GroupNorm into a sign function (two values per group),
    # on which ANY rounding difference from the fp32 reference is amplified -- a degenerate network, not a kernel case
    hw = rng.choice([h for h in (8, 16, 24) if h % (1 << (nlev - 1)) == 0 and h >> (nlev - 1) >= 2])
    B = rng.choice([1, 2, 3])
    nb, nr = rng.randint(0, 30), rng.randint(0, 10)
    seed = rng.randint(0, 10 ** 6)
    layout_seed = rng.randint(0, 10 ** 6)
    if os.environ.get("FUZZ_CASE"):
        import json
        cse = json.loads(os.environ["FUZZ_CASE"])
        cfg = UNetConfig(image_size=16, model_channels=cse["mc"], num_heads=cse["heads"], channel_mult=tuple(cse["mult"]),
                         attention_resolutions=tuple(cse["att"]), num_res_blocks=cse["rb"], context_dim=cse["ctx"], pos_in_dim=cse["pin"], pos_out_dim=cse["ctx"])
        hw, B, nb, nr, seed, layout_seed = cse["hw"], cse["B"], cse["boxes"], cse["rel"], cse["seed"], cse["lseed"]
        budget = 0.0
    desc = (f'{{"mc": {cfg.model_channels}, "mult": {list(cfg.channel_mult)}, "att": {list(cfg.attention_resolutions)}, "heads": {cfg.num_heads}, "rb": {cfg.num_res_blocks}, '
            f'"ctx": {cfg.context_dim}, "pin": {cfg.pos_in_dim}, "hw": {hw}, "B": {B}, "boxes": {nb}, "rel": {nr}, "seed": {seed}, "lseed": {layout_seed}}}')
    lr = random.Random(layout_seed)
    try:
        sd = recipe.state_dict(cfg, seed)
        # arithmetic parity: fp16-representable weight matrices on both sides (the engine's split 1x1-conv weights then have Wlo = 0);
        # strict mode: the fp32 weights themselves
        rw = (lambda a: a) if STRICT else (lambda a: a.astype(np.float16).astype(np.float32))
        sd = {k: (rw(np.asarray(v)) if np.asarray(v).ndim >= 2 else np.asarray(v)) for k, v in sd.items()}
        fc = recipe.sd_first_conv(cfg, seed)
        if STRICT:
            import dataclasses
            m = UNetModel(dataclasses.replace(cfg, split_weights=True), sd, device=DEV, sd_first_conv=fc)
            m.set_strict(True)
        else:
            m = UNetModel(cfg, sd, device=DEV, sd_first_conv=fc)
        m.grounding_tokenizer_input = GroundingNetInput()
        # random layouts: mostly sane boxes, sometimes degenerate / inverted / partly outside [0, 1]
        bx = []
        for _ in range(B):
            row = []
            for _ in range(nb):
                x0, y0 = lr.uniform(-0.1, 0.9), lr.uniform(-0.1, 0.9)
                w, h = lr.uniform(0.02, 0.8), lr.uniform(0.02, 0.8)
                if lr.random() < 0.05:
                    w = 0.0
                row.append((x0, y0, min(x0 + w, 1.2), min(y0 + h, 1.2)))
            bx.append(row)
        inp = {k: T(v) for k, v in recipe.synth_inputs(cfg, B, hw, n_boxes=nb, n_rel=nr, seed=seed, boxes=bx if nb else None).items()}
        rt = (lambda t_: t_) if STRICT else (lambda t_: t_.half().float())
        osd = {k: (rt(T(np.asarray(v)).float()) if np.asarray(v).ndim >= 2 else T(np.asarray(v)).float()) for k, v in sd.items()}
        ofc = {k: (rt(T(np.asarray(v)).float()) if np.asarray(v).ndim >= 2 else T(np.asarray(v)).float()) for k, v in fc.items()}
        eng = m.engine
        z = torch.zeros_like
        x = rt(inp["x"]).to(DEV)          # the oracle takes the fp16-rounded latent (xh below); strict: the fp32 latent
        tval = float(lr.choice([1, 201, 481, 981]))
        tt = torch.full((B,), int(tval), dtype=torch.long)
        xh = rt(inp["x"])
        for mode in ("cond", "null", "scale0_sd"):
            if mode == "null":
                eng.set_conditioning(inp["uc"], inp["relations"], z(inp["boxes"]), z(inp["masks"]), z(inp["positive_embeddings"]), hw)
                out = eng.forward(x, tval, 1.0, False, 1).clone()
                with torch.no_grad():
                    ref = unet_ref.unet_forward(osd, cfg, xh, tt, rt(inp["uc"]), rt(inp["relations"]), z(inp["boxes"]),
                                                z(inp["masks"]), z(inp["positive_embeddings"]))
            else:
                eng.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], hw)
                sc, sdc = (1.0, False) if mode == "cond" else (0.0, True)
                out = eng.forward(x, tval, sc, sdc, 1).clone()
                with torch.no_grad():
                    ref = unet_ref.unet_forward(osd, cfg, xh, tt, rt(inp["context"]), rt(inp["relations"]), inp["boxes"],
                                                inp["masks"], inp["positive_embeddings"], fuser_scale=sc, first_conv=ofc if sdc else None)
            n += 1
            nan_ref = torch.isnan(ref)
            if nan_ref.any():
                # a used box with an empty slice poisons the whole sample in the reference: the engine must agree sample by sample
                ok = torch.equal(torch.isnan(out.cpu()).flatten(1).all(1), nan_ref.flatten(1).all(1))
                keep = ~nan_ref.flatten(1).all(1)
                r = rel_l2(out.cpu()[keep], ref[keep]) if keep.any() else 0.0
                if not ok:
                    raise AssertionError("NaN-poisoned samples differ from the oracle's")
            else:
                if not torch.isfinite(out).all():
                    raise AssertionError("non-finite output")
                r = rel_l2(out, ref)
            worst = max(worst, r)
            if os.environ.get("FUZZ_CASE"):
                print(f"{mode}: rel_l2 {r:.3e}  max|err| {float((out.cpu() - ref).abs().max()):.3e}  |ref|max {float(ref.abs().max()):.3f}")
            if r > BOUND:
                raise AssertionError(f"{mode}: rel_l2 {r:.3e}")
        del m, eng
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL [{desc}]: {type(e).__name__}: {traceback.format_exc().strip().splitlines()[-1][:300]}", flush=True)
print(f"fuzz_engine{' (STRICT mode)' if STRICT else ''}: {n} forwards checked, {fails} failing architectures, worst rel_l2 {worst:.2e}, {time.time() - t0:.0f} s")
