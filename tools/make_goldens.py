#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

The reference (LayoutLLM-T2I @ /root/reference) is imported read-only; its classes are filled with
recipe weights (layoutllm_t2i_amd/recipe.py), run on recipe inputs (tests/golden_cases.py), and
only the OUTPUTS are written.  Nothing from the reference is copied into the repo, and nothing
here runs on the GPU box (/root/reference does not exist there).

    python tools/make_goldens.py            # writes tests/golden/<case>.npz

interface.py cannot be imported here (omegaconf/sng_parser/clip/torchvision missing), so its two
tiny sampler callbacks (set_alpha_scale, alpha_generator; interface.py:34-75) are executed straight
from the reference file by extracting just those two function definitions with ``ast`` at run time.
"""
from __future__ import annotations

import ast
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GLIGEN_REFERENCE", "/root/reference/GLIGEN")
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np
import torch

from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import TINY, VAE_TINY
import golden_cases as gc

from ldm.modules.diffusionmodules.openaimodel import UNetModel, ResBlock, Upsample, Downsample
from ldm.modules.diffusionmodules.util import timestep_embedding, FourierEmbedder
from ldm.modules.diffusionmodules.text_grounding_net import PositionNet
from ldm.modules import attention as A
from ldm.models.diffusion.plms import PLMSSampler
from ldm.models.diffusion.ldm import LatentDiffusion
from grounding_input.text_layout_tokinzer_input import GroundingNetInput

T = torch.from_numpy


def ref_interface_fns():
    src = open(os.path.join(REF, "interface.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("set_alpha_scale", "alpha_generator")]
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "interface.py[subset]", "exec"), ns)
    return ns["set_alpha_scale"], ns["alpha_generator"]


def fill(module: torch.nn.Module, tag: str, seed: int = 0):
    sd = {}
    for name, t in module.state_dict().items():
        sd[name] = T(np.asarray(recipe.tensor(f"{tag}.{name}" if tag else name, tuple(t.shape), seed)))
    module.load_state_dict(sd, strict=True)
    return module.eval()


def tiny_unet():
    cfg = TINY
    m = UNetModel(image_size=cfg.image_size, in_channels=cfg.in_channels, model_channels=cfg.model_channels,
                  out_channels=cfg.out_channels, num_res_blocks=cfg.num_res_blocks,
                  attention_resolutions=list(cfg.attention_resolutions), channel_mult=list(cfg.channel_mult),
                  num_heads=cfg.num_heads, context_dim=cfg.context_dim, fuser_type="gatedSA",
                  grounding_tokenizer=dict(target="ldm.modules.diffusionmodules.text_grounding_net.PositionNet",
                                           params=dict(in_dim=cfg.pos_in_dim, out_dim=cfg.pos_out_dim)))
    fill(m, "", 0)
    m.grounding_tokenizer_input = GroundingNetInput()

    def restore_sd():  # reference hard-codes a 4->320 conv (openaimodel.py:401); same semantics, tiny width
        sdw = recipe.sd_first_conv(cfg, 0)
        conv = torch.nn.Conv2d(cfg.in_channels, cfg.model_channels, 3, padding=1)
        conv.load_state_dict({k: T(v) for k, v in sdw.items()})
        m.input_blocks[0][0] = conv
        m.first_conv_type = "SD"
    m.restore_first_conv_from_SD = restore_sd
    return m


@torch.no_grad()
def run_case(case):
    k, nm = case["kind"], case["name"]
    inp = {a: T(v) for a, v in gc.case_inputs(case).items()}
    tag = f"golden.{nm}"
    if k == "schedule":
        diff = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
        s = PLMSSampler(diff, model=None)
        s.make_schedule(case["S"])
        return dict(ddim_timesteps=np.asarray(s.ddim_timesteps), ddim_alphas=np.asarray(s.ddim_alphas),
                    ddim_alphas_prev=np.asarray(s.ddim_alphas_prev),
                    ddim_sqrt_one_minus_alphas=np.asarray(s.ddim_sqrt_one_minus_alphas),
                    alphas_cumprod=diff.alphas_cumprod.numpy())
    if k == "alpha_gen":
        _, ag = ref_interface_fns()
        return dict(a50=np.asarray(ag(50, [0.3, 0.0, 0.7]), np.float64), a10=np.asarray(ag(10, [0.3, 0.0, 0.7]), np.float64),
                    a20=np.asarray(ag(20, [0.5, 0.25, 0.25]), np.float64), a7=np.asarray(ag(7, None), np.float64))
    if k == "timestep_embedding":
        return dict(out=timestep_embedding(inp["t"], case["dim"]).numpy())
    if k == "fourier":
        return dict(out=FourierEmbedder(num_freqs=8)(inp["boxes"]).numpy())
    if k == "position_net":
        m = fill(PositionNet(in_dim=768, out_dim=768), "position_net", 0)
        return dict(out=m(inp["boxes"], inp["masks"], inp["positive_embeddings"]).numpy())
    if k == "res_block":
        m = fill(ResBlock(case["cin"], case["te"], 0, out_channels=case["cout"]), tag)
        return dict(out=m(inp["x"], inp["emb"]).numpy())
    if k == "self_attn":
        C, H = case["C"], case["heads"]
        m = fill(A.SelfAttention(C, heads=H, dim_head=C // H), tag)
        return dict(out=m(inp["x"]).numpy())
    if k == "cross_attn":
        C, H = case["C"], case["heads"]
        m = fill(A.CrossAttention(C, gc.CTX, gc.CTX, heads=H, dim_head=C // H), tag)
        return dict(out=m(inp["x"], inp["ctx"], inp["ctx"]).numpy())
    if k == "ff":
        m = fill(A.FeedForward(case["C"], glu=True), tag)
        return dict(out=m(inp["x"]).numpy())
    if k == "gated_sa":
        C, H = case["C"], case["heads"]
        m = fill(A.GatedSelfAttentionDense(C, gc.CTX, H, C // H), tag)
        m.scale = case["scale"]
        return dict(out=m(inp["x"], inp["objs"]).numpy())
    if k == "rela":
        C, H, hw = case["C"], case["heads"], case["hw"]
        m = fill(A.RelationCrossAttention(C, gc.CTX, gc.CTX, H, C // H), tag)
        return dict(out=m(inp["x"], inp["relations"], inp["boxes"], inp["masks"], hw, hw).numpy())
    if k == "spatial_transformer":
        C, H = case["C"], case["heads"]
        m = fill(A.SpatialTransformer(C, gc.CTX, gc.CTX, H, C // H, depth=1, fuser_type="gatedSA"), tag)
        for mod in m.modules():
            if type(mod) == A.GatedSelfAttentionDense:
                mod.scale = case["scale"]
        return dict(out=m(inp["x"], inp["context"], inp["objs"], inp["relations"], inp["boxes"], inp["masks"]).numpy())
    if k == "down":
        m = fill(Downsample(case["C"], True, dims=2, out_channels=case["C"]), tag)
        return dict(out=m(inp["x"]).numpy())
    if k == "up":
        m = fill(Upsample(case["C"], True, dims=2, out_channels=case["C"]), tag)
        return dict(out=m(inp["x"]).numpy())
    if k == "unet":
        m = tiny_unet()
        set_alpha_scale, _ = ref_interface_fns()
        set_alpha_scale(m, case["scale"])
        if case["sdconv"]:
            m.restore_first_conv_from_SD()
        batch = dict(boxes=inp["boxes"], masks=inp["masks"], text_embeddings=inp["positive_embeddings"])
        g = m.grounding_tokenizer_input.prepare(batch, None)
        d = dict(x=inp["x"], timesteps=torch.tensor(case["t"], dtype=torch.long), context=inp["context"],
                 relations=inp["relations"], inpainting_extra_input=None, grounding_extra_input=None)
        if case["grounding"] == "real":
            d["grounding_input"] = g
        else:
            d["context"] = inp["uc"]
        return dict(out=m(d).numpy())
    if k == "plms":
        from functools import partial
        m = tiny_unet()
        set_alpha_scale, alpha_generator = ref_interface_fns()
        diff = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
        sampler = PLMSSampler(diff, m, alpha_generator_func=partial(alpha_generator, type=case["alpha_type"]),
                              set_alpha_scale=set_alpha_scale)
        batch = dict(boxes=inp["boxes"], masks=inp["masks"], text_embeddings=inp["positive_embeddings"])
        g = m.grounding_tokenizer_input.prepare(batch, None)
        d = dict(x=inp["x"].clone(), timesteps=None, context=inp["context"], relations=inp["relations"],
                 grounding_input=g, inpainting_extra_input=None, grounding_extra_input=None)
        shape = (case["B"], 4, case["hw"], case["hw"])
        out = sampler.sample(S=case["S"], shape=shape, input=d, uc=inp["uc"], guidance_scale=case["guidance"])
        return dict(out=out.numpy())
    if k == "vae":
        import contextlib
        import io
        from ldm.models.autoencoder import AutoencoderKL
        cfg = VAE_TINY
        dd = dict(double_z=True, z_channels=cfg.z_channels, resolution=256, in_channels=3, out_ch=cfg.out_ch, ch=cfg.ch,
                  ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks, attn_resolutions=[], dropout=0.0)
        with contextlib.redirect_stdout(io.StringIO()):
            m = AutoencoderKL(dd, cfg.embed_dim, scale_factor=cfg.scale_factor).eval()
        sd = {n: T(np.asarray(v)) for n, v in recipe.vae_state_dict(cfg, 0).items()}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(x.startswith(("encoder.", "quant_conv.")) for x in missing), (missing[:3], unexpected)
        return dict(out=m.decode(inp["z"]).numpy())
    raise ValueError(k)


def main():
    outdir = os.path.join(REPO, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    only = set(sys.argv[1:])
    for case in gc.CASES:
        if only and case["name"] not in only:
            continue
        res = run_case(case)
        path = os.path.join(outdir, case["name"] + ".npz")
        np.savez_compressed(path, **res)
        desc = {k: (v.shape, str(v.dtype)) for k, v in res.items()}
        print(f"{case['name']:22s} -> {os.path.getsize(path)/1024:8.1f} KiB  {desc}")


if __name__ == "__main__":
    main()
