"""strict residual on the full UNet: batch / reps / shared-prefix effects"""
import sys, os, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import UNetConfig
from layoutllm_t2i_amd.model import UNetModel
from layoutllm_t2i_amd.weights import random_state_dict
from oracle import unet_ref
DEV = "cuda:0"
cfg = dataclasses.replace(UNetConfig(), split_weights=True)
sd = random_state_dict(cfg, torch.device(DEV), seed=3)
m = UNetModel(cfg, sd, device=DEV, allow_missing_sd_conv=True)
osd = {k: v.detach().float().cpu() for k, v in sd.items()}
del sd
eng = m.engine
torch.set_num_threads(32)
hw, B = 64, 4
inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, B, hw, n_boxes=8, n_rel=3, seed=2024).items()}
refs = {}
for k in (0, 1):
    s = lambda a: a[k:k + 1]
    with torch.no_grad():
        refs[k] = unet_ref.unet_forward(osd, cfg, s(inp["x"]), torch.tensor([481]), s(inp["context"]), s(inp["relations"]), s(inp["boxes"]), s(inp["masks"]), s(inp["positive_embeddings"]))
z = torch.zeros_like
cat = lambda a, b: torch.cat([a, b], 0)
two = dict(context=cat(inp["context"], inp["uc"]), relations=cat(inp["relations"], inp["relations"]), boxes=cat(inp["boxes"], z(inp["boxes"])),
           masks=cat(inp["masks"], z(inp["masks"])), positive_embeddings=cat(inp["positive_embeddings"], z(inp["positive_embeddings"])))
def show(tag, out):
    for k in (0, 1):
        d = out[k:k + 1].float().cpu() - refs[k]
        print(f"{tag} sample {k}: rel_l2={float(d.norm() / refs[k].norm()):.3e} max|err|={float(d.abs().max()):.2e}", flush=True)
for opts in ({}, {44: 0}):
    eng.clear_options()
    for k_, v_ in opts.items():
        eng.set_option(k_, v_)
    eng.set_option(50, 1)
    eng.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], hw)
    show(f"B=4 reps=1 opts={opts}", eng.forward(inp["x"].to(DEV), 481.0, 1.0, False, 1))
    eng.set_conditioning(two["context"], two["relations"], two["boxes"], two["masks"], two["positive_embeddings"], hw)
    show(f"2B=8 reps=2 opts={opts}", eng.forward(inp["x"].to(DEV), 481.0, 1.0, False, 2))
