"""bisect the strict mode's residual error on the FULL UNet (2B = 2) over dispatch options"""
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
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, B, hw, n_boxes=8, n_rel=3, seed=2024).items()}
s = lambda a: a[:1]
with torch.no_grad():
    ref = unet_ref.unet_forward(osd, cfg, s(inp["x"]), torch.tensor([481]), s(inp["context"]), s(inp["relations"]), s(inp["boxes"]), s(inp["masks"]), s(inp["positive_embeddings"]))
for opts in ({}, {30: 0}, {13: 0}, {17: 0}, {5: 0}, {46: 0}, {24: 0}, {3: 4}, {23: 0}):
    eng.clear_options()
    for k_, v_ in opts.items():
        eng.set_option(k_, v_)
    eng.set_option(50, 1)
    eng.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], hw)
    out = eng.forward(inp["x"].to(DEV), 481.0, 1.0, False, 1)[:1].float().cpu()
    d = out - ref
    print(f"hw={hw} B={B} opts={opts}: rel_l2={float(d.norm() / ref.norm()):.3e} max|err|={float(d.abs().max()):.2e} launches {eng.num_launches()}", flush=True)
