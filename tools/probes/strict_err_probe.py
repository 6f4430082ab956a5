"""which part of the strict forward carries its residual error?  tiny UNet at growing latent sides (tokens per attention), strict vs the fp32 oracle"""
import sys, os, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import TINY
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.weights import pack_state_dict
from oracle import unet_ref
DEV = "cuda:0"
cfg = dataclasses.replace(TINY, split_weights=True)
sd = recipe.state_dict(TINY, 0)
eng = UNetEngine(pack_state_dict(sd, cfg, DEV, recipe.sd_first_conv(TINY, 0)))
osd = {k: torch.from_numpy(np.asarray(v)).float() for k, v in sd.items()}
torch.set_num_threads(32)
for hw in (16, 32, 64):
    inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(TINY, 1, hw, n_boxes=4, seed=99).items()}
    eng.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], hw)
    with torch.no_grad():
        ref = unet_ref.unet_forward(osd, TINY, inp["x"], torch.tensor([481]), inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"])
    for opts in ({}, {51: 0}):
        eng.clear_options()
        eng.set_option(50, 1)
        for k_, v_ in opts.items():
            eng.set_option(k_, v_)
        out = eng.forward(inp["x"].to(DEV), 481.0, 1.0, False, 1).float().cpu()
        d = out - ref
        print(f"hw={hw} opts={opts}: rel_l2={float(d.norm() / ref.norm()):.3e} max|err|={float(d.abs().max()):.2e} |ref|rms={float(ref.pow(2).mean().sqrt()):.3f}", flush=True)
