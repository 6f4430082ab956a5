"""10 strict-mode forwards of the 2B = 8 batch (fuser off) for a rocprofv3 --kernel-trace --stats breakdown of the strict mode"""
import os, sys, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import UNetConfig
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.weights import pack_state_dict, random_state_dict
dev = torch.device("cuda:0")
cfg = dataclasses.replace(UNetConfig(), split_weights=True)
P = pack_state_dict(random_state_dict(cfg, dev, seed=0), cfg, dev, recipe.sd_first_conv(cfg, 0))
B = 4
inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, B, 64, n_boxes=8, n_rel=3, seed=1).items()}
z = torch.zeros_like
cat = lambda a, b: torch.cat([a, b], 0)
eng = UNetEngine(P)
eng.set_option(50, 1)
eng.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                     cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), 64)
x = inp["x"].to(dev)
fs = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for _ in range(11):
    eng.forward(x, 481.0, fs, False, 2)
torch.cuda.synchronize()
