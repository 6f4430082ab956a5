"""Under `rocprofv3 --kernel-trace`: a few graph replays of the full-model 2B = 8 forward (fuser on, then off).
tools/gap_report.py turns the trace into: span of one replay, sum of kernel durations, idle gaps between kernels."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import UNetConfig
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.weights import pack_state_dict, random_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
from layoutllm_t2i_amd import ops
for kv in sys.argv[2:]:                      # gl_set_option overrides for an A/B timeline: key=value ...
    k, v = kv.split("=")
    ops.set_option(int(k), int(v))
dev = torch.device("cuda:0")
cfg = UNetConfig()
P = pack_state_dict(random_state_dict(cfg, dev, seed=0), cfg, dev, recipe.sd_first_conv(cfg, 0))
inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, B, 64, n_boxes=8, n_rel=3, seed=1).items()}
z = torch.zeros_like
cat = lambda a, b: torch.cat([a, b], 0)
e = UNetEngine(P)
x = inp["x"].to(dev)
e.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                   cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), 64)
for fs in (1.0, 0.0):
    for _ in range(4):
        e.forward(x, 481.0, fs, False, 2)
    torch.cuda.synchronize()
