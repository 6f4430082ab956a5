"""Same-box A/B on the GPU box: forward time of the C++ engine (gl_unet_forward, engine-owned hipGraph) vs the Python
launch sequence of tests/engine_pyref.py (torch-captured graph) on the full model, interleaved rounds.
    python tools/engine_ab_probe.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_pyref import PyRefEngine
from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import UNetConfig
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.weights import pack_state_dict, random_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
cfg = UNetConfig()
P = pack_state_dict(random_state_dict(cfg, dev, seed=0), cfg, dev, recipe.sd_first_conv(cfg, 0))
inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, B, 64, n_boxes=8, n_rel=3, seed=1).items()}
z = torch.zeros_like
cat = lambda a, b: torch.cat([a, b], 0)
engines = {"cpp": UNetEngine(P), "pyref": PyRefEngine(P)}
x = inp["x"].to(dev)
for e in engines.values():
    e.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                       cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), 64)
    for fs in (1.0, 0.0):
        e.forward(x, 481.0, fs, False, 2)
torch.cuda.synchronize()
res = {k: {1.0: [], 0.0: []} for k in engines}
for rnd in range(6):
    for name, e in engines.items():
        for fs in (1.0, 0.0):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                e.forward(x, 481.0, fs, False, 2)
            e1.record()
            torch.cuda.synchronize()
            res[name][fs].append(e0.elapsed_time(e1) / 10)
for name in engines:
    for fs in (1.0, 0.0):
        v = sorted(res[name][fs])
        print(f"{name:6s} fuser={'on ' if fs else 'off'} 2B={2 * B}: median {v[len(v) // 2]:.3f} ms  min {v[0]:.3f}  all {[round(t, 2) for t in res[name][fs]]}")
print("launches per forward (cpp, last variant run):", engines["cpp"].num_launches())
