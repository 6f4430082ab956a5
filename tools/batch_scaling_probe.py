"""UNet forward time (one graph replay) vs conditioning batch Bn: separates the per-launch fixed cost from the per-sample cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import UNetConfig
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.weights import pack_state_dict, random_state_dict

dev = torch.device("cuda:0")
cfg = UNetConfig()
sd = random_state_dict(cfg, dev, seed=0)
packed = pack_state_dict(sd, cfg, dev, None)
del sd
for scale in (1.0, 0.0):
    pts = []
    for Bn in (1, 2, 4, 8, 16, 32):
        inp = {k: torch.from_numpy(v).to(dev) for k, v in recipe.synth_inputs(cfg, Bn, 64, n_boxes=8, n_rel=3, seed=1).items()}
        e = UNetEngine(packed)
        e.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], 64)
        x = torch.randn(Bn, 4, 64, 64, device=dev)
        for _ in range(3):
            e.forward(x, 500.0, scale)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 20 if Bn <= 8 else 8
        for _ in range(n):
            e.forward(x, 500.0, scale)
        torch.cuda.synchronize()
        pts.append((Bn, (time.time() - t0) / n * 1e3))
        del e
        torch.cuda.empty_cache()
    print(f"fuser_scale {scale}: " + "  ".join(f"Bn={b}: {t:.2f} ms" for b, t in pts))
