"""Third pass of the 1x1 convs (gl_set_option 45 = row threshold): forward time and error against the fp32 oracle on UNROUNDED weights per threshold (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutllm_t2i_amd import ops, recipe
from layoutllm_t2i_amd.arch import UNetConfig
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.weights import pack_state_dict, random_state_dict
from oracle import unet_ref
dev = torch.device("cuda:0")
cfg = UNetConfig()
sd = random_state_dict(cfg, dev, seed=0)
P = pack_state_dict(sd, cfg, dev, recipe.sd_first_conv(cfg, 0))
sd_cpu = {k: v.detach().float().cpu() for k, v in sd.items()}
del sd
B = 4
inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, B, 64, n_boxes=8, n_rel=3, seed=1).items()}
z = torch.zeros_like
cat = lambda a, b: torch.cat([a, b], 0)
eng = UNetEngine(P)
eng.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                     cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), 64)
x = inp["x"].to(dev)
torch.set_num_threads(32)
with torch.no_grad():
    s = lambda a: a[0:1]
    ref = unet_ref.unet_forward(sd_cpu, cfg, s(inp["x"]), torch.full((1,), 481), s(inp["context"]), s(inp["relations"]), s(inp["boxes"]), s(inp["masks"]),
                                s(inp["positive_embeddings"]))
tol = 1e-4 + 1e-3 * ref.abs()
for thr in (0, 1024, 4096, 16384):
    ops.set_option(45, thr)
    ts = {}
    for fs in (1.0, 0.0):
        o = eng.forward(x, 481.0, fs, False, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.forward(x, 481.0, fs, False, 2)
        e1.record(); torch.cuda.synchronize()
        ts[fs] = e0.elapsed_time(e1) / 10
    o = eng.forward(x, 481.0, 1.0, False, 2)[0:1].float().cpu()
    d = (o - ref)
    print(f"threshold {thr:6d}: forward {ts[1.0]:.3f} / {ts[0.0]:.3f} ms   vs fp32-weight oracle rel_l2 {float(d.norm() / ref.norm()):.3e}  outside {float((d.abs() > tol).float().mean()) * 100:.1f} %", flush=True)
