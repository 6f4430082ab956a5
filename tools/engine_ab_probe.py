"""Same-box, same-process A/B on the GPU box: forward time of the C++ engine (gl_unet_forward, hipGraph replay) on the full
model under different gl_set_option settings, interleaved rounds; optionally also the Python launch sequence of
tests/engine_pyref.py (torch-captured graph) as a cross-check of the orchestration cost.
    python tools/engine_ab_probe.py [B] [variant ...]      variant = name:key=value[,key=value...]   e.g.  nofusedgn:17=0
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("GL_PROBE_LIB"):          # A/B against another build of the library (a different process per build)
    import layoutllm_t2i_amd._lib as _L
    _L.LIB_PATH = os.environ["GL_PROBE_LIB"]
from layoutllm_t2i_amd import ops, recipe
from layoutllm_t2i_amd.arch import UNetConfig
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.weights import pack_state_dict, random_state_dict

args = sys.argv[1:]
B = int(args[0]) if args and args[0].isdigit() else 4
variants = [("default", [])]
for v in args[1:] if args and args[0].isdigit() else args:
    if v == "pyref":
        variants.append(("pyref", None))
        continue
    name, kv = v.split(":")
    variants.append((name, [tuple(int(t) for t in p.split("=")) for p in kv.split(",")]))
dev = torch.device("cuda:0")
cfg = UNetConfig()
_sd = random_state_dict(cfg, dev, seed=0)
P = pack_state_dict(_sd, cfg, dev, recipe.sd_first_conv(cfg, 0))
# AB_PARITY=1: every variant's output (sample 0, cond, fuser on) against the fp32 oracle on the same UNROUNDED weights / latent / context
PARITY = bool(int(os.environ.get("AB_PARITY", "0")))
sd_cpu = {k: v.detach().float().cpu() for k, v in _sd.items()} if PARITY else None
del _sd
inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, B, 64, n_boxes=8, n_rel=3, seed=1).items()}
z = torch.zeros_like
cat = lambda a, b: torch.cat([a, b], 0)
eng = UNetEngine(P)
engines = {"cpp": eng}
if any(o is None for _, o in variants):
    from engine_pyref import PyRefEngine
    engines["pyref"] = PyRefEngine(P)
x = inp["x"].to(dev)


def condition(e):
    e.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                       cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), 64)


for e in engines.values():
    condition(e)
DEFAULTS = {17: 1, 21: 1, 13: 3, 7: 300, 8: 1, 5: -1, 6: 16, 3: 0, 10: -1, 2: 0, 4: 400, 16: 16, 23: 1, 24: 64, 25: 1, 27: 1, 29: 1, 30: 1, 31: 200, 33: 0, 34: 11, 35: 5, 37: 1, 38: 1, 41: 1, 42: 1, 43: 1, 44: 1, 45: 1024, 46: 11, 47: 100}
ref = None
if PARITY:
    from oracle import unet_ref
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = unet_ref.unet_forward(sd_cpu, cfg, inp["x"][0:1], torch.tensor([481]), inp["context"][0:1], inp["relations"][0:1], inp["boxes"][0:1],
                                    inp["masks"][0:1], inp["positive_embeddings"][0:1], fuser_scale=1.0)
parity = {}
res = {n: {1.0: [], 0.0: []} for n, _ in variants}
launches = {}
for rnd in range(5):
    for name, opts in variants:
        e = engines["pyref"] if opts is None else eng
        for k, v in (opts or []):
            ops.set_option(k, v)
        if opts is not None:
            condition(e)                 # some options act when the conditioning is set (key 43: rows of the relation chain)
        for fs in (1.0, 0.0):
            e.forward(x, 481.0, fs, False, 2)          # (re-)capture under these options
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                e.forward(x, 481.0, fs, False, 2)
            e1.record()
            torch.cuda.synchronize()
            res[name][fs].append(e0.elapsed_time(e1) / 10)
        if opts is not None:
            launches[name] = eng.num_launches()
        if ref is not None and name not in parity:
            o = e.forward(x, 481.0, 1.0, False, 2)[0:1].float().cpu()
            d = (o - ref).abs()
            parity[name] = (float((o - ref).norm() / ref.norm()), float((d > 1e-4 + 1e-3 * ref.abs()).float().mean()))
        for k, _ in (opts or []):
            ops.set_option(k, DEFAULTS[k])
for name, _ in variants:
    for fs in (1.0, 0.0):
        v = sorted(res[name][fs])
        print(f"{name:12s} fuser={'on ' if fs else 'off'} 2B={2 * B}: median {v[len(v) // 2]:.3f} ms  min {v[0]:.3f}  all {[round(t, 2) for t in res[name][fs]]}")
print("launches per fuser-off forward:", launches)
for name, (r, f) in parity.items():
    print(f"{name:12s} vs fp32 oracle (unrounded weights, sample 0 cond fuser on): rel_l2 {r:.3e}, outside rtol 1e-3 / atol 1e-4: {100 * f:.1f} %")
