#!/usr/bin/env python3
"""Spins the split-fp16 attention of one shape (default: level 0 of configs[1], d = 40, 4096 x 4096, 2B = 8) for PMC passes / timing.
    python tools/attn_split_probe.py [d Nq Nk B iters]      env KB_OPTS=53=2 selects the round-5 kernel"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import init_device
init_device()
for kv in filter(None, os.environ.get("KB_OPTS", "").split(",")):
    k, v = kv.split("=")
    ops.set_option(int(k), int(v))
a = [int(x) for x in sys.argv[1:]]
d, Nq, Nk, B, iters = (a + [40, 4096, 4096, 8, 10][len(a):])[:5]
H, C = 8, 8 * d
DEV = "cuda:0"
h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
q, k, v = h(B, Nq, C), h(B, Nk, C), h(B, Nk, C)
vt = torch.empty(B, H, d, ops.vt_ld(Nk), dtype=torch.float16, device=DEV)
ops.transpose_v(v, Nk * C, C, vt, B, H, d, Nk)
ql, kl, vtl = q * 1e-3, k * 1e-3, vt * 1e-3
out, outl = torch.empty(B, Nq, C, dtype=torch.float16, device=DEV), torch.empty(B, Nq, C, dtype=torch.float16, device=DEV)
fn = lambda: ops.attention(q, Nq * C, C, k, Nk * C, C, vt, out, Nq * C, C, B, H, d, Nq, Nk, d ** -0.5, q_prescaled=True, q_lo=ql, k_lo=kl, vt_lo=vtl, out_lo=outl)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / iters * 1e-3
print(f"split attention d={d} Nq={Nq} Nk={Nk} B={B}: {t * 1e6:.1f} us, {3 * 4.0 * B * H * Nq * Nk * d / t / 1e12:.0f} TF/s of issued three-pass work")
