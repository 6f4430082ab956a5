#!/usr/bin/env python3
"""Spins ONE three-pass conv / GEMM shape for PMC passes and timing (strict operand forms; env KB_OPTS=52=0 selects the K-walk).
    python tools/s3_probe.py conv side cin cout [B2]      |      python tools/s3_probe.py gemm M N K [res|bias|geglu]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import EPI_BIAS, EPI_GEGLU, EPI_RES, init_device
init_device()
for kv in filter(None, os.environ.get("KB_OPTS", "").split(",")):
    k, v = kv.split("=")
    ops.set_option(int(k), int(v))
DEV = "cuda:0"
h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
a = sys.argv[1:] or ["conv", "64", "320", "320"]
iters = int(os.environ.get("ITERS", "20"))
if a[0] == "conv":
    side, cin, cout = int(a[1]), int(a[2]), int(a[3])
    B2 = int(a[4]) if len(a) > 4 else 8
    x, w = h(B2 * side * side, 2 * cin), h(cout, 18 * cin) * ((9 * cin) ** -0.5)
    o = torch.empty(B2 * side * side, cout, dtype=torch.float32, device=DEV)
    bias = torch.zeros(cout, device=DEV)
    fn = lambda: ops.conv3x3(x, w, o, B2, side, side, bias, in_split=3, w_split=True)
    fl = 2.0 * B2 * side * side * cout * 9 * cin
else:
    M, N, K = int(a[1]), int(a[2]), int(a[3])
    epi = a[4] if len(a) > 4 else "res"
    x, w, bias = h(M, 2 * K), h(N, 2 * K) * (K ** -0.5), torch.zeros(N, device=DEV)
    if epi == "geglu":
        o = torch.empty(M, N, dtype=torch.float16, device=DEV)
        fn = lambda: ops.gemm(x, w, o, bias, EPI_GEGLU, hilo_a=True, wsplit=2, hilo_out=True)
    elif epi == "res":
        o, r = torch.empty(M, N, dtype=torch.float32, device=DEV), torch.randn(M, N, device=DEV)
        fn = lambda: ops.gemm(x, w, o, bias, EPI_RES, res=r, hilo_a=True, wsplit=2)
    else:
        o = torch.empty(M, 2 * N, dtype=torch.float16, device=DEV)
        fn = lambda: ops.gemm(x, w, o, bias, hilo_a=True, wsplit=2, hilo_out=True)
    fl = 2.0 * M * N * K
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
print(f"{' '.join(a)}: {t * 1e6:.1f} us, {fl / t / 1e12:.0f} TF/s algorithmic, {3 * fl / t / 1e12:.0f} TF/s issued")
