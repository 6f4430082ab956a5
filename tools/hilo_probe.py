"""Dispatch A/B for the split-fp16 1x1 products (A = [hi | lo], K = 2C, one weight copy via kwrap) at the engine's shapes:
time per shape under gl_set_option variants.   python tools/hilo_probe.py [name:key=value,...] ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import EPI_BIAS, EPI_RES, init_device

init_device()
DEV = "cuda:0"
DEFAULTS = {5: -1, 30: 1, 31: 200, 4: 400, 7: 300, 13: 3, 34: 11, 35: 5, 37: 1, 6: 16}
variants = [("default", [])]
for v in sys.argv[1:]:
    name, kv = v.split(":")
    variants.append((name, [tuple(int(t) for t in p.split("=")) for p in kv.split(",")]))
# (M, C_out, C_in, epilogue): proj_in (bias), proj_out (res, fp32 stream), skip convs (bias, K = 2 * concat width)
shapes = [(32768, 320, 320, "bias"), (32768, 320, 320, "res"), (8192, 640, 640, "bias"), (8192, 640, 640, "res"), (2048, 1280, 1280, "bias"),
          (2048, 1280, 1280, "res"), (512, 1280, 1280, "bias"), (512, 1280, 1280, "res"),
          (8192, 640, 320, "bias"), (2048, 1280, 640, "bias"), (512, 1280, 2560, "bias"), (2048, 1280, 2560, "bias"), (2048, 1280, 1920, "bias"),
          (8192, 640, 1920, "bias"), (8192, 640, 1280, "bias"), (8192, 640, 960, "bias"), (32768, 320, 960, "bias"), (32768, 320, 640, "bias")]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f"{'M, N, C_in, epi (K = 2 C_in)':34s} " + " ".join(f"{n:>12s}" for n, _ in variants))
tot = [0.0] * len(variants)
for M, N, Cin, epi in shapes:
    a = torch.randn(M, 2 * Cin, device=DEV).half()
    w = (torch.randn(N, Cin, device=DEV) * Cin ** -0.5).half()
    b = torch.zeros(N, device=DEV)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    r = torch.randn(M, N, device=DEV)
    row = []
    for vi, (name, opts) in enumerate(variants):
        for k, v in opts:
            ops.set_option(k, v)
        fn = (lambda: ops.gemm(a, w, out, b, EPI_RES, res=r, hilo_a=True)) if epi == "res" else (lambda: ops.gemm(a, w, out, b, EPI_BIAS, hilo_a=True))
        t = timeit(fn)
        row.append(t)
        tot[vi] += t
        for k, _ in opts:
            ops.set_option(k, DEFAULTS[k])
    print(f"{str((M, N, Cin, epi)):34s} " + " ".join(f"{t:12.1f}" for t in row))
print(f"{'sum (one launch each), us':34s} " + " ".join(f"{t:12.1f}" for t in tot))
