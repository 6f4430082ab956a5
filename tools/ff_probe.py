"""Times gl_ff_fused against the two-launch form (GEGLU gl_gemm + output gl_gemm) on the level-0 FeedForward shapes.
usage: python tools/ff_probe.py [M ...]   (default 32768 = the 2B = 8 batch at 64x64)"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import EPI_GEGLU, EPI_RES, init_device

dev = torch.device("cuda:0")
init_device(0)
C = 320
for M in [int(a) for a in sys.argv[1:]] or [32768]:
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    x = torch.randn(M, C, device=dev, generator=g).half()
    w1 = (torch.randn(8 * C, C, device=dev, generator=g) / math.sqrt(C)).half()
    b1 = torch.randn(8 * C, device=dev, generator=g) * 0.1
    w2 = (torch.randn(C, 4 * C, device=dev, generator=g) / math.sqrt(4 * C)).half()
    b2 = torch.randn(C, device=dev, generator=g) * 0.1
    res = torch.randn(M, C, device=dev, generator=g)
    h = torch.empty(M, 4 * C, dtype=torch.float16, device=dev)
    out = torch.empty(M, C, dtype=torch.float16, device=dev)

    def two():
        ops.gemm(x, w1, h, b1, EPI_GEGLU)
        ops.gemm(h, w2, out, b2, EPI_RES, res=res)

    def fused():
        ops.ff_fused(x, w1, b1, w2, b2, res, out)

    for name, fn in (("two-launch", two), ("fused", fused), ("two-launch", two), ("fused", fused)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(f"M={M} C={C} {name:14s} {us:8.1f} us  {24.0 * M * C * C / us / 1e6:7.1f} TF/s")
