"""time of the split-fp16 attention kernel on the UNet's shapes (2B = 8)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import init_device
init_device()
DEV = "cuda:0"
B2, H = 8, 8
h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
for d, Nq, Nk in ((40, 4096, 4096), (40, 4096, 77), (80, 1024, 1024), (80, 1024, 77), (160, 256, 256), (160, 64, 64)):
    C = H * d
    q, ql, k, kl = h(B2, Nq, C) * 0.4, h(B2, Nq, C) * 1e-4, h(B2, Nk, C), h(B2, Nk, C) * 1e-4
    v, vl = h(B2, Nk, C), h(B2, Nk, C) * 1e-4
    vt = torch.empty(B2, H, d, ops.vt_ld(Nk), dtype=torch.float16, device=DEV)
    vtl = torch.empty_like(vt)
    ops.transpose_v(v, Nk * C, C, vt, B2, H, d, Nk)
    ops.transpose_v(vl, Nk * C, C, vtl, B2, H, d, Nk)
    o = torch.empty(B2, Nq, 2 * C, dtype=torch.float16, device=DEV)
    fn = lambda: ops.attention(q, Nq * C, C, k, Nk * C, C, vt, o, Nq * 2 * C, 2 * C, B2, H, d, Nq, Nk, d ** -0.5, q_prescaled=True, q_lo=ql, k_lo=kl, vt_lo=vtl, out_lo=o[:, :, C:])
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    print(f"split attention d={d} Nq={Nq} Nk={Nk}: {best:8.1f} us   ({3 * 4.0 * B2 * H * Nq * Nk * d / best / 1e6:6.1f} TF/s of issued MFMA work)", flush=True)
