"""Runs the hot GEMM / conv / attention shapes of the 2B = 8 forward a few times each; used under `rocprofv3 --pmc ...` to
read per-kernel counters (tools/pmc_round.sh).  Shapes: GEGLU FF1 at L0 / L1 / L2, FF2 at L0, the K = 320 residual
projection, QKV at L0, the 3x3 convs at L0 / L1 / L2, self-attention at L0 / L1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import EPI_GEGLU, EPI_RES, init_device

init_device()
DEV = "cuda:0"
h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
B2 = 8
REP = int(os.environ.get("PROBE_REP", "5"))
which = set(sys.argv[1:]) or {"geglu", "ff2", "res", "qkv", "conv", "attn"}
jobs = []
if "geglu" in which:
    for M, C in ((32768, 320), (8192, 640), (2048, 1280)):
        a, w, b = h(M, C), h(8 * C, C) * C ** -0.5, torch.zeros(8 * C, device=DEV)
        o = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
        jobs.append(lambda a=a, w=w, b=b, o=o: ops.gemm(a, w, o, b, EPI_GEGLU))
if "ff2" in which:
    a, w, b = h(32768, 1280), h(320, 1280) * 1280 ** -0.5, torch.zeros(320, device=DEV)
    r = torch.randn(32768, 320, device=DEV)
    o = torch.empty(32768, 320, dtype=torch.float32, device=DEV)
    jobs.append(lambda a=a, w=w, b=b, o=o, r=r: ops.gemm(a, w, o, b, EPI_RES, res=r))
if "res" in which:
    a, w, b = h(32768, 320), h(320, 320) * 320 ** -0.5, torch.zeros(320, device=DEV)
    r = torch.randn(32768, 320, device=DEV)
    o = torch.empty(32768, 320, dtype=torch.float32, device=DEV)
    jobs.append(lambda a=a, w=w, b=b, o=o, r=r: ops.gemm(a, w, o, b, EPI_RES, res=r))
if "qkv" in which:
    a, w = h(32768, 320), h(960, 320) * 320 ** -0.5
    o = torch.empty(32768, 960, dtype=torch.float16, device=DEV)
    jobs.append(lambda a=a, w=w, o=o: ops.gemm(a, w, o))
if "conv" in which:
    for side, c in ((64, 320), (32, 640), (16, 1280)):
        x, w, b = h(B2 * side * side, c), h(c, 9 * c) * (9 * c) ** -0.5, torch.zeros(c, device=DEV)
        o = torch.empty(B2 * side * side, c, dtype=torch.float16, device=DEV)
        jobs.append(lambda x=x, w=w, b=b, o=o, side=side: ops.conv3x3(x, w, o, B2, side, side, b))
if "attn" in which:
    for d, N in ((40, 4096), (80, 1024)):
        C = 8 * d
        q, k, v = h(B2, N, C), h(B2, N, C), h(B2, N, C)
        vt = torch.empty(B2, 8, d, ops.vt_ld(N), dtype=torch.float16, device=DEV)
        ops.transpose_v(v, N * C, C, vt, B2, 8, d, N)
        o = torch.empty(B2, N, C, dtype=torch.float16, device=DEV)
        jobs.append(lambda q=q, k=k, vt=vt, o=o, d=d, N=N, C=C: ops.attention(q, N * C, C, k, N * C, C, vt, o, N * C, C, B2, 8, d, N, N, d ** -0.5,
                                                                                q_prescaled=bool(int(os.environ.get("PROBE_PRE", "1")))))   # the engine's form
for j in jobs:
    for _ in range(REP):
        j()
torch.cuda.synchronize()
