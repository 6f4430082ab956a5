"""Runs the level-0 self-attention (d=40, N=4096, B*H=64) a few times; used under rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import init_device
init_device()
DEV = "cuda:0"
d, H, N, B = int(os.environ.get("AP_D", 40)), 8, int(os.environ.get("AP_N", 4096)), 8
C = H * d
h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
q, k, v = h(B, N, C), h(B, N, C), h(B, N, C)
vt = torch.empty(B, H, d, N, dtype=torch.float16, device=DEV)
ops.transpose_v(v, N * C, C, vt, B, H, d, N)
out = torch.empty(B, N, C, dtype=torch.float16, device=DEV)
for _ in range(int(os.environ.get("AP_ITERS", 3))):
    ops.attention(q, N * C, C, k, N * C, C, vt, out, N * C, C, B, H, d, N, N, d ** -0.5)
torch.cuda.synchronize()
