"""Runs a few representative GEMM / conv shapes; used under rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import init_device
init_device()
DEV = "cuda:0"
h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
B2 = 8
# conv L0 320->320 (128x160 tiles), conv L1 640->640, gemm L1 qkv (128x128), gemm L0 ff2
x0, w0 = h(B2 * 64 * 64, 320), h(320, 9 * 320) * 0.02
o0 = torch.empty(B2 * 64 * 64, 320, dtype=torch.float16, device=DEV)
x1, w1 = h(B2 * 32 * 32, 640), h(640, 9 * 640) * 0.02
o1 = torch.empty(B2 * 32 * 32, 640, dtype=torch.float16, device=DEV)
a2, w2 = h(8192, 640), h(1920, 640) * 0.04
o2 = torch.empty(8192, 1920, dtype=torch.float16, device=DEV)
a3, w3 = h(32768, 1280), h(320, 1280) * 0.03
o3 = torch.empty(32768, 320, dtype=torch.float16, device=DEV)
for _ in range(3):
    ops.conv3x3(x0, w0, o0, B2, 64, 64)
    ops.conv3x3(x1, w1, o1, B2, 32, 32)
    ops.gemm(a2, w2, o2)
    ops.gemm(a3, w3, o3)
torch.cuda.synchronize()
