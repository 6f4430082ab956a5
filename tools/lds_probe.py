"""Runs the level-0 and level-1 3x3 convs with the 4 x (32 x 160) kernel (option 13 = 0) and with the intra-block K-split
kernel (option 13 = 2); used under `rocprofv3 --pmc ...` to compare LDS activity per kernel name."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import init_device
init_device()
DEV = "cuda:0"
h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
B2 = 8
cases = [(64, 960, 320), (64, 320, 320), (32, 1920, 640)]
for side, cin, cout in cases:
    x, w = h(B2 * side * side, cin), h(cout, 9 * cin) * 0.02
    o = torch.empty(B2 * side * side, cout, dtype=torch.float16, device=DEV)
    b = torch.zeros(cout, device=DEV)
    for opt in (0, 2):
        ops.set_option(13, opt)
        for _ in range(10):
            ops.conv3x3(x, w, o, B2, side, side, b)
    torch.cuda.synchronize()
ops.set_option(13, 1)
