"""Times one GEMM shape with different epilogues, with and without the main loop (gl_set_option 12)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import init_device, EPI_BIAS, EPI_GEGLU, EPI_RES
init_device()
DEV = "cuda:0"
h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(32768, 2560, 320), (32768, 320, 320), (8192, 5120, 640)]
for M, N, K in shapes:
    a, w = h(M, K), h(N, K) * 0.05
    bias = torch.randn(N, device=DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    outg = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
    res = h(M, N)
    for dbg in (0, 3):
        ops.set_option(12, dbg)
        t_none = timeit(lambda: ops.gemm(a, w, out))
        t_bias = timeit(lambda: ops.gemm(a, w, out, bias))
        t_res = timeit(lambda: ops.gemm(a, w, out, bias, EPI_RES, res=res))
        t_geglu = timeit(lambda: ops.gemm(a, w, outg, bias, EPI_GEGLU)) if N % 64 == 0 else float("nan")
        print(f"{(M, N, K)} dbg={dbg}: none {t_none:7.1f}  bias {t_bias:7.1f}  res {t_res:7.1f}  geglu {t_geglu:7.1f} us")
    ops.set_option(12, 0)
