import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import EPI_BIAS, init_device
init_device()
DEV = "cuda:0"
B = 4
shapes = [(64, 512, 512, 0), (64, 512, 512, 1), (128, 512, 512, 0), (128, 512, 512, 1), (256, 512, 256, 0), (256, 256, 256, 0), (256, 256, 256, 1),
          (512, 256, 128, 0), (512, 128, 128, 0)]
for side, cin, cout, up in shapes:
    x = torch.randn(B * side * side, cin, device=DEV, dtype=torch.float16)
    w = (torch.randn(cout, 9 * cin, device=DEV) * (9 * cin) ** -0.5).to(torch.float16)
    b = torch.randn(cout, device=DEV) * 0.1
    so = side * 2 if up else side
    out = torch.empty(B * so * so, cout, device=DEV, dtype=torch.float16)
    c0 = ops.gemm8_launch_count()
    ops.conv3x3(x, w, out, B, side, side, b, upsample2x=bool(up))
    c1 = ops.gemm8_launch_count()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv3x3(x, w, out, B, side, side, b, upsample2x=bool(up))
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e3
    fl = 2.0 * B * so * so * cout * 9 * cin
    print(f"side {side} {cin}->{cout} up={up}: g8={c1-c0} {t:8.1f} us {fl/t*1e-6:7.0f} TF/s", flush=True)
