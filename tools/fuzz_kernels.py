"""Randomised shape sweep over the C-ABI kernels (GPU box): calls the parametrised checks of tests/test_gpu_kernels.py with
random -- ragged, boundary-straddling -- shapes instead of the fixed ones, for a time budget.  Every check compares the
HIP kernel with torch fp32 on the same fp16-rounded operands (and, where a kernel has two dispatch forms, the forms with
each other).  usage: python tools/fuzz_kernels.py [seconds] [seed]; prints one line per failure and a summary."""
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_kernels as T  # noqa: E402
import test_gpu_strict as S  # noqa: E402
STRICT = bool(int(os.environ.get("FUZZ_STRICT", "0")))      # FUZZ_STRICT=1: only the strict-mode kernels (three-pass loop, split attention, [hi | lo] V^T tail)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
from layoutllm_t2i_amd._lib import init_device  # noqa: E402
init_device(0)
# FUZZ_OPTS="30=2,34=2": gl_set_option knobs for the sweep (e.g. force the 8-wave GEMM / conv kernel wherever it applies)
from layoutllm_t2i_amd import ops as _fz_ops  # noqa: E402
for _kv in filter(None, os.environ.get("FUZZ_OPTS", "").split(",")):
    _fz_ops.set_option(int(_kv.split("=")[0]), int(_kv.split("=")[1]))


def c64(lo, hi):
    return 64 * rng.randint(lo // 64, hi // 64)


def strict_case():
    """round 6: the three-pass main loop against the K-walk and fp64 (GEMM and conv), the split-fp16 attention kernels (round-5 kernel below 512
    queries, the software-pipelined one from 512 on at d = 32 / 40 / 48), the [hi | lo] QKV projection's transposed tail"""
    k = rng.choice(["gemm3", "gemm3", "conv3", "attn_s", "attn_s", "attn_pipe", "attn_pipe", "vt2"])
    if k == "gemm3":
        return S.test_gemm_three_pass_loop_vs_kwalk, (rng.randint(1, 6000), 32 * rng.randint(1, 60), c64(64, 2560), rng.choice(["res", "hilo"]))
    if k == "conv3":
        mode = rng.choice(["s1", "s1", "s2", "up"])
        hw = rng.choice([12, 16, 20, 24, 32]) if mode != "s2" else rng.choice([24, 32, 40])
        return S.test_conv_three_pass_loop_vs_kwalk, (mode, c64(64, 1280), 32 * rng.randint(1, 24), hw, rng.choice([2, 3, 4]))
    if k == "attn_s":
        d = rng.choice([8, 16, 24, 32, 40, 48, 64, 80, 128, 160])
        return S.test_split_attention, (d, rng.choice([1, 2, 4]), rng.randint(2, 500), rng.randint(1, 900), rng.choice([1, 2]))
    if k == "attn_pipe":
        return S.test_split_attention, (rng.choice([32, 40, 48]), rng.choice([1, 2, 4]), rng.randint(512, 2200), rng.randint(1, 2300), 1)
    H, d = rng.choice([(8, 40), (8, 80), (4, 48), (8, 24), (8, 160), (2, 32)])
    return S.test_qkv_projection_writes_both_vt_halves, (rng.choice([1, 2, 3]), rng.randint(8, 900), H * d, d)


def case():
    if STRICT:
        return strict_case()
    k = rng.choice(["gemm", "gemm", "skinny", "res32", "conv", "attn", "attn_pre", "gn", "gn1", "gn32", "hilo_a", "hilo_out", "ln", "ff", "qkv", "merge_ln"])
    if k == "gemm":
        return T.test_gemm_bias, (rng.randint(1, 3000), 8 * rng.randint(1, 200), c64(64, 1536))
    if k == "skinny":
        return T.test_gemm_skinny, (rng.randint(1, 1024), 32 * rng.randint(1, 48), c64(64, 2560), rng.choice(["bias", "silu", "res", "gate", "res32"]))
    if k == "res32":
        return T.test_gemm_residual_stream_fp32, (rng.randint(1, 2500), 8 * rng.randint(4, 170), c64(64, 2560))
    if k == "conv":
        mode = rng.choice(["s1", "s2", "up"])
        hw = rng.choice([4, 6, 8, 10, 12, 16, 24]) if mode != "up" else rng.choice([4, 6, 8, 12])
        return T.test_conv3x3, (mode, c64(64, 640), 32 * rng.randint(1, 20), hw)
    if k in ("attn", "attn_pre"):
        d = rng.choice([8, 16, 24, 32, 40, 48, 64, 80, 96, 128, 160])
        H = rng.choice([1, 2, 4, 8])
        Nq, Nk = rng.randint(2, 1300), rng.randint(1, 1400)
        return (T.test_attention if k == "attn" else T.test_attention_prescaled_q), (d, H, Nq, Nk, rng.choice([1, 2]))
    if k == "gn":
        C1 = 32 * rng.choice([2, 4, 8, 10, 20, 30, 40, 60, 80])
        C2 = rng.choice([0, 0, 320, 640]) if C1 % 8 == 0 else 0
        if (C1 + C2) % 32 or C1 + C2 > 2560:
            C2 = 0
        return T.test_groupnorm, (C1, C2, rng.choice([16, 36, 64, 100, 144, 256, 576, 1024, 2304]), rng.random() < 0.5, rng.choice([1e-5, 1e-6]))
    if k == "gn1":
        import ctypes  # noqa: F401
        from layoutllm_t2i_amd import _lib
        for _ in range(50):
            C1 = 32 * rng.choice([8, 10, 20, 30, 40, 60, 80])
            C2 = rng.choice([0, 320, 640, 1280])
            HW = rng.choice([16, 64, 100, 144, 256, 576, 1024])
            if C1 + C2 <= 2560 and _lib.lib().gl_groupnorm_launches(C1 + C2, HW) == 1:
                return T.test_groupnorm_single_launch_form_matches_two_launch_form, (C1, C2, HW)
        return None
    if k == "gn32":
        # round 4: fp32 inputs (the residual stream), [hi | lo] output and the raw [hi | lo] split of the input, every launch form
        C1 = 32 * rng.choice([2, 8, 10, 20, 30, 40, 60, 80])
        C2 = rng.choice([0, 0, 320, 640])
        if (C1 + C2) % 32 or C1 + C2 > 2560:
            C2 = 0
        return T.test_groupnorm_fp32_stream_hilo_and_raw_split, (C1, C2, rng.choice([16, 36, 64, 100, 144, 256, 576, 1024, 2304]), rng.random() < 0.5)
    if k == "hilo_a":
        # split-fp16 activations against one copy of W (kwrap): 4-wave, 8-wave, split-K and skinny dispatch
        return T.test_gemm_split_fp16_activation, (rng.randint(1, 3000), 32 * rng.randint(1, 40), c64(64, 1536), rng.random() < 0.5)
    if k == "hilo_out":
        return T.test_gemm_hilo_output, (rng.randint(1, 2500), 8 * rng.randint(4, 170), c64(64, 2560))
    if k == "ln":
        return T.test_layernorm, (8 * rng.randint(1, 256),)
    if k == "ff":
        return T.test_ff_fused, (rng.randint(1, 3000), rng.choice([64, 128, 192, 256, 320]), rng.choice(["res32_f16", "gate_f32", "res16_f16", "res32_f32"]))
    if k == "qkv":
        H, d = rng.choice([(2, 32), (4, 16), (8, 8), (8, 40), (8, 80), (4, 48), (8, 24), (8, 160)])     # C = H * d is a multiple of 64 (K of the GEMM)
        return T.test_gemm_qkv_writes_v_transposed, (rng.choice([1, 2, 3]), rng.randint(1, 700), H * d, H)
    if k == "merge_ln":
        return T.test_rela_merge_with_fused_layernorm, (8 * rng.randint(1, 256), rng.choice([2, 4, 6, 8, 12]), rng.random() < 0.3)
    return None


t0, n, fails = time.time(), 0, 0
while time.time() - t0 < budget:
    c = case()
    if c is None:
        continue
    fn, args = c
    n += 1
    try:
        fn(*args)
    except Exception as e:      # noqa: BLE001
        fails += 1
        msg = traceback.format_exc().strip().splitlines()[-1]
        print(f"FAIL {fn.__name__}{args}: {type(e).__name__}: {msg[:300]}", flush=True)
print(f"fuzz: {n} cases, {fails} failures, {time.time() - t0:.0f} s")
