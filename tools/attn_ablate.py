"""Ablation timing of the d = 40 / d = 80 self-attention kernel: which of {QK^T MFMAs, softmax VALU, exp, P.V MFMAs, tile staging}
the time is made of, and how much of it overlaps.  Builds patched copies of csrc/attention.hip (text substitutions below, never
committed into the product source), links each with the product objects into tools/probes/_build/libablate_<name>.so and times
gl_attention through it in a fresh process per variant.

    python tools/attn_ablate.py build          # here (hipcc cross-compiles)
    python tools/attn_ablate.py run            # on the GPU box: prints one line per variant
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "layoutllm_t2i_amd", "csrc")
OUT = os.path.join(HERE, "probes", "_build")

EXP_A = "__builtin_amdgcn_exp2f(s[qt][kh][r])"
EXP_B = "__builtin_amdgcn_exp2f(s[qt][kh][r + 1])"
PV = "for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma32(vf, *reinterpret_cast<const half8_t*>(&pf[qt][j]), o[qt][dt]);"
QK = "for (int qt = 0; qt < QT; ++qt) s[qt][kh] = mfma32(kf, qf[qt][ks], ks == 0 ? (PRE == 1 ? negm[qt] : zero16) : s[qt][kh]);"
SM_BEGIN = "        uint4 pf[QT][4];\n#pragma unroll\n        for (int qt = 0; qt < QT; ++qt) {\n            if (tail) {"
LOAD = "if (t + 1 < ntiles) load_tile(key0 + KT);"
STORE = "if (t + 1 < ntiles) store_tile((t + 1) & 1);"

VARIANTS = {
    "full": [],
    "noexp": [(EXP_A, "(s[qt][kh][r])"), (EXP_B, "(s[qt][kh][r + 1])")],
    "nopv": [(PV, "for (int qt = 0; qt < QT; ++qt) { asm volatile(\"\" :: \"v\"(pf[qt][j].x), \"v\"(pf[qt][j].y), \"v\"(pf[qt][j].z), \"v\"(pf[qt][j].w), \"v\"(vf)); }")],
    "noqk": [(QK, "for (int qt = 0; qt < QT; ++qt) { if (ks == 0) s[qt][kh] = zero16; asm volatile(\"\" : \"+v\"(s[qt][kh]) : \"v\"(kf)); }")],
    # softmax removed: P is an opaque zero, the scores are consumed by an empty asm
    "nosm": [(SM_BEGIN, "        uint4 pf[QT][4];\n#pragma unroll\n        for (int qt = 0; qt < QT; ++qt) {\n"
              "            for (int j = 0; j < 4; ++j) { pf[qt][j] = make_uint4(0u, 0u, 0u, 0u); asm volatile(\"\" : \"+v\"(pf[qt][j].x), \"+v\"(pf[qt][j].y), \"+v\"(pf[qt][j].z), \"+v\"(pf[qt][j].w) : \"v\"(s[qt][0]), \"v\"(s[qt][1])); }\n"
              "            if (p.Nq > 0) continue;\n            if (tail) {")],
    "nomfma": [(PV, "for (int qt = 0; qt < QT; ++qt) { asm volatile(\"\" :: \"v\"(pf[qt][j].x), \"v\"(pf[qt][j].y), \"v\"(pf[qt][j].z), \"v\"(pf[qt][j].w), \"v\"(vf)); }"),
               (QK, "for (int qt = 0; qt < QT; ++qt) { if (ks == 0) s[qt][kh] = zero16; asm volatile(\"\" : \"+v\"(s[qt][kh]) : \"v\"(kf)); }")],
    "nostage": [(LOAD, ""), (STORE, "")],
}
TMAX = "            tmax = fmaxf(tmax, sv(31));"
VARIANTS["nomax"] = [(TMAX, "            tmax = 0.0f * sv(31);")]
CVT = "const half2_t ph = __builtin_convertvector(pv, half2_t);"
VARIANTS["nocvt"] = [(CVT, "half2_t ph; { const float pq = p0 + p1; ph = *reinterpret_cast<const half2_t*>(&pq); }")]
DISPATCH = "            return launch_attn<DQK, 1, 8>(a, st);"
VARIANTS["qt2w8"] = [(DISPATCH, "            return launch_attn<DQK, 2, 8>(a, st);")]
VARIANTS["qt2w4"] = [(DISPATCH, "            return launch_attn<DQK, 2, 4>(a, st);")]
VARIANTS["qt1w4"] = [(DISPATCH, "            return launch_attn<DQK, 1, 4>(a, st);")]
SHFL = "tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));\n                float inc = 0.0f;"
VARIANTS["swap"] = [(SHFL, "{ float t_a = tmax, t_b = tmax; asm volatile(\"v_permlane32_swap_b32 %0, %1\" : \"+v\"(t_a), \"+v\"(t_b)); tmax = fmaxf(t_a, t_b); }\n                float inc = 0.0f;")]
VARIANTS["qt2w8_swap"] = VARIANTS["qt2w8"] + VARIANTS["swap"]
KDECL = "__global__ __launch_bounds__(64 * NW) void attn_kernel"
CAP = [(KDECL, "__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 8))) void attn_kernel")]
VARIANTS["qt2w8_cap"] = VARIANTS["qt2w8"] + CAP
VARIANTS["qt2w4_cap"] = VARIANTS["qt2w4"] + CAP
VARIANTS["qt2w8_cap_swap"] = VARIANTS["qt2w8_swap"] + CAP
VARIANTS["mfmaonly"] = VARIANTS["nosm"] + VARIANTS["nostage"]
VARIANTS["nostage_nobar"] = [(LOAD, ""), (STORE, "continue;")]
VARIANTS["noexp_nostage"] = VARIANTS["noexp"] + VARIANTS["nostage"]


def build():
    sys.path.insert(0, REPO)
    from layoutllm_t2i_amd.csrc import build as B
    B.build(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "attention.hip")).read()
    objs = [os.path.join(B.OBJDIR, s.replace(".hip", ".o")) for s in B.SOURCES if s != "attention.hip"]
    for name, subs in VARIANTS.items():
        text = src
        for a, b in subs:
            assert text.count(a) >= 1, (name, a)
            text = text.replace(a, b)
        p = os.path.join(OUT, f"attention_{name}.hip")
        open(p, "w").write(text)
        o = p.replace(".hip", ".o")
        subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + CSRC,
                        "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-c", p, "-o", o], check=True)
        subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libablate_{name}.so"), o, *objs], check=True)
        os.remove(o)
        print("built", name, flush=True)


def one(name):
    sys.path.insert(0, REPO)
    import layoutllm_t2i_amd._lib as L
    L.LIB_PATH = os.path.join(OUT, f"libablate_{name}.so")
    import torch
    from layoutllm_t2i_amd import ops
    L.init_device()
    dev = "cuda:0"
    res = []
    for d, N in ((40, 4096), (80, 1024)):
        C = 8 * d
        q, k, v = (torch.randn(8, N, C, device=dev).to(torch.float16) for _ in range(3))
        vt = torch.empty(8, 8, d, ops.vt_ld(N), dtype=torch.float16, device=dev)
        ops.transpose_v(v, N * C, C, vt, 8, 8, d, N)
        o = torch.empty(8, N, C, dtype=torch.float16, device=dev)
        f = lambda: ops.attention(q, N * C, C, k, N * C, C, vt, o, N * C, C, 8, 8, d, N, N, d ** -0.5, q_prescaled=True)
        for _ in range(200 if d == 40 else 400):      # clocks ramp over the first tens of milliseconds
            f()
        best = 1e9
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        res.append(best)
    print(f"{name:10s} d=40 N=4096: {res[0]:7.1f} us    d=80 N=1024: {res[1]:6.1f} us", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        names = sys.argv[2:] or list(VARIANTS)
        for rep in range(2):
            for name in names:
                subprocess.run([sys.executable, os.path.abspath(__file__), "one", name], check=False)
    else:
        one(sys.argv[2])
