#!/usr/bin/env python3
"""Per-shape microbenchmark of the hot-path kernels at config-2 sizes (2B = 8 samples, 64x64 latent).

    python tools/kbench.py [gemm] [conv] [attn] [norm]      (GPU box)

Walks the real block plan, collects every distinct GEMM / conv / attention / norm shape with its
multiplicity per UNet forward, times each with HIP events and prints TFLOP/s (or GB/s), the time per
forward it accounts for, and totals.  Used to direct kernel optimisation; numbers quoted in DESIGN.md.
"""
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import EPI_BIAS, EPI_GEGLU, EPI_RES, init_device
from layoutllm_t2i_amd.arch import UNetConfig, build_plan

DEV = "cuda:0"
B2 = int(os.environ.get("KB_B2", "8"))
SIDE = int(os.environ.get("KB_SIDE", "64"))
ITERS = int(os.environ.get("KB_ITERS", "20"))
STRICT = bool(int(os.environ.get("KB_STRICT", "0")))    # the strict mode's operand forms: [hi | lo] activations, [Whi | Wlo] weights, three passes


def timeit(fn, iters=ITERS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def collect():
    cfg = UNetConfig()
    plan = build_plan(cfg)
    gemms, convs, attns, gns, lns = OrderedDict(), OrderedDict(), OrderedDict(), OrderedDict(), OrderedDict()

    def add(d, key, n=1):
        d[key] = d.get(key, 0) + n
    side = SIDE
    mo, R, Lc, H = 30, 10, 77, cfg.num_heads

    def res(l, side, skip):
        M = B2 * side * side
        add(gns, (l.cin, side * side))
        add(convs, (side, l.cin, l.cout, 1, 0))
        add(gns, (l.cout, side * side))
        add(convs, (side, l.cout, l.cout, 1, 0))
        if l.cin != l.cout:
            add(gemms, (M, l.cout, l.cin, "bias"))

    def st(l, side):
        C, d, N = l.cin, l.d_head, side * side
        M = B2 * N
        add(gns, (C, N))
        add(gemms, (M, C, C, "bias"), 2)               # proj_in, proj_out
        add(lns, (M, C), 5)
        add(gemms, (M, 3 * C, C, "bias"))               # attn1 qkv
        add(attns, (d, N, N))
        add(gemms, (M, C, C, "res"), 3)                 # attn1.o, attn2.q(~), attn2.o
        add(gemms, (M, 8 * C, C, "geglu"))
        add(gemms, (M, C, 4 * C, "res"))
        add(attns, (d, N, Lc))
        # fuser (only on scale-1 steps)
        add(gemms, (B2 * (N + mo), 3 * C, C, "bias", "fuser"))
        add(attns, (d, N, N + mo, "fuser"))
        add(gemms, (M, C, C, "res", "fuser"))
        add(gemms, (M, 8 * C, C, "geglu", "fuser"))
        add(gemms, (M, C, 4 * C, "res", "fuser"))
    cur = side
    for b in plan.input_blocks[1:]:
        for l in b.layers:
            if l.kind == "res":
                res(l, cur, False)
            elif l.kind == "st":
                st(l, cur)
            elif l.kind == "down":
                add(convs, (cur, l.cin, l.cout, 2, 0))
                cur //= 2
    for l in plan.middle.layers:
        res(l, cur, False) if l.kind == "res" else st(l, cur)
    for b in plan.output_blocks:
        for l in b.layers:
            if l.kind == "res":
                res(l, cur, True)
            elif l.kind == "st":
                st(l, cur)
            elif l.kind == "up":
                add(convs, (cur, l.cin, l.cout, 1, 1))
                cur *= 2
    return gemms, convs, attns, gns, lns


def class_summary(iters=8, peak_tflops=2500.0, strict=False):
    """Per kernel CLASS of one fuser-off UNet forward (every distinct conv / plain-GEMM / attention shape x its multiplicity): FLOPs,
    standalone op-level time (HIP events on the current stream, fp16 outputs) and the fraction of the dense-fp16 MFMA peak -- what
    bench.py reports as ``roofline.classes``, measured in the run itself.  ``strict``: the strict mode's operand forms ([hi | lo] activations,
    [Whi | Wlo] weights, three MFMA passes, split attention); FLOPs stay the ALGORITHMIC ones, so ``frac`` is comparable with the default mode's
    (the matrix pipe issues three times that work)."""
    init_device()
    gemms, convs, attns, _, _ = collect()
    h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
    out = {}
    fl_c = t_c = 0.0
    for (side, cin, cout, stride, up), n in convs.items():
        x, w = h(B2 * side * side, cin), h(cout, 9 * cin) * ((9 * cin) ** -0.5)
        ho = side * 2 if up else side // stride
        o = torch.empty(B2 * ho * ho, cout, dtype=torch.float16, device=DEV)
        bias = torch.zeros(cout, device=DEV)
        if strict:
            x, w = h(B2 * side * side, 2 * cin), h(cout, 18 * cin) * ((9 * cin) ** -0.5)
            o = torch.empty(B2 * ho * ho, cout, dtype=torch.float32, device=DEV)
            t_c += n * timeit(lambda: ops.conv3x3(x, w, o, B2, side, side, bias, stride=stride, upsample2x=bool(up), in_split=3, w_split=True), iters)
        else:
            t_c += n * timeit(lambda: ops.conv3x3(x, w, o, B2, side, side, bias, stride=stride, upsample2x=bool(up)), iters)
        fl_c += n * 2.0 * B2 * ho * ho * cout * 9 * cin
    fl_g = t_g = 0.0
    for key, n in gemms.items():
        if len(key) > 4:
            continue                                    # fuser-only shapes
        M, N, K, epi = key[:4]
        a, w, bias = h(M, K), h(N, K) * (K ** -0.5), torch.zeros(N, device=DEV)
        if strict:
            a, w = h(M, 2 * K), h(N, 2 * K) * (K ** -0.5)
            if epi == "geglu":
                o = torch.empty(M, N, dtype=torch.float16, device=DEV)
                fn = lambda: ops.gemm(a, w, o, bias, EPI_GEGLU, hilo_a=True, wsplit=2, hilo_out=True)
            elif epi == "res":
                o, r = torch.empty(M, N, dtype=torch.float32, device=DEV), torch.randn(M, N, device=DEV)
                fn = lambda: ops.gemm(a, w, o, bias, EPI_RES, res=r, hilo_a=True, wsplit=2)
            else:
                o = torch.empty(M, 2 * N, dtype=torch.float16, device=DEV)
                fn = lambda: ops.gemm(a, w, o, bias, hilo_a=True, wsplit=2, hilo_out=True)
        elif epi == "geglu":
            o = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
            fn = lambda: ops.gemm(a, w, o, bias, EPI_GEGLU)
        elif epi == "res":
            o, r = torch.empty(M, N, dtype=torch.float32, device=DEV), torch.randn(M, N, device=DEV)
            fn = lambda: ops.gemm(a, w, o, bias, EPI_RES, res=r)      # the engine's form: fp32 residual stream in and out
        else:
            o = torch.empty(M, N, dtype=torch.float16, device=DEV)
            fn = lambda: ops.gemm(a, w, o, bias)
        t_g += n * timeit(fn, iters)
        fl_g += n * 2.0 * M * N * K
    fl_a = t_a = 0.0
    for key, n in attns.items():
        if len(key) > 3:
            continue
        d, Nq, Nk = key[:3]
        H, C = 8, 8 * d
        q, k, v = h(B2, Nq, C), h(B2, Nk, C), h(B2, Nk, C)
        vt = torch.empty(B2, H, d, ops.vt_ld(Nk), dtype=torch.float16, device=DEV)
        ops.transpose_v(v, Nk * C, C, vt, B2, H, d, Nk)
        o = torch.empty(B2, Nq, C, dtype=torch.float16, device=DEV)
        if strict:
            ql, kl, vtl, ol = q * 1e-3, k * 1e-3, vt * 1e-3, torch.empty_like(o)
            t_a += n * timeit(lambda: ops.attention(q, Nq * C, C, k, Nk * C, C, vt, o, Nq * C, C, B2, H, d, Nq, Nk, d ** -0.5, q_prescaled=True,
                                                    q_lo=ql, k_lo=kl, vt_lo=vtl, out_lo=ol), iters)
        else:
            t_a += n * timeit(lambda: ops.attention(q, Nq * C, C, k, Nk * C, C, vt, o, Nq * C, C, B2, H, d, Nq, Nk, d ** -0.5, q_prescaled=True), iters)
        fl_a += n * 4.0 * B2 * H * Nq * Nk * d
    for name, fl, t in (("conv3x3", fl_c, t_c), ("plain_gemm", fl_g, t_g), ("attention", fl_a, t_a)):
        out[name] = {"tflop_per_forward": round(fl / 1e12, 3), "ms_per_forward": round(t * 1e3, 3), "achieved_tflops": round(fl / t / 1e12, 1),
                     "frac": round(fl / t / 1e12 / peak_tflops, 4)}
    return out


def main():
    which = set(sys.argv[1:]) or {"gemm", "conv", "attn", "norm"}
    init_device()
    # KB_OPTS="22=1,13=2": gl_set_option knobs for A/B runs (see include/gligen_hip.h)
    for kv in filter(None, os.environ.get("KB_OPTS", "").split(",")):
        k, v = kv.split("=")
        ops.set_option(int(k), int(v))
    gemms, convs, attns, gns, lns = collect()
    h = lambda *s: torch.randn(*s, device=DEV).to(torch.float16)
    tot_on = tot_off = 0.0
    if "gemm" in which:
        print(f"{'GEMM M,N,K,epi':46s} {'x/fwd':>5s} {'us':>9s} {'TF/s':>8s} {'ms/fwd':>8s}")
        for key, n in gemms.items():
            M, N, K, epi = key[:4]
            fuser = len(key) > 4
            a, w = h(M, K), h(N, K) * (K ** -0.5)
            bias = torch.zeros(N, device=DEV)
            if STRICT:
                # the engine's strict forms: A rows [hi | lo], weight rows [Whi | Wlo], three passes; projections write [hi | lo] rows,
                # residual projections the fp32 stream
                a, w = h(M, 2 * K), h(N, 2 * K) * (K ** -0.5)
                if epi == "geglu":
                    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
                    fn = lambda: ops.gemm(a, w, out, bias, EPI_GEGLU, hilo_a=True, wsplit=2, hilo_out=True)
                elif epi == "res":
                    out, r = torch.empty(M, N, dtype=torch.float32, device=DEV), torch.randn(M, N, device=DEV)
                    fn = lambda: ops.gemm(a, w, out, bias, EPI_RES, res=r, hilo_a=True, wsplit=2)
                else:
                    out = torch.empty(M, 2 * N, dtype=torch.float16, device=DEV)
                    fn = lambda: ops.gemm(a, w, out, bias, hilo_a=True, wsplit=2, hilo_out=True)
            elif epi == "geglu":
                out = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
                fn = lambda: ops.gemm(a, w, out, bias, EPI_GEGLU)
            elif epi == "res":
                out = torch.empty(M, N, dtype=torch.float16, device=DEV)
                r = h(M, N)
                fn = lambda: ops.gemm(a, w, out, bias, EPI_RES, res=r)
            else:
                out = torch.empty(M, N, dtype=torch.float16, device=DEV)
                fn = lambda: ops.gemm(a, w, out, bias)
            t = timeit(fn)
            fl = 2.0 * M * N * K
            tot_on += t * n
            tot_off += 0 if fuser else t * n
            print(f"{str(key):46s} {n:5d} {t * 1e6:9.1f} {fl / t / 1e12:8.1f} {t * n * 1e3:8.3f}")
        print(f"GEMM total ms/forward: fuser-on {tot_on * 1e3:.2f}  fuser-off {tot_off * 1e3:.2f}")
    if "conv" in which:
        tc = 0.0
        print(f"{'CONV side,Cin,Cout,stride,up':46s} {'x/fwd':>5s} {'us':>9s} {'TF/s':>8s} {'ms/fwd':>8s}")
        for (side, cin, cout, stride, up), n in convs.items():
            x = h(B2 * side * side, cin)
            w = h(cout, 9 * cin) * ((9 * cin) ** -0.5)
            ho = side * 2 if up else side // stride
            out = torch.empty(B2 * ho * ho, cout, dtype=torch.float16, device=DEV)
            bias = torch.zeros(cout, device=DEV)
            fn = lambda: ops.conv3x3(x, w, out, B2, side, side, bias, stride=stride, upsample2x=bool(up))
            if STRICT:
                x, w = h(B2 * side * side, 2 * cin), h(cout, 18 * cin) * ((9 * cin) ** -0.5)
                out = torch.empty(B2 * ho * ho, cout, dtype=torch.float32, device=DEV)
                fn = lambda: ops.conv3x3(x, w, out, B2, side, side, bias, stride=stride, upsample2x=bool(up), in_split=3, w_split=True)
            t = timeit(fn)
            fl = 2.0 * B2 * ho * ho * cout * 9 * cin
            tc += t * n
            print(f"{str((side, cin, cout, stride, up)):46s} {n:5d} {t * 1e6:9.1f} {fl / t / 1e12:8.1f} {t * n * 1e3:8.3f}")
        print(f"CONV total ms/forward: {tc * 1e3:.2f}")
    if "attn" in which:
        print(f"{'ATTN d,Nq,Nk':46s} {'x/fwd':>5s} {'us':>9s} {'TF/s':>8s} {'ms/fwd':>8s}")
        ta = 0.0
        for key, n in attns.items():
            d, Nq, Nk = key[:3]
            H, C = 8, 8 * d
            q, k, v = h(B2, Nq, C), h(B2, Nk, C), h(B2, Nk, C)
            ldvt = ops.vt_ld(Nk) if not os.environ.get("KB_VTPOW2") else (Nk + 63) // 64 * 64
            vt = torch.empty(B2, H, d, ldvt, dtype=torch.float16, device=DEV)
            ops.transpose_v(v, Nk * C, C, vt, B2, H, d, Nk)
            out = torch.empty(B2, Nq, C, dtype=torch.float16, device=DEV)
            pre = bool(int(os.environ.get("KB_PRE", "1")))      # the engine's form: scale folded into q
            fn = lambda: ops.attention(q, Nq * C, C, k, Nk * C, C, vt, out, Nq * C, C, B2, H, d, Nq, Nk, d ** -0.5, q_prescaled=pre)
            if STRICT:
                ql, kl, vtl, outl = q * 1e-3, k * 1e-3, vt * 1e-3, torch.empty_like(out)
                fn = lambda: ops.attention(q, Nq * C, C, k, Nk * C, C, vt, out, Nq * C, C, B2, H, d, Nq, Nk, d ** -0.5, q_prescaled=pre,
                                           q_lo=ql, k_lo=kl, vt_lo=vtl, out_lo=outl)
            t = timeit(fn)
            tt = timeit(lambda: ops.transpose_v(v, Nk * C, C, vt, B2, H, d, Nk))
            fl = 4.0 * B2 * H * Nq * Nk * d
            ta += (t + tt) * n
            print(f"{str(key):46s} {n:5d} {t * 1e6:9.1f} {fl / t / 1e12:8.1f} {t * n * 1e3:8.3f}   (+transpose_v {tt * 1e6:.1f} us)")
        print(f"ATTN total ms/forward (fuser on): {ta * 1e3:.2f}")
    if "norm" in which:
        print(f"{'GN C,HW':46s} {'x/fwd':>5s} {'us':>9s} {'GB/s':>8s} {'ms/fwd':>8s}")
        tn = 0.0
        for (C, HW), n in gns.items():
            x = h(B2 * HW, C)
            out = torch.empty_like(x)
            partial = torch.empty(B2 * 64 * 64, dtype=torch.float32, device=DEV)
            g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
            fn = lambda: ops.groupnorm(x, None, B2, HW, g, b, 1e-5, True, out, partial)
            t = timeit(fn)
            by = 3.0 * x.numel() * 2
            tn += t * n
            print(f"{str((C, HW)):46s} {n:5d} {t * 1e6:9.1f} {by / t / 1e9:8.1f} {t * n * 1e3:8.3f}")
        for (M, C), n in lns.items():
            x = h(M, C)
            y = torch.empty_like(x)
            g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
            fn = lambda: ops.layernorm(x, y, g, b, B2, M // B2)
            t = timeit(fn)
            by = 2.0 * x.numel() * 2
            tn += t * n
            print(f"{'LN ' + str((M, C)):46s} {n:5d} {t * 1e6:9.1f} {by / t / 1e9:8.1f} {t * n * 1e3:8.3f}")
        print(f"NORM total ms/forward: {tn * 1e3:.2f}")


if __name__ == "__main__":
    main()
