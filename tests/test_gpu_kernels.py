"""Per-kernel parity on a real MI355X, through the C ABI (layoutllm_t2i_amd.ops -> libgligen_hip.so).

Reference = the same op in torch fp32 on the CPU, evaluated on the SAME fp16-rounded inputs/weights
(isolates kernel error from quantisation of the inputs).  Tolerance: rtol 1e-3 and an atol that is
MAGNITUDE-SCALED, 1e-4 x max(1, max|ref|) -- this is looser than north_star's rtol 1e-3 / atol 1e-4 wherever
the outputs exceed 1 (an fp16 output alone rounds by 4.9e-4 relative); attention uses 2e-3 / 2e-4 (P is rounded
to fp16 before P.V).  north_star's unscaled tolerance is applied to the WHOLE UNet in tests/test_gpu_configs.py
(fraction of elements outside it asserted).  Sampler arithmetic is checked bit-exactly.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from layoutllm_t2i_amd import host, ops, recipe
from layoutllm_t2i_amd._lib import (EPI_BIAS, EPI_GATE_RES, EPI_GEGLU, EPI_RES, EPI_ROWBIAS, EPI_SILU, init_device)
from layoutllm_t2i_amd.weights import geglu_interleave, pack_conv3x3

DEV = "cuda:0"
RTOL, ATOL = 1e-3, 1e-4


def rnd(tag, shape, scale=1.0):
    return torch.from_numpy(recipe.normal(f"gpu.{tag}", tuple(shape), 11)) * scale


def h16(x):
    """fp16-rounded fp32 copy (CPU) and the fp16 device tensor."""
    xh = x.to(torch.float16)
    return xh.float(), xh.to(DEV)


def check(out, ref, name, rtol=RTOL, atol=ATOL, frac_ok=0.0):
    out = out.float().cpu()
    ref = ref.float()
    assert out.shape == ref.shape, (name, out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{name}: non-finite output"
    scale = max(1.0, float(ref.abs().max()))
    err = (out - ref).abs()
    tol = atol * scale + rtol * ref.abs()
    bad = (err > tol).float().mean().item()
    rel_l2 = float((out - ref).norm() / (ref.norm() + 1e-30))
    print(f"[{name}] max|err|={err.max().item():.3e} rel_l2={rel_l2:.3e} |ref|max={float(ref.abs().max()):.3f} viol={bad:.2e}")
    assert bad <= frac_ok, f"{name}: {bad:.3e} of elements outside rtol={rtol} atol={atol}*{scale:.2f}; max err {err.max().item():.3e}, rel_l2 {rel_l2:.3e}"


@pytest.fixture(scope="module", autouse=True)
def _init():
    init_device()


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (512, 1280, 640), (240, 512, 832), (8, 1280, 320), (1024, 960, 320)])
def test_gemm_bias(M, N, K):
    a, ad = h16(rnd(f"a{M}", (M, K)))
    w, wd = h16(rnd(f"w{N}", (N, K), 1 / math.sqrt(K)))
    b = rnd(f"b{N}", (N,), 0.1)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(ad, wd, out, b.to(DEV))
    check(out, F.linear(a, w, b), f"gemm_bias_{M}x{N}x{K}")


def test_gemm_epilogues():
    M, N, K = 384, 640, 320
    a, ad = h16(rnd("ea", (M, K)))
    w, wd = h16(rnd("ew", (N, K), 1 / math.sqrt(K)))
    b = rnd("eb", (N,), 0.1)
    r, rd = h16(rnd("er", (M, N)))
    base = F.linear(a, w, b)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(ad, wd, out, b.to(DEV), EPI_SILU)
    check(out, F.silu(base), "gemm_silu")
    ops.gemm(ad, wd, out, b.to(DEV), EPI_RES, res=rd)
    check(out, base + r, "gemm_res")
    gate = torch.tensor([-0.37], dtype=torch.float32, device=DEV)
    ops.gemm(ad, wd, out, b.to(DEV), EPI_GATE_RES, res=rd, gate=gate)
    check(out, r + (-0.37) * base, "gemm_gate_res")
    rb, rbd = h16(rnd("erb", (3, N)))
    ops.gemm(ad, wd, out, b.to(DEV), EPI_ROWBIAS, rowbias=rbd, rows_per_sample=128)
    check(out, base + rb.repeat_interleave(128, 0), "gemm_rowbias")
    # no bias, strided output view
    big = torch.zeros(M, 2 * N, dtype=torch.float16, device=DEV)
    ops.gemm(ad, wd, big[:, N:], None)
    check(big[:, N:], F.linear(a, w), "gemm_strided_out")
    assert float(big[:, :N].abs().max()) == 0.0


@pytest.mark.parametrize("C", [64, 320])
def test_gemm_geglu(C, M=200):
    a, ad = h16(rnd("ga", (M, C)))
    w, _ = h16(rnd("gw", (8 * C, C), 1 / math.sqrt(C)))
    b = rnd("gb", (8 * C,), 0.1)
    wd = geglu_interleave(w).to(torch.float16).to(DEV)
    bd = geglu_interleave(b).contiguous().to(DEV)
    out = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    ops.gemm(ad, wd, out, bd, EPI_GEGLU)
    x, g = F.linear(a, w, b).chunk(2, dim=-1)
    check(out, x * F.gelu(g), f"gemm_geglu_{C}")


@pytest.mark.parametrize("M,C,mode", [(4096, 320, "res32_f16"), (1000, 320, "gate_f32"), (300, 64, "res32_f16"), (257, 128, "res16_f16"),
                                      (512, 192, "gate_f32"), (640, 256, "res32_f32")])
def test_ff_fused(M, C, mode):
    """gl_ff_fused: GEGLU projection -> erf-GELU gate -> output projection -> (gated) residual in ONE launch, the [M, 4C]
    intermediate never stored.  Checked against torch fp32 on the fp16-rounded operands (with the intermediate rounded to
    fp16 like both HIP forms do) and against the two-launch gl_gemm form; ragged last block; rows past M untouched."""
    x, xd = h16(rnd(f"ffx{M}{C}", (M, C)))
    w1, _ = h16(rnd(f"ffw1{C}", (8 * C, C), 1 / math.sqrt(C)))
    b1 = rnd(f"ffb1{C}", (8 * C,), 0.1)
    w2, w2d = h16(rnd(f"ffw2{C}", (C, 4 * C), 1 / math.sqrt(4 * C)))
    b2 = rnd(f"ffb2{C}", (C,), 0.1)
    w1d = geglu_interleave(w1).to(torch.float16).to(DEV).contiguous()
    b1d = geglu_interleave(b1).contiguous().to(DEV)
    res32 = mode.startswith("res32") or mode.startswith("gate")
    r = rnd(f"ffr{M}{C}", (M, C)) * 2.0 + 0.3
    if not res32:
        r = r.half().float()
    rd = r.to(DEV) if res32 else r.half().to(DEV)
    gate = torch.tensor([0.61], dtype=torch.float32, device=DEV) if mode.startswith("gate") else None
    odt = torch.float32 if mode.endswith("f32") else torch.float16
    out = torch.full((M + 9, C), 7.0, dtype=odt, device=DEV)
    ops.ff_fused(xd, w1d, b1d, w2d, b2.to(DEV), rd, out[:M], gate=gate)
    xg, gg = F.linear(x, w1, b1).chunk(2, dim=-1)
    hmid = (xg * F.gelu(gg)).half().float()
    yv = F.linear(hmid, w2, b2)
    ref = r + 0.61 * yv if gate is not None else yv + r
    check(out[:M], ref, f"ff_fused_{M}_{C}_{mode}")
    assert float((out[M:].float() - 7.0).abs().max()) == 0.0, "rows past M were written"
    # the two-launch form on the same operands
    hbuf = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    ops.gemm(xd, w1d, hbuf, b1d, EPI_GEGLU)
    out2 = torch.empty(M, C, dtype=odt, device=DEV)
    ops.gemm(hbuf, w2d, out2, b2.to(DEV), EPI_GATE_RES if gate is not None else EPI_RES, res=rd, gate=gate)
    d = float((out[:M].float() - out2.float()).norm() / out2.float().norm())
    print(f"[ff_fused {M}x{C} {mode}] rel-L2 vs two-launch form = {d:.2e}")
    assert d < (4e-4 if odt == torch.float16 else 2e-5), d
    if mode == "res32_f16":
        # GL_OUT_F16_HILO: [hi | lo] rows for a split-fp16 consumer (proj_out); hi = the fp16 output above, hi + lo = the fp32 output
        hl = torch.full((M + 1, 2 * C), 7.0, dtype=torch.float16, device=DEV)
        ops.ff_fused(xd, w1d, b1d, w2d, b2.to(DEV), rd, hl[:M], gate=gate, hilo_out=True)
        f32o = torch.empty(M, C, dtype=torch.float32, device=DEV)
        ops.ff_fused(xd, w1d, b1d, w2d, b2.to(DEV), rd, f32o, gate=gate)
        v = f32o.cpu()
        assert torch.equal(hl[:M, :C], out[:M]) and torch.equal(hl[:M, C:].cpu().float(), (v - v.half().float()).half().float())
        assert float((hl[M:].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K,epi", [(240, 320, 320, "bias"), (240, 320, 320, "gate"), (240, 1280, 1280, "gate"), (240, 640, 2560, "gate"),
                                       (8, 1280, 320, "silu"), (8, 1280, 1280, "bias"), (120, 320, 1280, "res32"), (960, 320, 320, "res"),
                                       (33, 64, 64, "bias")])
def test_gemm_skinny(M, N, K, epi):
    """M <= ~1000 rows (the 30-slot relation chain: 240 rows at B = 4; the timestep MLP: 8 rows): register-operand kernel,
    32 x 32 tile per block, four waves split K (gl_set_option(24, MiB) bounds its use; 0 = LDS-staged kernels only).
    Checked against torch fp32 AND against the LDS-staged kernel on the same inputs; rows past M stay untouched."""
    a, ad = h16(rnd(f"sa{M}{K}", (M, K)))
    w, wd = h16(rnd(f"sw{N}{K}", (N, K), 1 / math.sqrt(K)))
    b = rnd(f"sb{N}", (N,), 0.1)
    r, rd = h16(rnd(f"sr{M}{N}", (M, N)))
    base = F.linear(a, w, b)
    gate = torch.tensor([0.61], dtype=torch.float32, device=DEV)

    def run(out):
        if epi == "bias":
            ops.gemm(ad, wd, out[:M], b.to(DEV))
            return base
        if epi == "silu":
            ops.gemm(ad, wd, out[:M], b.to(DEV), EPI_SILU)
            return F.silu(base)
        if epi == "res":
            ops.gemm(ad, wd, out[:M], b.to(DEV), EPI_RES, res=rd)
            return base + r
        if epi == "gate":
            ops.gemm(ad, wd, out[:M], b.to(DEV), EPI_GATE_RES, res=rd, gate=gate)
            return r + 0.61 * base
        r32 = rnd(f"sr32{M}{N}", (M, N)) * 3.0 + 1.7
        ops.gemm(ad, wd, out[:M], b.to(DEV), EPI_RES, res=r32.to(DEV))
        return base + r32

    dt = torch.float32 if epi == "res32" else torch.float16
    out = torch.full((M + 40, N), 7.0, dtype=dt, device=DEV)
    ref = run(out)
    check(out[:M], ref, f"gemm_skinny_{M}x{N}x{K}_{epi}")
    assert float((out[M:] - 7.0).abs().max()) == 0.0, "rows past M were written"
    ops.set_option(24, 0)
    try:
        out2 = torch.full((M + 40, N), 7.0, dtype=dt, device=DEV)
        run(out2)
    finally:
        ops.set_option(24, 64)
    d = float((out[:M].float() - out2[:M].float()).norm() / out2[:M].float().norm())
    assert d < 5e-4, d          # same fp16 operands, fp32 accumulation in another order


@pytest.mark.parametrize("M,N,K,epi", [(512, 1280, 5120, "res"), (512, 1280, 2560, "bias"), (2048, 1280, 5120, "gate"), (100, 640, 2048, "rowbias")])
def test_gemm_split_k(M, N, K, epi):
    """few output tiles + long K -> the launcher cuts K into slices (fp32 partials + reduce/epilogue kernel)."""
    a, ad = h16(rnd(f"ska{M}{K}", (M, K)))
    w, wd = h16(rnd(f"skw{N}{K}", (N, K), 1 / math.sqrt(K)))
    b = rnd(f"skb{N}", (N,), 0.1)
    r, rd = h16(rnd(f"skr{M}{N}", (M, N)))
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    base = F.linear(a, w, b)
    ops.set_option(24, 0)               # keep the small-M case on the LDS-staged kernel this test is about
    if epi == "res":
        ops.gemm(ad, wd, out, b.to(DEV), EPI_RES, res=rd)
        ref = base + r
    elif epi == "gate":
        gate = torch.tensor([0.61], dtype=torch.float32, device=DEV)
        ops.gemm(ad, wd, out, b.to(DEV), EPI_GATE_RES, res=rd, gate=gate)
        ref = r + 0.61 * base
    elif epi == "rowbias":
        rb, rbd = h16(rnd("skrb", (4, N)))
        ops.gemm(ad, wd, out, b.to(DEV), EPI_ROWBIAS, rowbias=rbd, rows_per_sample=25)
        ref = base + rb.repeat_interleave(25, 0)
    else:
        ops.gemm(ad, wd, out, b.to(DEV))
        ref = base
    ops.set_option(24, 64)
    check(out, ref, f"gemm_splitk_{M}x{N}x{K}_{epi}")


def test_gemm_conv_256row_variant():
    """the 256-row kernels (4 waves x 64-row wave tiles, BK 32) forced everywhere via gl_set_option(7, 1); by default they
    serve problems with >= 300 such tiles (the 2B = 32 batch of configs[4])"""
    ops.set_option(7, 1)
    ops.set_option(24, 0)
    try:
        test_gemm_bias(512, 1280, 640)
        test_gemm_bias(1024, 960, 320)
        test_gemm_bias(300, 320, 320)
        test_gemm_epilogues()
        test_gemm_geglu(320)
        test_gemm_two_source()
        test_gemm_residual_stream_fp32(700, 320, 640)
        test_conv3x3("s1", 320, 320, 16)
        test_conv3x3("s2", 320, 320, 16)
        test_conv3x3("up", 64, 128, 8)
        test_conv3x3_epilogues()
    finally:
        ops.set_option(7, 300)      # library default
        ops.set_option(24, 64)


def test_gemm_conv_8wave_variant():
    """the 8-wave deep-pipelined kernel (gemm8.hip: 256 x {160, 128} tiles, LDS-DMA ring, split-K up to one round of blocks)
    forced wherever it applies via gl_set_option(30, 2) -- by default it serves the 64x64 / 32x32 levels of the full-size UNet,
    shapes too large for a CPU reference here.  Every epilogue it has: bias / SiLU / residual (fp16 and fp32 stream, with the
    fp16 copy) / gate / row bias / GEGLU / V^T tail / two-source A / split-K partials; convs stride 1, stride 2, nearest-2x;
    ragged M and N tails.  gl_debug_read(9) proves the launches went to it.  Run twice: with 256-row tiles only (key 46 = 0) and with the
    half-height 128-row tiles wherever the dispatch rules allow them (46 = 15: small grids -- every shape here except GEGLU and the V^T tail --
    and multi-round grids with a mostly empty last round: the 8192 x 1920 x 640 QKV projection with its V^T tail)."""
    for half_height in (0, 15):
        _gemm_conv_8wave_variant(half_height)


def _gemm_conv_8wave_variant(half_height):
    ops.set_option(30, 2)
    ops.set_option(24, 0)
    ops.set_option(46, half_height)

    def on8(n, fn, *a):
        ops.set_option(24, 0)          # (some of the called tests restore the skinny-GEMM default on exit)
        c0 = ops.gemm8_launch_count()
        fn(*a)
        got = ops.gemm8_launch_count() - c0
        assert got == n, f"{fn.__name__}{a}: {got} launches on the 8-wave kernel, expected {n}"

    try:
        on8(1, test_gemm_bias, 512, 1280, 640)          # BN 160
        on8(1, test_gemm_bias, 300, 320, 320)           # ragged M tail
        on8(1, test_gemm_bias, 1024, 960, 320)
        on8(1, test_gemm_bias, 700, 512, 832)           # BN 128, K = 13 tiles
        on8(1, test_gemm_bias, 257, 200, 128)           # N tail (200 = 128 + 72), one row in the second M tile
        on8(1, test_gemm_bias, 512, 320, 64)            # a single K-tile: prologue and tail only
        on8(5, test_gemm_epilogues)
        on8(1, test_gemm_geglu, 320, 700)
        on8(1, test_gemm_geglu, 64, 256)
        on8(1, test_gemm_two_source)
        on8(1, test_gemm_split_k, 512, 1280, 5120, "res")
        on8(1, test_gemm_split_k, 2048, 1280, 5120, "gate")
        on8(1, test_gemm_split_k, 512, 1280, 2560, "bias")
        on8(5, test_gemm_residual_stream_fp32, 700, 320, 640)
        on8(5, test_gemm_residual_stream_fp32, 512, 1280, 5120)
        on8(5, test_gemm_residual_stream_fp32, 8192, 640, 640)
        on8(2, test_gemm_qkv_writes_v_transposed, 2, 256, 320, 8)
        on8(2, test_gemm_qkv_writes_v_transposed, 1, 1054, 640, 8)
        on8(2, test_gemm_qkv_writes_v_transposed, 2, 286, 64, 4)
        on8(2, test_gemm_qkv_writes_v_transposed, 1, 4126, 320, 8)
        on8(1, test_conv3x3, "s1", 320, 320, 16)
        on8(1, test_conv3x3, "s2", 320, 320, 32)
        on8(1, test_conv3x3, "up", 64, 128, 8)
        on8(1, test_conv3x3, "up", 320, 320, 8)
        on8(1, test_conv3x3, "s1", 1280, 1280, 16)      # split-K conv (few tiles, 180 K-tiles)
        on8(1, test_conv3x3, "s1", 64, 128, 24)         # 1152 rows: ragged last tile; K = 9 tiles
        on8(2, test_conv3x3_epilogues, 16)
        on8(1, test_gemm_bias, 1300, 640, 640)          # 128-row tiles: ragged last tile of 11
        on8(2, test_gemm_qkv_writes_v_transposed, 8, 1024, 640, 8)     # 384 tiles of 256 rows = 1.5 rounds -> 768 half-height tiles
        on8(1, test_conv3x3, "s1", 128, 256, 20)        # 800 rows: 6.25 tiles of 128
    finally:
        ops.set_option(30, 1)
        ops.set_option(24, 64)
        ops.set_option(46, 11)


def test_gemm_conv_ksplit_variant():
    """intra-block K-split kernels (gl_set_option(13, 2) forces them everywhere; 1 = the default per-shape rule): 64-row wave tiles, partial accumulators exchanged in the
    epilogue; includes split-K + K-split (extra workspace slices) and ragged M / N edges"""
    ops.set_option(13, 2)
    ops.set_option(24, 0)
    try:
        test_gemm_bias(512, 1280, 640)
        test_gemm_bias(300, 320, 320)
        test_gemm_bias(240, 512, 832)
        test_gemm_bias(1024, 960, 320)
        test_gemm_epilogues()
        test_gemm_geglu(320)
        test_gemm_two_source()
        test_gemm_split_k(512, 1280, 5120, "res")
        test_gemm_split_k(100, 640, 2048, "rowbias")
        test_gemm_residual_stream_fp32(700, 320, 640)
        test_gemm_residual_stream_fp32(512, 1280, 5120)
        test_conv3x3("s1", 320, 320, 16)
        test_conv3x3("s2", 320, 320, 16)
        test_conv3x3("up", 64, 128, 8)
        test_conv3x3("s1", 1280, 1280, 8)
        test_conv3x3_epilogues()
    finally:
        ops.set_option(13, 3)
        ops.set_option(24, 64)


@pytest.mark.parametrize("M,N,K", [(700, 320, 640), (512, 1280, 5120), (130, 640, 2048), (8192, 640, 640)])
def test_gemm_residual_stream_fp32(M, N, K):
    """GL_OUT_F32_ROWMAJOR + res_f32: the residual is read and the sum stored in fp32 (no fp16 rounding of the stream),
    with and without the fp16 copy; also fp32 residual -> fp16-only output (the transformer block's last sum), the
    gated form, and the split-K reduction path (long K / few tiles) which must give the same fp32 values."""
    a, ad = h16(rnd(f"rsa{M}{K}", (M, K)))
    w, wd = h16(rnd(f"rsw{N}{K}", (N, K), 1 / math.sqrt(K)))
    b = rnd(f"rsb{N}", (N,), 0.1)
    r = rnd(f"rsr{M}{N}", (M, N)) * 3.0 + 1.7          # fp32 stream values that are NOT fp16-representable
    rd = r.to(DEV)
    base = F.linear(a, w, b)
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    o16 = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(ad, wd, o32, b.to(DEV), EPI_RES, res=rd, out16=o16)
    ref = base + r
    err = (o32.cpu() - ref).abs().max().item()
    print(f"[gemm_f32stream {M}x{N}x{K}] max|err| fp32 out = {err:.3e}")
    # fp32 accumulate of fp16 products: error ~1e-6 * sqrt(K) scale, far below fp16 resolution of the sum (1e-3)
    assert torch.allclose(o32.cpu(), ref, rtol=2e-5, atol=2e-5 * max(1.0, float(base.abs().max())))
    assert torch.equal(o16.cpu(), o32.cpu().half()), "the fp16 copy is the rounding of the fp32 sum"
    o32b = torch.empty_like(o32)
    ops.gemm(ad, wd, o32b, b.to(DEV), EPI_RES, res=rd)
    assert torch.equal(o32b, o32)
    only16 = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(ad, wd, only16, b.to(DEV), EPI_RES, res=rd)
    assert torch.equal(only16, o16)
    gate = torch.tensor([-0.43], dtype=torch.float32, device=DEV)
    ops.gemm(ad, wd, o32b, b.to(DEV), EPI_GATE_RES, res=rd, gate=gate)
    assert torch.allclose(o32b.cpu(), r + (-0.43) * base, rtol=2e-5, atol=2e-5 * max(1.0, float(base.abs().max())))
    # plain fp32 output without residual (proj_in, skip 1x1 conv, conv_in / down / up)
    ops.gemm(ad, wd, o32b, b.to(DEV), out16=o16)
    assert torch.allclose(o32b.cpu(), base, rtol=2e-5, atol=2e-5 * max(1.0, float(base.abs().max())))
    assert torch.equal(o16.cpu(), o32b.cpu().half())


@pytest.mark.parametrize("B,rows,C,H", [(2, 256, 320, 8), (1, 1054, 640, 8), (2, 286, 64, 4), (3, 64, 1280, 8), (1, 4126, 320, 8)])
def test_gemm_qkv_writes_v_transposed(B, rows, C, H):
    """fused QKV projection whose V third goes straight to gl_attention's V^T layout from the GEMM epilogue: must be the
    exact transpose of what the plain GEMM writes row-major (same fp16 values), for 8-aligned and ragged (N + 30) rows
    per sample, every tile shape the dispatcher picks, and the attention on top must ignore the never-written pad keys."""
    d = C // H
    M = B * rows
    a, ad = h16(rnd(f"vta{M}{C}", (M, C)))
    w, wd = h16(rnd(f"vtw{C}", (3 * C, C), 1 / math.sqrt(C)))
    plain = torch.empty(M, 3 * C, dtype=torch.float16, device=DEV)
    ops.set_option(5, 0)            # the fused form never splits K: compare with the unsplit plain GEMM (same fp32 summation order)
    ops.set_option(24, 0)           # ... on the LDS-staged kernel (the skinny kernel splits K over its four waves)
    try:
        ops.gemm(ad, wd, plain)
    finally:
        ops.set_option(5, -1)
        ops.set_option(24, 64)
    fused = torch.full((M, 3 * C), float("nan"), dtype=torch.float16, device=DEV)
    vt = torch.full((B, H, d, ops.vt_ld(rows)), float("nan"), dtype=torch.float16, device=DEV)     # NaN pads on purpose
    ops.gemm(ad, wd, fused, vt=vt, vt_col0=2 * C, vt_rows=rows)
    assert torch.equal(fused[:, :2 * C], plain[:, :2 * C]), "Q and K columns are untouched"
    assert torch.isnan(fused[:, 2 * C:]).all(), "the V columns must not be written row-major"
    want = plain[:, 2 * C:].view(B, rows, H, d).permute(0, 2, 3, 1)
    assert torch.equal(vt[..., :rows], want), float((vt[..., :rows].float() - want.float()).abs().max())
    assert torch.isnan(vt[..., rows:]).all()
    # attention over it: NaN pad keys must not leak (masked in the kernel), result == attention on a zero-padded transpose
    Nq = min(rows, 300)
    out = torch.empty(B * Nq, C, dtype=torch.float16, device=DEV)
    ops.attention(fused, rows * 3 * C, 3 * C, fused[:, C:], rows * 3 * C, 3 * C, vt, out, Nq * C, C, B, H, d, Nq, rows, d ** -0.5)
    vt0 = torch.zeros_like(vt)
    ops.transpose_v(plain[:, 2 * C:], rows * 3 * C, 3 * C, vt0, B, H, d, rows)
    out0 = torch.empty_like(out)
    ops.attention(plain, rows * 3 * C, 3 * C, plain[:, C:], rows * 3 * C, 3 * C, vt0, out0, Nq * C, C, B, H, d, Nq, rows, d ** -0.5)
    assert torch.isfinite(out).all() and torch.equal(out, out0)


def test_gemm_two_source():
    M, K1, K2, N = 260, 128, 192, 256
    a1, a1d = h16(rnd("ta1", (M, K1)))
    a2, a2d = h16(rnd("ta2", (M, K2)))
    w, wd = h16(rnd("tw", (N, K1 + K2), 1 / math.sqrt(K1 + K2)))
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a1d, wd, out, None, a2=a2d)
    check(out, F.linear(torch.cat([a1, a2], 1), w), "gemm_two_source")


def test_gemm_full_size_l0_ff():
    """config-2 level-0 GEGLU projection: M = 4096 tokens (B=1), N = 2560, K = 320."""
    M, C = 4096, 320
    a, ad = h16(rnd("fa", (M, C)))
    w, wd = h16(rnd("fw", (4 * C, C), 1 / math.sqrt(C)))
    out = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    ops.gemm(ad, wd, out, None)
    check(out, F.linear(a, w), "gemm_l0")


# ------------------------------------------------------------------------------------------- conv
def _nhwc(x):  # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


@pytest.mark.parametrize("mode", ["s1", "s2", "up"])
@pytest.mark.parametrize("Cin,Cout,hw", [(64, 128, 8), (320, 320, 16), (1280, 1280, 8)])
def test_conv3x3(mode, Cin, Cout, hw):
    B = 2
    x, _ = h16(rnd(f"cx{Cin}{mode}", (B, Cin, hw, hw)))
    w, _ = h16(rnd(f"cw{Cin}{Cout}", (Cout, Cin, 3, 3), 1 / math.sqrt(9 * Cin)))
    b = rnd(f"cb{Cout}", (Cout,), 0.1)
    xd = _nhwc(x).to(torch.float16).to(DEV)
    wd = pack_conv3x3(w).to(DEV)
    if mode == "s1":
        ref = F.conv2d(x, w, b, padding=1)
        ho = hw
    elif mode == "s2":
        ref = F.conv2d(x, w, b, stride=2, padding=1)
        ho = hw // 2
    else:
        ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
        ho = hw * 2
    out = torch.empty(B * ho * ho, Cout, dtype=torch.float16, device=DEV)
    ops.conv3x3(xd, wd, out, B, hw, hw, b.to(DEV), stride=2 if mode == "s2" else 1, upsample2x=(mode == "up"))
    check(out, _nhwc(ref), f"conv_{mode}_{Cin}_{Cout}_{hw}")


def test_conv3x3_first_and_last():
    """first conv (4 channels zero-padded to 64) and the out conv (Cout = 4, fp32 NCHW store)."""
    B, hw, mc = 2, 16, 64
    x = rnd("lat", (B, 4, hw, hw))
    w, _ = h16(rnd("fcw", (mc, 4, 3, 3), 1 / 6))
    b = rnd("fcb", (mc,), 0.1)
    xin = torch.empty(2 * B * hw * hw, 64, dtype=torch.float16, device=DEV)
    ops.pack_latent(x.to(DEV), 64, 2, xin)
    xr = x.to(torch.float16).float()
    ref_pack = torch.cat([_nhwc(xr)] * 2, 0)
    assert torch.equal(xin[:, :4].float().cpu(), ref_pack) and float(xin[:, 4:].abs().max()) == 0.0
    out = torch.empty(2 * B * hw * hw, mc, dtype=torch.float16, device=DEV)
    ops.conv3x3(xin, pack_conv3x3(w, 64).to(DEV), out, 2 * B, hw, hw, b.to(DEV))
    check(out, torch.cat([_nhwc(F.conv2d(xr, w, b, padding=1))] * 2, 0), "conv_first")
    h, _ = h16(rnd("lh", (B, mc, hw, hw)))
    w2, _ = h16(rnd("lw", (4, mc, 3, 3), 1 / math.sqrt(9 * mc)))
    b2 = rnd("lb", (4,), 0.1)
    eps = torch.empty(B, 4, hw, hw, dtype=torch.float32, device=DEV)
    ops.conv3x3(_nhwc(h).to(torch.float16).to(DEV), pack_conv3x3(w2).to(DEV), eps, B, hw, hw, b2.to(DEV), nchw_hw=hw * hw)
    check(eps, F.conv2d(h, w2, b2, padding=1), "conv_last_nchw_f32", rtol=1e-4, atol=1e-5)


def test_first_conv_split_latent_and_weights():
    """gl_pack_latent split + weights.pack_first_conv: the latent goes in as [hi | lo | hi] channels against [Whi | Whi | Wlo] weights in the 64
    padded input channels the first conv carries anyway -- against the fp64 conv of the UNROUNDED latent and weights the error drops from the
    fp16-operand level to fp32-accumulation level; with fp16-representable x and W the result equals the unsplit form bit for bit."""
    from layoutllm_t2i_amd.weights import pack_first_conv
    B, hw, mc = 2, 16, 64
    x = rnd("slat", (B, 4, hw, hw)) * 1.3
    w = rnd("sfcw", (mc, 4, 3, 3), 1 / 6)
    b = rnd("sfcb", (mc,), 0.1)
    xs = torch.empty(B * hw * hw, 64, dtype=torch.float16, device=DEV)
    ops.pack_latent(x.to(DEV), 64, 1, xs, split=True)
    hi = x.half()
    lo = (x - hi.float()).half()
    assert torch.equal(xs[:, 0:4].cpu(), _nhwc(hi)) and torch.equal(xs[:, 4:8].cpu(), _nhwc(lo)) and torch.equal(xs[:, 8:12].cpu(), _nhwc(hi))
    assert float(xs[:, 12:].abs().max()) == 0.0
    out = torch.empty(B * hw * hw, mc, dtype=torch.float32, device=DEV)
    ops.conv3x3(xs, pack_first_conv(w, 64).to(DEV), out, B, hw, hw, b.to(DEV))
    ref = _nhwc(F.conv2d(x.double(), w.double(), b.double(), padding=1).float())
    plain_in = torch.empty_like(xs)
    ops.pack_latent(x.to(DEV), 64, 1, plain_in)
    plain = torch.empty_like(out)
    ops.conv3x3(plain_in, pack_conv3x3(w.half(), 64).to(DEV), plain, B, hw, hw, b.to(DEV))
    e_split = float((out.cpu() - ref).norm() / ref.norm())
    e_plain = float((plain.cpu() - ref).norm() / ref.norm())
    print(f"[first_conv_split] rel_l2 split={e_split:.2e} plain fp16={e_plain:.2e}")
    assert e_split < 2e-6 and e_plain > 20 * e_split, (e_split, e_plain)
    # fp16-representable operands: lo = 0 and Wlo = 0, the extra channels add exact zeros
    xr, wr = x.half().float(), w.half().float()
    ops.pack_latent(xr.to(DEV), 64, 1, xs, split=True)
    ops.conv3x3(xs, pack_first_conv(wr, 64).to(DEV), out, B, hw, hw, b.to(DEV))
    ops.pack_latent(xr.to(DEV), 64, 1, plain_in)
    ops.conv3x3(plain_in, pack_conv3x3(wr.half(), 64).to(DEV), plain, B, hw, hw, b.to(DEV))
    assert torch.equal(out, plain)


def test_conv3x3_epilogues(hw=8):
    B, C = 2, 128
    x, _ = h16(rnd("cex", (B, C, hw, hw)))
    w, _ = h16(rnd("cew", (C, C, 3, 3), 1 / math.sqrt(9 * C)))
    b = rnd("ceb", (C,), 0.1)
    emb, embd = h16(rnd("cee", (B, 3 * C)))
    res, _ = h16(rnd("cer", (B, C, hw, hw)))
    xd = _nhwc(x).to(torch.float16).to(DEV)
    wd = pack_conv3x3(w).to(DEV)
    out = torch.empty(B * hw * hw, C, dtype=torch.float16, device=DEV)
    base = F.conv2d(x, w, b, padding=1)
    ops.conv3x3(xd, wd, out, B, hw, hw, b.to(DEV), epi=EPI_ROWBIAS, rowbias=embd[:, C:2 * C], rows_per_sample=hw * hw)
    check(out, _nhwc(base + emb[:, C:2 * C, None, None]), "conv_rowbias")
    ops.conv3x3(xd, wd, out, B, hw, hw, b.to(DEV), epi=EPI_RES, res=_nhwc(res).to(torch.float16).to(DEV))
    check(out, _nhwc(base + res), "conv_res")


# ------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, H):
    B, Nq, C = q.shape
    d = C // H
    qh = q.view(B, Nq, H, d).transpose(1, 2)
    kh = k.view(B, -1, H, d).transpose(1, 2)
    vh = v.view(B, -1, H, d).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2)) * d ** -0.5
    return torch.matmul(s.softmax(-1), vh).transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize("d,H,Nq,Nk,B", [
    (16, 4, 64, 64, 2), (32, 4, 200, 230, 2), (40, 8, 256, 286, 1), (64, 2, 130, 77, 2),
    (80, 8, 1024, 1054, 1), (160, 8, 256, 286, 1), (160, 8, 64, 64, 2), (40, 8, 30, 10, 2), (40, 8, 4096, 4126, 1)])
def test_attention(d, H, Nq, Nk, B):
    C = H * d
    q, qd = h16(rnd(f"aq{d}{Nq}", (B, Nq, C)))
    k, kd = h16(rnd(f"ak{d}{Nk}", (B, Nk, C)))
    v, vd = h16(rnd(f"av{d}{Nk}", (B, Nk, C)))
    ldvt = (Nk + 63) // 64 * 64
    vt = torch.full((B, H, d, ldvt), float("nan"), dtype=torch.float16, device=DEV)
    ops.transpose_v(vd, Nk * C, C, vt, B, H, d, Nk)
    vt_ref = torch.zeros(B, H, d, ldvt)
    vt_ref[..., :Nk] = v.view(B, Nk, H, d).permute(0, 2, 3, 1)
    assert torch.equal(vt.float().cpu(), vt_ref), "transpose_v"
    out = torch.empty(B, Nq, C, dtype=torch.float16, device=DEV)
    ops.attention(qd, Nq * C, C, kd, Nk * C, C, vt, out, Nq * C, C, B, H, d, Nq, Nk, d ** -0.5)
    # P is rounded to fp16 before P.V (4.9e-4 relative per weight): allow 2x the base tolerance
    check(out, _attn_ref(q, k, v, H), f"attn_d{d}_q{Nq}_k{Nk}", rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("d,H,Nq,Nk,B", [(16, 4, 64, 64, 2), (40, 8, 256, 286, 1), (64, 2, 130, 77, 2), (80, 8, 1024, 1054, 1),
                                        (160, 8, 256, 286, 1), (40, 8, 30, 10, 2), (40, 8, 4096, 4126, 1)])
def test_attention_prescaled_q(d, H, Nq, Nk, B):
    """q_prescaled: d^-1/2 * log2(e) folded into Q beforehand (the packer folds it into the q projection weights), the running
    max subtracted inside the MFMA.  Reference = softmax over the SAME fp16-rounded prescaled q; includes rows whose first
    tile is far below / above zero (m starts at 0 and must lock onto the true row max on tile 0) and a late outlier key."""
    from layoutllm_t2i_amd.weights import q_fold
    C = H * d
    c = q_fold(d)
    q = rnd(f"pq{d}{Nq}", (B, Nq, C))
    q[:, 0] *= 30.0                                  # one query with huge logits (|s| in the hundreds of exp2 units)
    k = rnd(f"pk{d}{Nk}", (B, Nk, C))
    if Nk > 100:
        k[:, Nk - 3] = q[:, 1] * 4.0                 # a late key that jumps query 1's running max by >> 2^8
    qs, qd = h16(q * c)                              # what the q projection GEMM would write
    k, kd = h16(k)
    v, vd = h16(rnd(f"pv{d}{Nk}", (B, Nk, C)))
    vt = torch.full((B, H, d, ops.vt_ld(Nk)), float("nan"), dtype=torch.float16, device=DEV)
    ops.transpose_v(vd, Nk * C, C, vt, B, H, d, Nk)
    out = torch.empty(B, Nq, C, dtype=torch.float16, device=DEV)
    ops.attention(qd, Nq * C, C, kd, Nk * C, C, vt, out, Nq * C, C, B, H, d, Nq, Nk, 123.0, q_prescaled=True)   # scale must be ignored
    ref = _attn_ref(qs / c, k, v, H)                 # softmax(d^-1/2 (q'/c) k^T) == softmax over the prescaled logits
    check(out, ref, f"attn_prescaled_d{d}_q{Nq}_k{Nk}", rtol=2e-3, atol=2e-4)


def test_attention_padding_column_max_vs_fma_path():
    """d = 40 (padded to 48): with q_prescaled the running max travels in the operands' padding column (K pad column = 1.0,
    Q pad column = -m, fp16) so the QK^T MFMA delivers logit - m; gl_set_option(29, 0) restores the FMA path.  Both must match
    the reference and each other on logits of every size (rows whose max is tiny, huge, negative; a late outlier; ragged Nq / Nk;
    the 8 head-dim padded d = 24 case as well)."""
    for (d, H, Nq, Nk, B) in [(40, 8, 300, 1054, 1), (40, 8, 4096, 4126, 1), (24, 4, 130, 200, 2), (8, 2, 64, 70, 1)]:
        test_attention_prescaled_q(d, H, Nq, Nk, B)
        from layoutllm_t2i_amd.weights import q_fold
        C = H * d
        q = rnd(f"pmq{d}{Nq}", (B, Nq, C))
        q[:, 0] *= 60.0
        q[:, 2] *= 1e-3
        k = rnd(f"pmk{d}{Nk}", (B, Nk, C))
        k[:, Nk - 2] = -q[:, 3] * 5.0                      # a late key far BELOW query 3's max, and (by symmetry) far above others'
        _, qd = h16(q * q_fold(d))
        _, kd = h16(k)
        _, vd = h16(rnd(f"pmv{d}{Nk}", (B, Nk, C)))
        vt = torch.full((B, H, d, ops.vt_ld(Nk)), float("nan"), dtype=torch.float16, device=DEV)
        ops.transpose_v(vd, Nk * C, C, vt, B, H, d, Nk)
        o1 = torch.empty(B, Nq, C, dtype=torch.float16, device=DEV)
        o0 = torch.empty_like(o1)
        ops.attention(qd, Nq * C, C, kd, Nk * C, C, vt, o1, Nq * C, C, B, H, d, Nq, Nk, 1.0, q_prescaled=True)
        ops.set_option(29, 0)
        try:
            ops.attention(qd, Nq * C, C, kd, Nk * C, C, vt, o0, Nq * C, C, B, H, d, Nq, Nk, 1.0, q_prescaled=True)
        finally:
            ops.set_option(29, 1)
        assert torch.isfinite(o1).all()
        dl = float((o1.float() - o0.float()).norm() / o0.float().norm())
        assert dl < 6e-4, (d, Nq, Nk, dl)                  # two roundings of P to fp16 about different reference maxima


@pytest.mark.parametrize("opt", [3, 4])
def test_attention_block_size_variants(opt):
    """8-wave (256-query) and 4-wave blocks forced via gl_set_option(3, 3|4); includes a ragged last slab"""
    ops.set_option(3, opt)
    try:
        test_attention(40, 8, 600, 1054, 1)
        test_attention(80, 8, 300, 1054, 1)
        test_attention_prescaled_q(40, 8, 256, 286, 1)
        test_attention_prescaled_q(80, 8, 1024, 1054, 1)
    finally:
        ops.set_option(3, 0)


def test_attention_strided_qkv_and_spike():
    """fused-QKV addressing (row stride 3C, batch stride (N+30)*3C, Nq < rows) and an outlier key that
    forces a late running-max jump in the online softmax."""
    B, H, d, N, mo = 2, 8, 40, 192, 30
    C = H * d
    rows = N + mo
    qkv, qkvd = h16(rnd("sqkv", (B, rows, 3 * C)))
    qkv[0, 150, C:2 * C] *= 6.0      # spike one key (tile 2)
    qkv = qkv.to(torch.float16).float()
    qkvd = qkv.to(torch.float16).to(DEV)
    flat = qkvd.view(B * rows, 3 * C)
    vt = torch.empty(B, H, d, 256, dtype=torch.float16, device=DEV)
    ops.transpose_v(flat[:, 2 * C:], rows * 3 * C, 3 * C, vt, B, H, d, rows)
    out = torch.empty(B * N, C, dtype=torch.float16, device=DEV)
    ops.attention(flat, rows * 3 * C, 3 * C, flat[:, C:], rows * 3 * C, 3 * C, vt, out, N * C, C, B, H, d, N, rows, d ** -0.5)
    ref = _attn_ref(qkv[:, :N, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], H)
    check(out.view(B, N, C), ref, "attn_strided_spike", rtol=2e-3, atol=2e-4)


# ------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("C1,C2,HW,silu,eps", [(320, 0, 4096, True, 1e-5), (640, 320, 256, True, 1e-5), (1280, 1280, 64, True, 1e-5),
                                               (64, 0, 64, False, 1e-6), (1280, 0, 256, False, 1e-6), (960, 0, 1024, True, 1e-5),
                                               (1280, 0, 64, True, 1e-5), (1280, 0, 144, False, 1e-6), (1280, 1280, 256, True, 1e-5),
                                               (1280, 640, 256, True, 1e-5), (256, 0, 16, True, 1e-5)])
def test_groupnorm(C1, C2, HW, silu, eps):
    B = 2
    C = C1 + C2
    x1, x1d = h16(rnd(f"g1{C1}{HW}", (B * HW, C1)) * 1.5 + 0.3)
    x2 = x2d = None
    if C2:
        x2, x2d = h16(rnd(f"g2{C2}{HW}", (B * HW, C2)) * 0.7 - 0.2)
    gam = 1 + 0.1 * rnd(f"gg{C}", (C,))
    bet = 0.1 * rnd(f"gb{C}", (C,))
    out = torch.empty(B * HW, C, dtype=torch.float16, device=DEV)
    partial = torch.empty(B * 64 * 64, dtype=torch.float32, device=DEV)
    ops.groupnorm(x1d, x2d, B, HW, gam.to(DEV), bet.to(DEV), eps, silu, out, partial)
    x = x1 if x2 is None else torch.cat([x1, x2], 1)
    xn = x.view(B, HW, C).permute(0, 2, 1)
    ref = F.group_norm(xn, 32, gam, bet, eps)
    if silu:
        ref = F.silu(ref)
    check(out, ref.permute(0, 2, 1).reshape(B * HW, C), f"groupnorm_{C1}+{C2}_{HW}")


@pytest.mark.parametrize("C1,C2,HW", [(1280, 1280, 256), (320, 0, 1024), (640, 320, 256), (640, 0, 1024), (1280, 640, 256), (1280, 0, 576), (320, 0, 100)])
def test_groupnorm_single_launch_form_matches_two_launch_form(C1, C2, HW):
    """the single-launch kernels (knob 17: whole-vector groups on small maps; group BUNDLES of 2 / 4 groups where a group is
    10 / 20 / 30 / 60 channels) against the statistics + apply pair on the same input: same result to fp16 rounding (different
    fp32 summation order); the launch-count query tells which form a shape takes"""
    from layoutllm_t2i_amd import _lib
    l = _lib.lib()
    assert l.gl_groupnorm_launches(1280, 256) == 1 and l.gl_groupnorm_launches(2560, 256) == 1 and l.gl_groupnorm_launches(1280, 64) == 1
    assert l.gl_groupnorm_launches(640, 1024) == 1 and l.gl_groupnorm_launches(1920, 256) == 1 and l.gl_groupnorm_launches(320, 1024) == 1
    assert l.gl_groupnorm_launches(320, 4096) == 2 and l.gl_groupnorm_launches(960, 1024) == 2 and l.gl_groupnorm_launches(128, 4096) == 2   # > 80 KB per block / 4-channel groups
    B = 2
    C = C1 + C2
    assert l.gl_groupnorm_launches(C, HW) == 1
    x1, x1d = h16(rnd(f"gf1{C1}{HW}", (B * HW, C1)) * 1.3 + 0.4)
    x2d = None
    if C2:
        _, x2d = h16(rnd(f"gf2{C2}{HW}", (B * HW, C2)) * 0.6 - 3.0)
    gam, bet = (1 + 0.1 * rnd(f"gfg{C}", (C,))).to(DEV), (0.1 * rnd(f"gfb{C}", (C,))).to(DEV)
    partial = torch.empty(B * 64 * 64, dtype=torch.float32, device=DEV)
    a, b = (torch.full((B * HW + 3, C), 7.0, dtype=torch.float16, device=DEV) for _ in range(2))
    ops.groupnorm(x1d, x2d, B, HW, gam, bet, 1e-5, True, a[:B * HW], partial)
    ops.set_option(17, 0)
    try:
        ops.groupnorm(x1d, x2d, B, HW, gam, bet, 1e-5, True, b[:B * HW], partial)
    finally:
        ops.set_option(17, 1)
    d = (a.float() - b.float()).abs()
    assert float(d.max()) <= 4e-3 and float((d > 0).float().mean()) < 0.03, (float(d.max()), float((d > 0).float().mean()))
    assert float((a[B * HW:].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("C1,C2,HW,silu", [(320, 0, 4096, True), (640, 320, 1024, True), (320, 0, 1024, False), (640, 0, 256, True), (1280, 640, 256, True),
                                          (1280, 1280, 64, True), (1280, 0, 256, False), (960, 0, 1024, True), (640, 0, 1024, False), (64, 0, 100, True)])
def test_groupnorm_fp32_stream_hilo_and_raw_split(C1, C2, HW, silu):
    """gl_groupnorm_ex on fp32 inputs (the residual stream), every launch form (statistics + apply, small-map, group bundles):
    hi = the fp16 normalised rows, hi + lo = the same rows to ~2^-22 (the split-fp16 operand of proj_in), and raw = the INPUT
    concat as [hi | lo] (the operand of a ResBlock's 1x1 skip_connection), written into column views of wider buffers."""
    B = 2
    C = C1 + C2
    x1 = rnd(f"gx1{C1}{HW}", (B * HW, C1)) * 1.5 + 0.3
    x2 = rnd(f"gx2{C2}{HW}", (B * HW, C2)) * 0.7 - 0.2 if C2 else None
    gam = 1 + 0.1 * rnd(f"gxg{C}", (C,))
    bet = 0.1 * rnd(f"gxb{C}", (C,))
    out = torch.full((B * HW + 1, 2 * C), 7.0, dtype=torch.float16, device=DEV)
    raw = torch.full((B * HW + 1, 2 * C + 8), 7.0, dtype=torch.float16, device=DEV)
    partial = torch.empty(B * 64 * 64, dtype=torch.float32, device=DEV)
    ops.groupnorm(x1.to(DEV), x2.to(DEV) if C2 else None, B, HW, gam.to(DEV), bet.to(DEV), 1e-5, silu, out[:B * HW, :C], partial,
                  out_lo=out[:B * HW, C:], raw=raw[:B * HW])
    x = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.group_norm(x.double().view(B, HW, C).permute(0, 2, 1), 32, gam.double(), bet.double(), 1e-5)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(B * HW, C).float()
    o = out.cpu().float()
    check(o[:B * HW, :C], ref, f"groupnorm_f32_hi_{C1}+{C2}_{HW}")
    # hi + lo: only the fp32 arithmetic of the kernel is left (rstd, exp of the SiLU): 1e-5 relative
    check(o[:B * HW, :C] + o[:B * HW, C:], ref, f"groupnorm_f32_hilo_{C1}+{C2}_{HW}", rtol=2e-5, atol=2e-6)
    r = raw.cpu().float()
    assert torch.equal(r[:B * HW, :C], x.half().float())                                       # raw hi = fp16(x), bitwise
    assert torch.equal(r[:B * HW, C:2 * C], (x - x.half().float()).half().float())             # raw lo = fp16(x - hi), bitwise
    assert float((o[B * HW:] - 7.0).abs().max()) == 0.0 and float((r[B * HW:] - 7.0).abs().max()) == 0.0 and float((r[:, 2 * C:] - 7.0).abs().max()) == 0.0


def test_groupnorm_fp32_launch_forms():
    """fp32 inputs take the same launch form as fp16 ones on every UNet shape (the 80 KB-per-block cap of the bundle form binds
    before the register budget of its fp32 instantiations, <= 8 vectors per thread, does); the query tells the engine"""
    from layoutllm_t2i_amd import _lib
    l = _lib.lib()
    for C, HW in ((1280, 256), (2560, 64), (320, 1024), (640, 256), (640, 1024), (1920, 256), (320, 4096), (960, 1024), (1280, 4096)):
        assert l.gl_groupnorm_launches_ex(C, HW, 1) == l.gl_groupnorm_launches_ex(C, HW, 0) == l.gl_groupnorm_launches(C, HW), (C, HW)
    assert l.gl_groupnorm_launches_ex(640, 1024, 1) == 1 and l.gl_groupnorm_launches_ex(320, 4096, 1) == 2


@pytest.mark.parametrize("M,N,K,res", [(4096, 320, 320, False), (700, 640, 1280, True), (512, 1280, 1280, True), (8192, 640, 640, True),
                                       (32768, 320, 320, True), (2048, 1280, 2560, False), (240, 320, 640, False)])
def test_gemm_split_fp16_activation(M, N, K, res):
    """hilo_a: A = [hi | lo] of an fp32 activation against ONE copy of W (gl_gemm_args.kwrap) -- 4-wave, 8-wave, split-K and
    skinny dispatch.  Against the fp32 product of the UNROUNDED activation the error drops from the fp16-operand level
    (~3e-4 relative) to fp32-accumulation level; the plain fp16 product of the same rows is measured beside it."""
    x = rnd(f"sa{M}{K}", (M, K)) * 1.7 + 0.4
    w, wd = h16(rnd(f"sw{N}{K}", (N, K), 1 / math.sqrt(K)))
    b = rnd(f"sb{N}", (N,), 0.1)
    hi = x.half()
    lo = (x - hi.float()).half()
    a = torch.cat([hi, lo], 1).to(DEV)
    r = rnd(f"sr{M}{N}", (M, N)) if res else None
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(a, wd, out, b.to(DEV), EPI_RES if res else EPI_BIAS, res=r.to(DEV) if res else None, hilo_a=True)
    ref = F.linear(x.double(), w.double(), b.double()).float() + (r if res else 0)
    e_split = float((out.cpu() - ref).norm() / ref.norm())
    plain = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(hi.to(DEV), wd, plain, b.to(DEV), EPI_RES if res else EPI_BIAS, res=r.to(DEV) if res else None)
    e_plain = float((plain.cpu() - ref).norm() / ref.norm())
    print(f"[gemm_hilo_{M}x{N}x{K}] rel_l2 split={e_split:.2e} plain fp16={e_plain:.2e}")
    assert e_split < 2e-6 and e_plain > 20 * e_split, (e_split, e_plain)


@pytest.mark.parametrize("M,N,K,res", [(300, 320, 320, False), (8192, 640, 640, True), (2048, 1280, 2560, True), (33000, 320, 320, True), (2048, 1280, 1920, False)])
def test_gemm_split_fp16_activation_and_weight(M, N, K, res):
    """wsplit: weight rows [Whi | Wlo] of an UNROUNDED fp32 weight.  wsplit = 1 uses Whi only and must equal the plain hilo_a product on a
    [N, K] copy of Whi bit for bit; wsplit = 2 is the three-pass product xhi.Whi + xlo.Whi + xhi.Wlo (K index 3K: kwrap steps the weight
    column back once, the third A segment comes in as the second source) -- against the fp64 product of the unrounded x and W the error drops
    from the fp16-weight level (~2e-4) to fp32-accumulation level.  4-wave, 8-wave (both tile heights), split-K dispatch."""
    x = rnd(f"wa{M}{K}", (M, K)) * 1.7 + 0.4
    w = rnd(f"ww{N}{K}", (N, K), 1 / math.sqrt(K))
    b = rnd(f"wb{N}", (N,), 0.1)
    hi, whi = x.half(), w.half()
    lo, wlo = (x - hi.float()).half(), (w - whi.float()).half()
    a = torch.cat([hi, lo], 1).to(DEV)
    w2 = torch.cat([whi, wlo], 1).contiguous().to(DEV)
    r = rnd(f"wr{M}{N}", (M, N)) if res else None
    kw = dict(res=r.to(DEV)) if res else {}
    epi = EPI_RES if res else EPI_BIAS
    two = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(a, whi.to(DEV), two, b.to(DEV), epi, hilo_a=True, **kw)
    two_w = torch.empty_like(two)
    ops.gemm(a, w2, two_w, b.to(DEV), epi, hilo_a=True, wsplit=1, **kw)
    assert torch.equal(two, two_w)
    plain_w = torch.empty_like(two)
    ops.gemm(hi.to(DEV), w2, plain_w, b.to(DEV), epi, wsplit=1, **kw)            # fp16 A against Whi through the wide rows
    plain = torch.empty_like(two)
    ops.gemm(hi.to(DEV), whi.to(DEV), plain, b.to(DEV), epi, **kw)
    assert torch.equal(plain, plain_w)
    three = torch.empty_like(two)
    ops.gemm(a, w2, three, b.to(DEV), epi, hilo_a=True, wsplit=2, **kw)
    ref = F.linear(x.double(), w.double(), b.double()).float() + (r if res else 0)
    e3 = float((three.cpu() - ref).norm() / ref.norm())
    e2 = float((two.cpu() - ref).norm() / ref.norm())
    print(f"[gemm_hilo_w_{M}x{N}x{K}] rel_l2 three-pass={e3:.2e} two-pass (fp16 weight)={e2:.2e}")
    assert e3 < 2e-6 and e2 > 20 * e3, (e3, e2)


@pytest.mark.parametrize("M,N,K", [(300, 320, 1280), (4096, 640, 2560), (512, 1280, 5120)])
def test_gemm_hilo_output(M, N, K):
    """GL_OUT_F16_HILO: the epilogue writes hi = fp16(v) and lo = fp16(v - hi) N columns apart (bias + fp32 residual epilogue)."""
    a, ad = h16(rnd(f"ha{M}", (M, K)))
    w, wd = h16(rnd(f"hw{N}", (N, K), 1 / math.sqrt(K)))
    b = rnd(f"hb{N}", (N,), 0.1)
    r = rnd(f"hr{M}{N}", (M, N))
    out = torch.full((M + 1, 2 * N), 7.0, dtype=torch.float16, device=DEV)
    ops.gemm(ad, wd, out[:M], b.to(DEV), EPI_RES, res=r.to(DEV), hilo_out=True)
    f32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(ad, wd, f32, b.to(DEV), EPI_RES, res=r.to(DEV))
    v = f32.cpu()
    o = out.cpu().float()
    assert torch.equal(o[:M, :N], v.half().float()) and torch.equal(o[:M, N:], (v - v.half().float()).half().float())
    assert float((o[M:] - 7.0).abs().max()) == 0.0
    check(o[:M, :N] + o[:M, N:], F.linear(a, w, b) + r, f"gemm_hilo_out_{M}x{N}x{K}", rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("C1,C2,HW", [(320, 0, 4096), (640, 320, 1024), (1280, 0, 144), (1280, 0, 256), (2560, 0, 64)])
def test_groupnorm_large_mean_channels(C1, C2, HW):
    """channels whose |mean| >> std (real checkpoints have them): a one-pass E[x^2] - mean^2 in fp32 loses the variance
    (mean^2 = 900 vs var = 0.04 leaves ~3 significant bits); the shifted / pairwise-merged statistics must match
    torch's Welford GroupNorm on the same fp16-rounded inputs."""
    B = 2
    C = C1 + C2
    x = rnd(f"gl{C}{HW}", (B * HW, C)) * 0.2
    x = x + torch.linspace(-30.0, 30.0, C)[None, :]                      # per-channel offsets up to 150 sigma
    x[:, : C // 32] = rnd(f"glc{C}", (B * HW, C // 32)) * 0.05 + 29.7    # one whole group: std 0.05 around 29.7
    x1, x1d = h16(x[:, :C1].contiguous())
    x2 = x2d = None
    if C2:
        x2, x2d = h16(x[:, C1:].contiguous())
    gam = 1 + 0.1 * rnd(f"glg{C}", (C,))
    bet = 0.1 * rnd(f"glb{C}", (C,))
    out = torch.empty(B * HW, C, dtype=torch.float16, device=DEV)
    partial = torch.empty(B * 64 * 64, dtype=torch.float32, device=DEV)
    ops.groupnorm(x1d, x2d, B, HW, gam.to(DEV), bet.to(DEV), 1e-5, False, out, partial)
    xr = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.group_norm(xr.double().view(B, HW, C).permute(0, 2, 1), 32, gam.double(), bet.double(), 1e-5).float()
    check(out, ref.permute(0, 2, 1).reshape(B * HW, C), f"groupnorm_large_mean_{C1}+{C2}_{HW}")


def test_layernorm_fp32_stream_and_stats():
    """fp32 input rows (the residual stream) and the (mean, rstd) side output rela_merge re-normalises with"""
    for C in (64, 320, 1280):
        B, rows = 2, 77
        x = rnd(f"lf{C}", (B * rows, C)) * 2 + 0.5 + 1e-4 * rnd(f"lf2{C}", (B * rows, C))    # not fp16-representable
        gam = 1 + 0.1 * rnd(f"lfg{C}", (C,))
        bet = 0.1 * rnd(f"lfb{C}", (C,))
        y = torch.empty(B * rows, C, dtype=torch.float16, device=DEV)
        st = torch.empty(B * rows, 2, dtype=torch.float32, device=DEV)
        ops.layernorm(x.to(DEV), y, gam.to(DEV), bet.to(DEV), B, rows, stats=st)
        ref = F.layer_norm(x, (C,), gam, bet, 1e-5)
        check(y, ref, f"layernorm_f32_{C}")
        mean = x.mean(-1)
        rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt()
        assert torch.allclose(st[:, 0].cpu(), mean, rtol=1e-5, atol=1e-6) and torch.allclose(st[:, 1].cpu(), rstd, rtol=1e-5, atol=1e-6)


def test_layernorm_two_sources_one_launch():
    """[x (fp32 stream) ; objs (fp16)] normalised into the fuser's concat buffer by ONE launch == the two separate launches"""
    B, N, mo, C = 2, 144, 30, 320
    rows = N + 32
    x = rnd("l2x", (B * N, C)) * 2 + 0.3
    o, od = h16(rnd("l2o", (B * mo, C)))
    gam, bet = (1 + 0.1 * rnd("l2g", (C,))).to(DEV), (0.1 * rnd("l2b", (C,))).to(DEV)
    a = torch.zeros(B * rows, C, dtype=torch.float16, device=DEV)
    b = torch.zeros(B * rows, C, dtype=torch.float16, device=DEV)
    ops.layernorm(x.to(DEV), a, gam, bet, B, N, rows, 0)
    ops.layernorm(od, a, gam, bet, B, mo, rows, N)
    ops.layernorm(x.to(DEV), b, gam, bet, B, N, rows, 0, x2=od, rows2=mo)
    assert torch.equal(a, b)
    ref = torch.zeros(B, rows, C)
    ref[:, :N] = F.layer_norm(x, (C,), gam.cpu(), bet.cpu(), 1e-5).view(B, N, C)
    ref[:, N:N + mo] = F.layer_norm(o, (C,), gam.cpu(), bet.cpu(), 1e-5).view(B, mo, C)
    check(b, ref.view(-1, C), "layernorm_two_sources")


@pytest.mark.parametrize("C", [64, 320, 640, 1280])
def test_layernorm(C):
    B, rows = 2, 100
    x, xd = h16(rnd(f"l{C}", (B * rows, C)) * 2 + 0.5)
    gam = 1 + 0.1 * rnd(f"lg{C}", (C,))
    bet = 0.1 * rnd(f"lb{C}", (C,))
    y = torch.empty(B * rows, C, dtype=torch.float16, device=DEV)
    ops.layernorm(xd, y, gam.to(DEV), bet.to(DEV), B, rows)
    ref = F.layer_norm(x, (C,), gam, bet, 1e-5)
    check(y, ref, f"layernorm_{C}")
    # remapped rows: write into the tail of a [B, rows+30, C] concat buffer
    cat = torch.zeros(B * (rows + 30), C, dtype=torch.float16, device=DEV)
    ops.layernorm(xd[:B * 30], cat, gam.to(DEV), bet.to(DEV), B, 30, rows + 30, rows)
    cref = torch.zeros(B, rows + 30, C)
    cref[:, rows:] = F.layer_norm(x[:B * 30], (C,), gam, bet, 1e-5).view(B, 30, C)
    check(cat, cref.view(-1, C), f"layernorm_remap_{C}")


# ------------------------------------------------------------------------------------------- rela
def _rela_closed_form(x, hid, f, rects, nvalid, poison, B, H, W, C, mo):
    acc = torch.zeros(B, H, W, C)
    for b in range(B):
        for i in range(int(nvalid[b])):
            t, bo, l, r = [int(v) for v in rects[b, i]]
            acc[b, t:bo, l:r] += f[b, i]
    y = 0.5 * ((hid.view(B, H, W, C) + acc / mo) + x.view(B, H, W, C))
    for b in range(B):
        if poison[b]:
            y[b] = float("nan")
    return y.view(B * H * W, C)


@pytest.mark.parametrize("C,hw", [(64, 8), (320, 32)])
def test_rela_pool_merge(C, hw):
    B, mo = 2, 30
    boxes = np.zeros((B, mo, 4), np.float32)
    masks = np.zeros((B, mo), np.float32)
    boxes[0, :4] = [(0.0, 0.0, 0.5, 0.5), (0.25, 0.25, 1.0, 1.0), (0.6, 0.1, 0.9, 0.45), (0.13, 0.55, 0.41, 0.99)]
    masks[0, :4] = 1
    boxes[1, :2] = [(0.1, 0.2, 0.8, 0.9), (0.5, 0.5, 1.2, 1.3)]
    masks[1, :2] = 1
    rects, nvalid, poison = host.box_rects(boxes, masks, hw, hw)
    hid, hidd = h16(rnd(f"rh{C}", (B * hw * hw, C)))
    x, xd = h16(rnd(f"rx{C}", (B * hw * hw, C)))
    dr, dn, dp = (torch.from_numpy(a).to(DEV) for a in (rects, nvalid, poison))
    feat = torch.empty(B * mo, C, dtype=torch.float16, device=DEV)
    ops.rela_pool(hidd, B, hw, hw, C, dr, dn, dp, mo, feat)
    ref = torch.zeros(B, mo, C)
    hv = hid.view(B, hw, hw, C)
    for b in range(B):
        for i in range(int(nvalid[b])):
            t, bo, l, r = [int(v) for v in rects[b, i]]
            ref[b, i] = hv[b, t:bo, l:r].reshape(-1, C).mean(0)
    check(feat, ref.view(B * mo, C), f"rela_pool_{C}_{hw}")
    # fused norm1: LayerNorm of the fp16-rounded pooled row, bitwise equal to pool followed by gl_layernorm
    gam, bet = (1 + 0.1 * rnd(f"rpg{C}", (C,))).to(DEV), (0.1 * rnd(f"rpb{C}", (C,))).to(DEV)
    feat2, fn_fused = torch.empty_like(feat), torch.empty_like(feat)
    ops.rela_pool(hidd, B, hw, hw, C, dr, dn, dp, mo, feat2, ln_gamma=gam, ln_beta=bet, ln_out=fn_fused)
    fn_sep = ops.layernorm(feat, torch.empty_like(feat), gam, bet, B, mo)
    assert torch.equal(feat2, feat)
    check(fn_fused, F.layer_norm(feat.float().cpu(), (C,), gam.cpu(), bet.cpu(), 1e-5), f"rela_pool_ln_{C}_{hw}")
    print("fused vs separate LN max diff", float((fn_fused.float() - fn_sep.float()).abs().max()))
    f, fd = h16(rnd(f"rf{C}", (B * mo, C)))
    y = torch.empty(B * hw * hw, C, dtype=torch.float16, device=DEV)
    ops.rela_merge(xd, hidd, fd, B, hw, hw, C, dr, dn, dp, mo, y)
    check(y, _rela_closed_form(x, hid, f.view(B, mo, C), rects, nvalid, poison, B, hw, hw, C, mo), f"rela_merge_{C}_{hw}")


@pytest.mark.parametrize("C,hw", [(64, 8), (320, 32), (1280, 8), (640, 16)])
def test_rela_pool_on_the_fp32_stream_and_layernorm_stats(C, hw):
    """round 4: gl_layernorm_stats = the statistics gl_layernorm stores (bitwise), and gl_rela_pool_ln3 = box means of LN3(x) evaluated in
    fp32 from the stream x and those statistics -- against torch fp32 (mean over the rectangle of layer_norm(x)) and against the fp16-hid
    form it replaces (which differs by the rounding of hid it no longer has)."""
    B, mo = 2, 30
    boxes = np.zeros((B, mo, 4), np.float32)
    masks = np.zeros((B, mo), np.float32)
    boxes[0, :4] = [(0.0, 0.0, 0.5, 0.5), (0.25, 0.25, 1.0, 1.0), (0.6, 0.1, 0.9, 0.45), (0.13, 0.55, 0.41, 0.99)]
    masks[0, :4] = 1
    boxes[1, :2] = [(0.1, 0.2, 0.8, 0.9), (0.5, 0.5, 1.2, 1.3)]
    masks[1, :2] = 1
    rects, nvalid, poison = host.box_rects(boxes, masks, hw, hw)
    dr, dn, dp = (torch.from_numpy(a).to(DEV) for a in (rects, nvalid, poison))
    x = rnd(f"px{C}{hw}", (B * hw * hw, C)) * 1.7 + 0.4
    g3, b3 = 1 + 0.1 * rnd(f"pg{C}", (C,)), 0.1 * rnd(f"pb{C}", (C,))
    xd = x.to(DEV)
    stats = torch.empty(B * hw * hw, 2, dtype=torch.float32, device=DEV)
    ops.layernorm_stats(xd, stats)
    stats_ln = torch.empty_like(stats)
    hid = ops.layernorm(xd, torch.empty(B * hw * hw, C, dtype=torch.float16, device=DEV), g3.to(DEV), b3.to(DEV), B, hw * hw, stats=stats_ln)
    assert torch.equal(stats, stats_ln)
    feat = torch.full((B * mo + 1, C), 7.0, dtype=torch.float16, device=DEV)
    g1, b1 = (1 + 0.1 * rnd(f"p1g{C}", (C,))).to(DEV), (0.1 * rnd(f"p1b{C}", (C,))).to(DEV)
    fn = torch.empty(B * mo, C, dtype=torch.float16, device=DEV)
    ops.rela_pool_ln3(xd, stats, g3.to(DEV), b3.to(DEV), B, hw, hw, C, dr, dn, dp, mo, feat[:B * mo], ln_gamma=g1, ln_beta=b1, ln_out=fn)
    hv = F.layer_norm(x.double(), (C,), g3.double(), b3.double(), 1e-5).view(B, hw, hw, C)
    ref = torch.zeros(B, mo, C, dtype=torch.float64)
    for b in range(B):
        for i in range(int(nvalid[b])):
            t, bo, l, r = [int(v) for v in rects[b, i]]
            ref[b, i] = hv[b, t:bo, l:r].reshape(-1, C).mean(0)
    check(feat[:B * mo], ref.view(B * mo, C).float(), f"rela_pool_ln3_{C}_{hw}")
    assert float((feat[B * mo:].float() - 7.0).abs().max()) == 0.0
    old = torch.empty(B * mo, C, dtype=torch.float16, device=DEV)
    ops.rela_pool(hid, B, hw, hw, C, dr, dn, dp, mo, old)
    assert float((old.float() - feat[:B * mo].float()).abs().max()) < 4e-3
    fn_sep = ops.layernorm(feat[:B * mo].contiguous(), torch.empty_like(fn), g1, b1, B, mo)        # fused norm1 of the pooled rows
    assert float((fn.float() - fn_sep.float()).abs().max()) < 4e-3


@pytest.mark.parametrize("C,hw,slots", [(320, 16, 8), (640, 8, 16), (64, 12, 8)])
def test_rela_chain_rows_per_sample(C, hw, slots):
    """ABI 13: ``slots`` rows per sample for feat / ln_out / f while rects keep max_objs = 30 slots and 1/30 stays the divisor: the pooled
    rows and the merged stream are BITWISE those of the 30-row form (rows are independent; the reference computes all 30, attention.py:348-351)."""
    B, mo = 3, 30
    boxes = np.zeros((B, mo, 4), np.float32)
    masks = np.zeros((B, mo), np.float32)
    boxes[0, :4] = [(0.0, 0.0, 0.5, 0.5), (0.25, 0.25, 1.0, 1.0), (0.6, 0.1, 0.9, 0.45), (0.13, 0.55, 0.41, 0.99)]
    masks[0, :4] = 1
    boxes[1, :7] = [(0.1 * i, 0.05 * i, 0.1 * i + 0.4, 0.05 * i + 0.5) for i in range(7)]
    masks[1, :7] = 1                                   # sample 2: no boxes at all (the uncond half)
    rects, nvalid, poison = host.box_rects(boxes, masks, hw, hw)
    assert int(nvalid.max()) <= slots
    dr, dn, dp = (torch.from_numpy(a).to(DEV) for a in (rects, nvalid, poison))
    x = (rnd(f"sx{C}{hw}", (B * hw * hw, C)) * 1.7 + 0.4).to(DEV)
    g3, b3 = (1 + 0.1 * rnd(f"sg{C}", (C,))).to(DEV), (0.1 * rnd(f"sb{C}", (C,))).to(DEV)
    g1, b1 = (1 + 0.1 * rnd(f"s1g{C}", (C,))).to(DEV), (0.1 * rnd(f"s1b{C}", (C,))).to(DEV)
    stats = torch.empty(B * hw * hw, 2, dtype=torch.float32, device=DEV)
    ops.layernorm_stats(x, stats)
    out = {}
    for ms in (mo, slots):
        feat = torch.full((B * ms + 1, C), 7.0, dtype=torch.float16, device=DEV)
        fn = torch.full((B * ms, C), 7.0, dtype=torch.float16, device=DEV)
        ops.rela_pool_ln3(x, stats, g3, b3, B, hw, hw, C, dr, dn, dp, mo, feat[:B * ms], ln_gamma=g1, ln_beta=b1, ln_out=fn, slots=ms if ms != mo else 0)
        assert float((feat[B * ms:].float() - 7.0).abs().max()) == 0.0                                  # nothing past the last row
        f = (feat[:B * ms].float() * 0.5 + fn.float()).half()                                            # stand-in for the chain's output rows
        y = torch.empty_like(x)
        ops.rela_merge(x, None, f, B, hw, hw, C, dr, dn, dp, mo, y, ln_stats=stats, gamma=g3, beta=b3, slots=ms if ms != mo else 0)
        g2, b2 = g1, b1
        y2, ln2 = torch.empty_like(x), torch.empty(B * hw * hw, C, dtype=torch.float16, device=DEV)
        ops.rela_merge(x, None, f, B, hw, hw, C, dr, dn, dp, mo, y2, ln_stats=stats, gamma=g3, beta=b3, ln2_gamma=g2, ln2_beta=b2, ln2_out=ln2,
                       slots=ms if ms != mo else 0)
        out[ms] = (feat[:B * ms].view(B, ms, C)[:, :slots].clone(), fn.view(B, ms, C)[:, :slots].clone(), y, y2, ln2)
    for a, b in zip(out[mo], out[slots]):
        assert torch.equal(a, b)
    with pytest.raises(Exception):
        ops.rela_merge(x, None, f, B, hw, hw, C, dr, dn, dp, mo, y, ln_stats=stats, gamma=g3, beta=b3, slots=mo + 1)


def test_layernorm_fp32_second_source_and_fp32_rowbias():
    """round 4: the fuser's LayerNorm over [x ; objs] with objs in fp32 (gl_layernorm x_f32 bit 2), and GL_EPI_ROWBIAS with an fp32 row bias
    (the emb_layers output): both against torch fp32 on the UNROUNDED second operand."""
    B, rows, ro, C = 2, 70, 6, 320
    x = rnd("l2x", (B * rows, C)) * 1.3 + 0.2
    o = rnd("l2o", (B * ro, C)) * 2.0 - 0.3
    gam, bet = 1 + 0.1 * rnd("l2g", (C,)), 0.1 * rnd("l2b", (C,))
    y = torch.zeros(B * (rows + 8), C, dtype=torch.float16, device=DEV)
    ops.layernorm(x.to(DEV), y, gam.to(DEV), bet.to(DEV), B, rows, rows + 8, 0, x2=o.to(DEV), rows2=ro)
    cat = torch.cat([x.view(B, rows, C), o.view(B, ro, C)], 1)
    ref = F.layer_norm(cat, (C,), gam, bet, 1e-5)
    check(y.view(B, rows + 8, C)[:, :rows + ro], ref, "layernorm_x2_fp32")
    y16 = torch.zeros_like(y)
    ops.layernorm(x.to(DEV), y16, gam.to(DEV), bet.to(DEV), B, rows, rows + 8, 0, x2=o.half().to(DEV), rows2=ro)
    assert torch.equal(y16.view(B, rows + 8, C)[:, :rows], y.view(B, rows + 8, C)[:, :rows])        # the x rows do not depend on the form of x2
    # fp32 row bias through the conv epilogue
    Bc, hw, Cin, Cout = 2, 8, 64, 128
    xi, xid = h16(rnd("rb32x", (Bc * hw * hw, Cin)))
    w = rnd("rb32w", (Cout, Cin, 3, 3), 1 / math.sqrt(9 * Cin)).half().float()
    bias = rnd("rb32b", (Cout,), 0.1)
    rb = rnd("rb32r", (Bc, Cout)) * 1.5
    out = torch.empty(Bc * hw * hw, Cout, dtype=torch.float32, device=DEV)
    ops.conv3x3(xid, pack_conv3x3(w).to(DEV), out, Bc, hw, hw, bias.to(DEV), epi=EPI_ROWBIAS, rowbias=rb.to(DEV), rows_per_sample=hw * hw)
    refc = F.conv2d(xi.view(Bc, hw, hw, Cin).permute(0, 3, 1, 2), w, bias, padding=1) + rb[:, :, None, None]
    check(out, refc.permute(0, 2, 3, 1).reshape(Bc * hw * hw, Cout), "conv_rowbias_fp32", rtol=2e-5, atol=2e-6)


def test_rela_merge_fp32_stream_recomputes_ln():
    """x / y in fp32 and hid = LN3(x) re-evaluated in fp32 from gl_layernorm's (mean, rstd): the result must match the
    fp32 closed form to fp32 rounding -- no fp16 rounding of x, hid or y on the residual stream."""
    B, mo, C, hw = 2, 30, 320, 16
    boxes = np.zeros((B, mo, 4), np.float32)
    masks = np.zeros((B, mo), np.float32)
    boxes[0, :3] = [(0.0, 0.0, 0.5, 0.5), (0.25, 0.25, 1.0, 1.0), (0.6, 0.1, 0.9, 0.45)]
    masks[0, :3] = 1
    boxes[1, :1] = [(0.1, 0.2, 0.8, 0.9)]
    masks[1, :1] = 1
    rects, nvalid, poison = host.box_rects(boxes, masks, hw, hw)
    x = rnd("rm32x", (B * hw * hw, C)) * 1.3 + 0.2
    gam = 1 + 0.1 * rnd("rm32g", (C,))
    bet = 0.1 * rnd("rm32b", (C,))
    f, fd = h16(rnd("rm32f", (B * mo, C)))
    dr, dn, dp = (torch.from_numpy(a).to(DEV) for a in (rects, nvalid, poison))
    hid16 = torch.empty(B * hw * hw, C, dtype=torch.float16, device=DEV)
    st = torch.empty(B * hw * hw, 2, dtype=torch.float32, device=DEV)
    ops.layernorm(x.to(DEV), hid16, gam.to(DEV), bet.to(DEV), B, hw * hw, stats=st)
    y = torch.empty(B * hw * hw, C, dtype=torch.float32, device=DEV)
    ops.rela_merge(x.to(DEV), None, fd, B, hw, hw, C, dr, dn, dp, mo, y, ln_stats=st, gamma=gam.to(DEV), beta=bet.to(DEV))
    hid = F.layer_norm(x, (C,), gam, bet, 1e-5)
    ref = _rela_closed_form(x, hid, f.view(B, mo, C), rects, nvalid, poison, B, hw, hw, C, mo)
    assert torch.allclose(y.cpu(), ref, rtol=1e-5, atol=1e-5), float((y.cpu() - ref).abs().max())


@pytest.mark.parametrize("C,hw,poisoned", [(320, 16, False), (640, 8, False), (1280, 8, False), (2048, 4, False), (320, 8, True)])
def test_rela_merge_with_fused_layernorm(C, hw, poisoned):
    """ln2_out form (one wave per token): y and LayerNorm(y) from one launch must equal, bit for bit, the elementwise
    gl_rela_merge followed by gl_layernorm -- for every channel width's register layout (1..4 vectors per lane), overlapping
    boxes, an empty sample and a NaN-poisoned sample."""
    B, mo = 3, 30
    boxes = np.zeros((B, mo, 4), np.float32)
    masks = np.zeros((B, mo), np.float32)
    boxes[0, :4] = [(0.0, 0.0, 0.5, 0.5), (0.25, 0.25, 1.0, 1.0), (0.6, 0.1, 0.9, 0.45), (0.3, 0.3, 0.7, 0.7)]
    masks[0, :4] = 1
    boxes[2, :2] = [(0.1, 0.2, 0.8, 0.9), (0.5, 0.5, 0.1, 0.1) if poisoned else (0.0, 0.5, 1.0, 1.0)]
    masks[2, :2] = 1
    rects, nvalid, poison = host.box_rects(boxes, masks, hw, hw)
    assert bool(poison[2]) == poisoned
    x = (rnd(f"rml{C}x", (B * hw * hw, C)) * 1.3 + 0.2).to(DEV)
    g3, b3, g2, b2 = ((1 + 0.1 * rnd(f"rml{C}{k}", (C,))).to(DEV) if k in "ac" else (0.1 * rnd(f"rml{C}{k}", (C,))).to(DEV) for k in "abcd")
    _, fd = h16(rnd(f"rml{C}f", (B * mo, C)))
    dr, dn, dp = (torch.from_numpy(a).to(DEV) for a in (rects, nvalid, poison))
    M = B * hw * hw
    hid16 = torch.empty(M, C, dtype=torch.float16, device=DEV)
    st = torch.empty(M, 2, dtype=torch.float32, device=DEV)
    ops.layernorm(x, hid16, g3, b3, B, hw * hw, stats=st)
    y0 = torch.empty(M, C, dtype=torch.float32, device=DEV)
    ops.rela_merge(x, None, fd, B, hw, hw, C, dr, dn, dp, mo, y0, ln_stats=st, gamma=g3, beta=b3)
    n0 = torch.empty(M, C, dtype=torch.float16, device=DEV)
    ops.layernorm(y0, n0, g2, b2, B, hw * hw)
    y1 = torch.full((M, C), 7.0, dtype=torch.float32, device=DEV)
    n1 = torch.full((M, C), 7.0, dtype=torch.float16, device=DEV)
    ops.rela_merge(x, None, fd, B, hw, hw, C, dr, dn, dp, mo, y1, ln_stats=st, gamma=g3, beta=b3, ln2_gamma=g2, ln2_beta=b2, ln2_out=n1)
    eq = lambda a, b: torch.equal(torch.nan_to_num(a.float(), nan=123.0), torch.nan_to_num(b.float(), nan=123.0))
    assert eq(y1, y0), float(torch.nan_to_num(y1 - y0).abs().max())
    assert eq(n1, n0), float(torch.nan_to_num(n1.float() - n0.float()).abs().max())
    assert torch.isnan(y1.view(B, -1)[2]).all() == poisoned and torch.isfinite(y1.view(B, -1)[:2]).all()


RELA_GOLDENS = [("rela_normal", 8), ("rela_degenerate", 8), ("rela_null", 8), ("rela_clamp", 8), ("rela_maskgap", 16), ("rela_empty_slice", 8)]


@pytest.mark.parametrize("name,hw", RELA_GOLDENS, ids=[g[0] for g in RELA_GOLDENS])
def test_rela_fuse_reference_goldens_through_hip(name, hw):
    """The six RelationCrossAttention goldens produced by the REFERENCE module (break rule at the first padded /
    degenerate box, x1/y1 clamping, mask gap, null grounding, NaN for an empty slice) through the HIP kernels exactly as
    the engine chains them: LN3 (+stats) -> rela_pool -> LN1 -> q GEMM -> attention over the relation tokens -> gated
    o-proj -> LN2 -> GEGLU FF -> gated ff2 -> rela_merge; module output = 2 y - x (attention.py:315-359, :398)."""
    import golden_cases as gc
    from layoutllm_t2i_amd import arch
    from layoutllm_t2i_amd.weights import geglu_interleave
    case = next(c for c in gc.CASES if c["name"] == name)
    inp = {a: torch.from_numpy(v) for a, v in gc.case_inputs(case).items()}
    C, heads, mo = case["C"], case["heads"], 30
    d = C // heads
    sd = {n: torch.from_numpy(np.asarray(recipe.tensor(f"golden.{name}.{n}", shp, 0))) for n, shp in arch.rela_params("", C, gc.CTX).items()}
    dv = lambda t: t.float().contiguous().to(DEV)
    hd = lambda t: t.to(torch.float16).contiguous().to(DEV)
    B, R = inp["x"].shape[0], inp["relations"].shape[1]
    N = hw * hw
    rects, nvalid, poison = host.box_rects(inp["boxes"].numpy(), inp["masks"].numpy(), hw, hw)
    dr, dn, dp = (torch.from_numpy(a).to(DEV) for a in (rects, nvalid, poison))
    x = inp["x"].reshape(B * N, C)
    xd = dv(x)
    e16 = lambda *shape: torch.empty(*shape, dtype=torch.float16, device=DEV)
    st = torch.empty(B * N, 2, dtype=torch.float32, device=DEV)
    hid = ops.layernorm(xd, e16(B * N, C), dv(sd["norm3.weight"]), dv(sd["norm3.bias"]), B, N, stats=st)
    fn = e16(B * mo, C)
    feat = ops.rela_pool(hid, B, hw, hw, C, dr, dn, dp, mo, e16(B * mo, C), ln_gamma=dv(sd["norm1.weight"]), ln_beta=dv(sd["norm1.bias"]), ln_out=fn)
    q = ops.gemm(fn, hd(sd["attn.to_q.weight"]), e16(B * mo, C))
    kv = ops.gemm(hd(inp["relations"].reshape(B * R, -1)), hd(torch.cat([sd["attn.to_k.weight"], sd["attn.to_v.weight"]], 0)), e16(B * R, 2 * C))
    vt = torch.zeros(B, heads, d, ops.vt_ld(R), dtype=torch.float16, device=DEV)
    ops.transpose_v(kv[:, C:], R * 2 * C, 2 * C, vt, B, heads, d, R)
    ar = e16(B * mo, C)
    ops.attention(q, mo * C, C, kv, R * 2 * C, 2 * C, vt, ar, mo * C, C, B, heads, d, mo, R, d ** -0.5)
    ga = torch.tanh(sd["alpha_attn"]).reshape(1).float().to(DEV)
    gdn = torch.tanh(sd["alpha_dense"]).reshape(1).float().to(DEV)
    f1 = ops.gemm(ar, hd(sd["attn.to_out.0.weight"]), e16(B * mo, C), dv(sd["attn.to_out.0.bias"]), EPI_GATE_RES, res=feat, gate=ga)
    fn2 = ops.layernorm(f1, e16(B * mo, C), dv(sd["norm2.weight"]), dv(sd["norm2.bias"]), B, mo)
    hg = ops.gemm(fn2, hd(geglu_interleave(sd["ff.net.0.proj.weight"])), e16(B * mo, 4 * C), dv(geglu_interleave(sd["ff.net.0.proj.bias"])), EPI_GEGLU)
    f2 = ops.gemm(hg, hd(sd["ff.net.2.weight"]), e16(B * mo, C), dv(sd["ff.net.2.bias"]), EPI_GATE_RES, res=f1, gate=gdn)
    y = torch.empty(B * N, C, dtype=torch.float32, device=DEV)
    ops.rela_merge(xd, None, f2, B, hw, hw, C, dr, dn, dp, mo, y, ln_stats=st, gamma=dv(sd["norm3.weight"]), beta=dv(sd["norm3.bias"]))
    out = (2.0 * y.cpu() - x).view(B, N, C)
    ref = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))["out"])
    nan_ref = torch.isnan(ref)
    assert torch.equal(torch.isnan(out), nan_ref), "NaN pattern (empty-slice poison) must match the reference exactly"
    if name == "rela_empty_slice":
        assert nan_ref.any()
    ok = ~nan_ref
    err = (out[ok] - ref[ok]).abs()
    rl2 = float((out[ok] - ref[ok]).norm() / ref[ok].norm())
    print(f"[{name}] rel_l2={rl2:.3e} max|err|={float(err.max()):.3e}")
    # the box features pass through fp16 GEMMs but enter the output divided by max_objs = 30; LN3 itself is fp32 here
    assert rl2 < 3e-4 and float(err.max()) < 2e-3, (rl2, float(err.max()))


def test_rela_poison_and_null():
    B, mo, C, hw = 2, 30, 64, 8
    boxes = np.zeros((B, mo, 4), np.float32)
    masks = np.zeros((B, mo), np.float32)
    boxes[0, :2] = [(0.1, 0.1, 0.6, 0.6), (0.7, 0.2, 0.2, 0.5)]   # second: right < left -> NaN sample
    masks[0, :2] = 1
    rects, nvalid, poison = host.box_rects(boxes, masks, hw, hw)
    assert list(nvalid) == [2, 0] and list(poison) == [1, 0]
    hid, hidd = h16(rnd("ph", (B * hw * hw, C)))
    x, xd = h16(rnd("px", (B * hw * hw, C)))
    f, fd = h16(rnd("pf", (B * mo, C)))
    dr, dn, dp = (torch.from_numpy(a).to(DEV) for a in (rects, nvalid, poison))
    y = torch.empty(B * hw * hw, C, dtype=torch.float16, device=DEV)
    ops.rela_merge(xd, hidd, fd, B, hw, hw, C, dr, dn, dp, mo, y)
    y = y.float().cpu().view(B, -1)
    assert torch.isnan(y[0]).all(), "poisoned sample must be NaN everywhere (reference: 0 * NaN)"
    ref1 = 0.5 * (hid.view(B, -1)[1] + x.view(B, -1)[1])      # null grounding: (LN3(x) + x) / 2
    assert torch.allclose(y[1], ref1, rtol=1e-3, atol=1e-3)


# ------------------------------------------------------------------------------------------- small kernels
def test_posnet_input_and_timestep_embedding():
    from oracle import unet_ref
    B, mo, dim = 2, 30, 768
    boxes = torch.from_numpy(np.abs(recipe.uniform("gpu.pb", (B, mo, 4), 3)))
    masks = torch.zeros(B, mo)
    masks[0, :5] = 1
    masks[1, :1] = 1
    emb = rnd("pe", (B, mo, dim))
    npos, nxy = rnd("pnp", (dim,), 0.5), rnd("pnx", (64,), 0.5)
    out = torch.empty(B * mo, dim + 64, dtype=torch.float16, device=DEV)
    ops.posnet_input(boxes.to(DEV), masks.to(DEV), emb.to(DEV), npos.to(DEV), nxy.to(DEV), 8, out)
    m = masks.unsqueeze(-1)
    ref = torch.cat([emb * m + (1 - m) * npos, unet_ref.fourier_embed(boxes, 8) * m + (1 - m) * nxy], -1)
    check(out, ref.view(B * mo, -1), "posnet_input")
    t = torch.tensor([1.0, 21.0, 500.0, 981.0])
    te = torch.empty(4, 320, dtype=torch.float16, device=DEV)
    ops.timestep_embedding(t.to(DEV), 320, te)
    check(te, unet_ref.timestep_embedding(t, 320), "timestep_embedding")
    x, xd = h16(rnd("sx", (16, 1280)) * 3)
    y = torch.empty_like(xd)
    ops.silu(xd, y)
    check(y, F.silu(x), "silu")


def test_sampler_arithmetic_bit_exact():
    """cfg_combine + plms_update reproduce plms.py:123,126-161 bit-for-bit in fp32."""
    from oracle import plms_ref
    B, hw = 2, 16
    n = B * 4 * hw * hw
    sched = host.make_schedule(50, host.alphas_cumprod())
    ref_s = plms_ref.make_schedule(50)
    for k in ("ddim_alphas", "ddim_alphas_prev", "ddim_sqrt_one_minus_alphas"):
        assert np.array_equal(np.asarray(sched[k], np.float64), np.asarray(ref_s[k], np.float64)), k
    x = rnd("bx", (B, 4, hw, hw))
    es = [rnd(f"be{j}", (B, 4, hw, hw)) for j in range(4)]
    e2b = torch.cat([es[0], es[1]], 0).to(DEV)
    e = torch.empty(B, 4, hw, hw, dtype=torch.float32, device=DEV)
    ops.cfg_combine(e2b, 7.5, e)
    assert torch.equal(e.cpu(), es[1] + 7.5 * (es[0] - es[1]))
    e0, e1, e2, e3 = es
    cases = [((1.0,), 1.0, e0), (host.PLMS_COEFS[0][0], host.PLMS_COEFS[0][1], (e0 + e1) / 2),
             (host.PLMS_COEFS[1][0], host.PLMS_COEFS[1][1], (3 * e0 - e1) / 2),
             (host.PLMS_COEFS[2][0], host.PLMS_COEFS[2][1], (23 * e0 - 16 * e1 + 5 * e2) / 12),
             (host.PLMS_COEFS[3][0], host.PLMS_COEFS[3][1], (55 * e0 - 59 * e1 + 37 * e2 - 9 * e3) / 24)]
    for index in (49, 33, 10, 0):
        for coefs, div, ep in cases:
            olds = es[1:len(coefs)]
            xp = torch.empty_like(e)
            sq_at, s1m, sq_ap, dirc = host.step_coefs(sched, index)
            ops.plms_update(x.to(DEV), e0.to(DEV), [o.to(DEV) for o in olds], coefs, div, sq_at, s1m, sq_ap, dirc, xp)
            # scalar sqrt with numpy float32 (IEEE correctly rounded; torch's CPU sqrt is 1 ulp off on some hosts)
            full = lambda v: torch.full((B, 1, 1, 1), float(v))
            a_t, a_prev = np.float32(sched["ddim_alphas"][index]), np.float32(sched["ddim_alphas_prev"][index])
            s1 = full(np.float32(sched["ddim_sqrt_one_minus_alphas"][index]))
            pred_x0 = (x - s1 * ep) / full(np.sqrt(a_t))
            ref = full(np.sqrt(a_prev)) * pred_x0 + full(np.sqrt(np.float32(1.0) - a_prev)) * ep
            assert torch.equal(xp.cpu(), ref), (index, coefs, float((xp.cpu() - ref).abs().max()))
