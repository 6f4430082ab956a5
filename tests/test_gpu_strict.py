"""Strict mode (gl_set_handle_option 50 on a handle created with split_weights): every matrix product takes
split-fp16 operands -- activations [hi | lo] (hi = fp16(x), lo = fp16(x - hi)), weights [Whi | Wlo] -- so that the forward
reproduces the fp32 reference within north_star's rtol 1e-3 / atol 1e-4.

Kernel level: each new operand form against the fp64 product of the UNROUNDED operands (a single-fp16 operand is >= 20 x
worse than the bounds asserted here), the hi halves bit-equal to what the plain kernels write.  Engine level: the whole tiny
UNet through the C engine against the oracle and the reference goldens; the full-size bound is in tests/test_gpu_configs.py.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from layoutllm_t2i_amd import ops, recipe
from layoutllm_t2i_amd._lib import EPI_BIAS, EPI_GEGLU, EPI_RES, EPI_SILU, init_device
from layoutllm_t2i_amd.weights import geglu_interleave, pack_conv3x3

DEV = "cuda:0"


def rnd(tag, shape, scale=1.0):
    return torch.from_numpy(recipe.normal(f"strict.{tag}", tuple(shape), 23)) * scale


def split(x):
    """fp32 -> (hi, lo) fp16 halves"""
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    return hi, lo


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


@pytest.fixture(scope="module", autouse=True)
def _init():
    init_device()


# ------------------------------------------------------------------------------------------- small pieces
def test_split_f32_is_exact():
    x = rnd("sp", (777, 320)) * 3.0 + 0.7
    y = torch.empty(777, 640, dtype=torch.float16, device=DEV)
    ops.split_f32(x.to(DEV), y)
    hi, lo = split(x)
    assert torch.equal(y[:, :320].cpu(), hi) and torch.equal(y[:, 320:].cpu(), lo)
    assert rel(y[:, :320].float() + y[:, 320:].float(), x) < 3e-7
    # small magnitudes: the lo halves (and below 6.1e-5 the hi halves) are fp16 SUBNORMALS and must survive the conversion
    xs = rnd("sps", (64, 320)) * 0.01
    ys = torch.empty(64, 640, dtype=torch.float16, device=DEV)
    ops.split_f32(xs.to(DEV), ys)
    his, los = split(xs)
    assert float(los.float().abs().max()) < 6.1e-5 and float(los.float().abs().max()) > 0
    assert torch.equal(ys[:, :320].cpu(), his) and torch.equal(ys[:, 320:].cpu(), los), "fp16 subnormals flushed by the device conversion"


@pytest.mark.parametrize("C,rows,rows2", [(320, 200, 0), (640, 70, 0), (1280, 33, 0), (320, 64, 30), (64, 256, 30)])
def test_layernorm_hi_lo_rows(C, rows, rows2):
    """gl_layernorm x_f32 bit 3: [hi | lo] rows; hi == the plain kernel's rows bitwise, hi + lo == LayerNorm in fp32."""
    B = 2
    x = rnd(f"lnx{C}", (B * rows, C)) * 2.0 + 0.3
    g, b = 1.0 + 0.1 * rnd(f"lng{C}", (C,)), 0.05 * rnd(f"lnb{C}", (C,))
    x2 = (rnd(f"lnx2{C}", (B * rows2, C)) * 1.5) if rows2 else None
    rows_out = rows + ((rows2 + 7) & ~7)
    y = torch.zeros(B * rows_out, 2 * C, dtype=torch.float16, device=DEV)
    plain = torch.zeros(B * rows_out, C, dtype=torch.float16, device=DEV)
    kw = dict(x2=x2.to(DEV), rows2=rows2) if rows2 else {}
    ops.layernorm(x.to(DEV), y, g.to(DEV), b.to(DEV), B, rows, rows_out, 0, **kw, y_lo=True)
    ops.layernorm(x.to(DEV), plain, g.to(DEV), b.to(DEV), B, rows, rows_out, 0, **kw)
    assert torch.equal(y[:, :C], plain)
    src = x.view(B, rows, C)
    if rows2:
        src = torch.cat([src, x2.view(B, rows2, C)], 1)
    want = F.layer_norm(src.double(), (C,), g.double(), b.double(), 1e-5)
    got = (y[:, :C].float() + y[:, C:].float()).view(B, rows_out, C)[:, :rows + rows2]
    r = rel(got, want)
    print(f"[layernorm hi+lo C={C}] rel_l2 vs fp64 = {r:.2e} (hi alone {rel(plain.view(B, rows_out, C)[:, :rows + rows2], want):.2e})")
    assert r < 1e-6


@pytest.mark.parametrize("force8", [0, 2])
@pytest.mark.parametrize("M,C", [(512, 320), (2048, 640), (300, 64)])
def test_geglu_hi_lo_output(M, C, force8):
    """GL_EPI_GEGLU + GL_OUT_F16_HILO (4-wave and 8-wave epilogues): hi == the plain output bitwise, hi + lo == the fp32 value."""
    xh, xl = split(rnd(f"ggx{M}{C}", (M, C)))
    a = torch.cat([xh, xl], 1).to(DEV)
    w32 = rnd(f"ggw{C}", (8 * C, C), 1 / math.sqrt(C))
    wh = w32.half()
    b = rnd(f"ggb{C}", (8 * C,), 0.1)
    wd, bd = geglu_interleave(wh).contiguous().to(DEV), geglu_interleave(b).contiguous().to(DEV)
    out = torch.empty(M, 8 * C, dtype=torch.float16, device=DEV)
    plain = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    ops.set_option(30, force8)
    try:
        ops.gemm(a, wd, out, bd, EPI_GEGLU, hilo_a=True, hilo_out=True)
        ops.gemm(a, wd, plain, bd, EPI_GEGLU, hilo_a=True)
    finally:
        ops.set_option(30, 1)
    assert torch.equal(out[:, :4 * C], plain)
    y = F.linear((xh.double() + xl.double()), wh.double(), b.double())
    want = y[:, :4 * C] * F.gelu(y[:, 4 * C:])
    r, r1 = rel(out[:, :4 * C].float() + out[:, 4 * C:].float(), want), rel(plain, want)
    print(f"[geglu hi+lo {M}x{C} g8={force8}] rel_l2 = {r:.2e} (hi alone {r1:.2e})")
    assert r < 1e-6            # (the erf approximation of the epilogue, |err| <= 1.5e-7 absolute, is the floor here; before hi and lo were
                               # taken from ONE pinned value -- common.h pin_value -- the 4-wave epilogue measured 5.8e-6: 30 outputs on rounding ties)


# ------------------------------------------------------------------------------------------- 3x3 conv with split operands
def conv_ref64(x, w, b, B, H, W, stride=1, ups=False):
    xi = x.double().view(B, H, W, -1).permute(0, 3, 1, 2)
    if ups:
        xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    y = F.conv2d(xi, w.double(), b.double(), stride=stride, padding=1)
    return y.permute(0, 2, 3, 1).reshape(-1, w.shape[0])


@pytest.mark.parametrize("force8", [0, 2])
@pytest.mark.parametrize("mode", ["s1", "s2", "up"])
@pytest.mark.parametrize("Cin,Cout,hw", [(64, 320, 16), (320, 320, 16), (128, 64, 12)])
def test_conv3x3_split_input_and_weight(mode, Cin, Cout, hw, force8):
    """gl_conv_args.in_split 2 / 3 and w_split: x = hi + lo against W (two passes), against [Whi | Wlo] (three passes), vs the fp64
    conv of the unrounded operands; stride 1 / stride 2 / nearest-2x, the 4-wave kernels and the 8-wave kernel."""
    B = 2
    stride, ups = (2 if mode == "s2" else 1), mode == "up"
    x = rnd(f"cx{Cin}{hw}", (B * hw * hw, Cin)) * 1.3 + 0.2
    xh, xl = split(x)
    w32 = rnd(f"cw{Cin}{Cout}", (Cout, Cin, 3, 3), 1 / math.sqrt(9 * Cin))
    whi = w32.half().float()
    wlo = (w32 - whi).half().float()
    b = rnd(f"cb{Cout}", (Cout,), 0.1)
    xs = torch.cat([xh, xl], 1).contiguous().to(DEV)
    ho = 2 * hw if ups else (hw + 2 - 3) // stride + 1
    out = torch.empty(B * ho * ho, Cout, dtype=torch.float32, device=DEV)
    wp_hi = pack_conv3x3(whi).to(DEV)
    wp_split = torch.cat([pack_conv3x3(whi), pack_conv3x3(wlo)], 1).contiguous().to(DEV)
    ops.set_option(30, force8)
    try:
        # two passes against the fp16 weight
        ops.conv3x3(xs, wp_hi, out, B, hw, hw, b.to(DEV), stride=stride, upsample2x=ups, in_split=2)
        want2 = conv_ref64(xh.double() + xl.double(), whi, b, B, hw, hw, stride, ups)
        r2 = rel(out, want2)
        # hi alone for comparison (plain conv on the hi half)
        ops.conv3x3(xh.contiguous().to(DEV), wp_hi, out, B, hw, hw, b.to(DEV), stride=stride, upsample2x=ups)
        r1 = rel(out, want2)
        # the split weight table read for its Whi half only
        ops.conv3x3(xs, wp_split, out, B, hw, hw, b.to(DEV), stride=stride, upsample2x=ups, in_split=2, w_split=True)
        r2w = rel(out, want2)
        ops.conv3x3(xh.contiguous().to(DEV), wp_split, out, B, hw, hw, b.to(DEV), stride=stride, upsample2x=ups, w_split=True)
        r1w = rel(out, want2)
        # three passes: + hi . Wlo
        ops.conv3x3(xs, wp_split, out, B, hw, hw, b.to(DEV), stride=stride, upsample2x=ups, in_split=3, w_split=True)
        want3 = conv_ref64(x, w32, b, B, hw, hw, stride, ups)
        r3 = rel(out, want3)
    finally:
        ops.set_option(30, 1)
    print(f"[conv split {mode} {Cin}->{Cout} @{hw} g8={force8}] 2-pass {r2:.2e} (w_split table {r2w:.2e}), hi alone {r1:.2e} / {r1w:.2e}, 3-pass vs fp32 weights {r3:.2e}")
    assert r2 < 2e-6 and r2w < 2e-6 and r3 < 2e-6 and abs(r1 - r1w) < 1e-6 and r1 > 20 * r2


@pytest.mark.parametrize("M,N,K,epi", [(8192, 320, 320, "res"), (33000, 320, 320, "res"), (32768, 960, 320, "hilo"), (8192, 2560, 320, "geglu"),
                                       (8192, 640, 2560, "res"), (2048, 1280, 1280, "res"), (2048, 1280, 5120, "res"), (512, 1280, 2560, "res"),
                                       (300, 320, 320, "res"), (4500, 1920, 640, "hilo")])
def test_gemm_three_pass_loop_vs_kwalk(M, N, K, epi):
    """Option 52: the dedicated three-pass main loop of the 8-wave kernel (one 32-wide slice of {xhi, xlo, Whi, Wlo} per ring stage, three
    MFMA groups) against the K-walk over [xhi | xlo | xhi] x [Whi | Whi | Wlo] it replaces and against the fp64 product of the unrounded
    operands: same products, other fp32 summation order.  Shapes: multi-round, ragged last tile, half-height tiles, split-K, every epilogue
    the strict forward uses (fp32 residual stream, [hi | lo] rows, GEGLU + [hi | lo])."""
    x = rnd(f"ta{M}{K}", (M, K)) * 1.7 + 0.4
    w = rnd(f"tw{N}{K}", (N, K), 1 / math.sqrt(K))
    b = rnd(f"tb{N}", (N,), 0.1)
    if epi == "geglu":
        wq, bq = geglu_interleave(w), geglu_interleave(b)
    else:
        wq, bq = w, b
    hi, lo = split(x)
    whi, wlo = split(wq)
    a = torch.cat([hi, lo], 1).to(DEV)
    w2 = torch.cat([whi, wlo], 1).contiguous().to(DEV)
    outs = []
    ops.set_option(30, 2)
    try:
        for key in (1, 0):
            ops.set_option(52, key)
            n0 = ops.gemm8_launch_count()
            if epi == "res":
                r = rnd(f"tr{M}{N}", (M, N))
                o = torch.empty(M, N, dtype=torch.float32, device=DEV)
                ops.gemm(a, w2, o, bq.to(DEV), EPI_RES, res=r.to(DEV), hilo_a=True, wsplit=2)
                got = o.cpu()
                ref = F.linear(x.double(), w.double(), b.double()) + r.double()
            elif epi == "hilo":
                o = torch.empty(M, 2 * N, dtype=torch.float16, device=DEV)
                ops.gemm(a, w2, o, bq.to(DEV), EPI_BIAS, hilo_a=True, wsplit=2, hilo_out=True)
                got = o[:, :N].float().cpu() + o[:, N:].float().cpu()
                ref = F.linear(x.double(), w.double(), b.double())
            else:
                o = torch.empty(M, N, dtype=torch.float16, device=DEV)
                ops.gemm(a, w2, o, bq.to(DEV), EPI_GEGLU, hilo_a=True, wsplit=2, hilo_out=True)
                got = o[:, :N // 2].float().cpu() + o[:, N // 2:].float().cpu()
                y = F.linear(x.double(), w.double(), b.double())
                ref = y[:, :N // 2] * F.gelu(y[:, N // 2:])
            if M >= 256:
                assert ops.gemm8_launch_count() == n0 + 1, "the 8-wave kernel was not used"
            outs.append((got, rel(got, ref)))
    finally:
        ops.set_option(52, 1)
        ops.set_option(30, 1)
    (g3, e3), (gk, ek) = outs
    d = rel(g3, gk)
    print(f"[three-pass loop {M}x{N}x{K} {epi}] vs fp64: loop {e3:.2e}, K-walk {ek:.2e}; loop vs K-walk {d:.2e}")
    bound = 2e-6 if epi == "res" else 4e-6          # [hi | lo] fp16 rows carry ~2^-22 of their own
    assert e3 < bound and ek < bound and d < bound


@pytest.mark.parametrize("mode,Cin,Cout,hw,B", [("s1", 320, 320, 64, 2), ("s1", 640, 640, 32, 8), ("s1", 1280, 1280, 8, 8), ("s1", 2560, 1280, 16, 4),
                                                ("s2", 320, 320, 64, 2), ("up", 1280, 1280, 16, 4), ("s1", 192, 64, 12, 2), ("s1", 960, 320, 24, 3)])
def test_conv_three_pass_loop_vs_kwalk(mode, Cin, Cout, hw, B):
    """Option 52 for gl_conv3x3 in_split = 3: dedicated three-pass loop vs the K-walk vs the fp64 conv of the unrounded operands
    (full-chip grids, split-K slices with odd stage counts, half-height tiles, stride 2, nearest-2x, non-power-of-two maps)."""
    stride, ups = (2 if mode == "s2" else 1), mode == "up"
    x = rnd(f"tcx{Cin}{hw}{B}", (B * hw * hw, Cin)) * 1.3 + 0.2
    xh, xl = split(x)
    w32 = rnd(f"tcw{Cin}{Cout}", (Cout, Cin, 3, 3), 1 / math.sqrt(9 * Cin))
    whi = w32.half().float()
    wlo = (w32 - whi).half().float()
    b = rnd(f"tcb{Cout}", (Cout,), 0.1)
    xs = torch.cat([xh, xl], 1).contiguous().to(DEV)
    ho = 2 * hw if ups else (hw + 2 - 3) // stride + 1
    wp = torch.cat([pack_conv3x3(whi), pack_conv3x3(wlo)], 1).contiguous().to(DEV)
    want = conv_ref64(x, w32, b, B, hw, hw, stride, ups)
    outs = []
    ops.set_option(30, 2)
    try:
        for key in (1, 0):
            ops.set_option(52, key)
            out = torch.empty(B * ho * ho, Cout, dtype=torch.float32, device=DEV)
            n0 = ops.gemm8_launch_count()
            ops.conv3x3(xs, wp, out, B, hw, hw, b.to(DEV), stride=stride, upsample2x=ups, in_split=3, w_split=True)
            assert ops.gemm8_launch_count() == n0 + 1, "the 8-wave kernel was not used"
            outs.append((out.cpu(), rel(out, want)))
    finally:
        ops.set_option(52, 1)
        ops.set_option(30, 1)
    (g3, e3), (gk, ek) = outs
    d = rel(g3, gk)
    print(f"[three-pass conv {mode} {Cin}->{Cout} @{hw} B{B}] vs fp64: loop {e3:.2e}, K-walk {ek:.2e}; loop vs K-walk {d:.2e}")
    bound = 4e-6           # fp32 accumulation over K = 9 Cin >= 11520 terms (both forms measure 2.0-2.7e-6 from Cin = 1280 on; single fp16: 3e-4)
    assert e3 < bound and ek < bound and d < bound


def test_conv3x3_split_nchw_out_conv():
    """the UNet's last conv (N = 4 -> fp32 NCHW store, 4-wave kernel) with a split input and split weights"""
    B, hw, Cin, Cout = 2, 16, 320, 4
    x = rnd("ocx", (B * hw * hw, Cin))
    xh, xl = split(x)
    w32 = rnd("ocw", (Cout, Cin, 3, 3), 1 / math.sqrt(9 * Cin))
    whi = w32.half().float()
    wlo = (w32 - whi).half().float()
    b = rnd("ocb", (Cout,), 0.1)
    wp = torch.cat([pack_conv3x3(whi), pack_conv3x3(wlo)], 1).contiguous().to(DEV)
    out = torch.empty(B, Cout, hw, hw, dtype=torch.float32, device=DEV)
    ops.conv3x3(torch.cat([xh, xl], 1).contiguous().to(DEV), wp, out, B, hw, hw, b.to(DEV), nchw_hw=hw * hw, in_split=3, w_split=True)
    want = conv_ref64(x, w32, b, B, hw, hw).view(B, hw * hw, Cout).permute(0, 2, 1).reshape(B, Cout, hw, hw)
    r = rel(out, want)
    print(f"[out conv split] rel_l2 = {r:.2e}")
    assert r < 2e-6


# ------------------------------------------------------------------------------------------- split attention
@pytest.mark.parametrize("B,rows,C,d", [(2, 1024, 320, 40), (2, 286, 320, 40), (4, 64, 1280, 160), (3, 94, 640, 80), (1, 64, 1280, 160)])
def test_qkv_projection_writes_both_vt_halves(B, rows, C, d):
    """ABI 15: a [hi | lo] fused QKV projection (three-pass, GL_OUT_F16_HILO) with the transposed tail writes vt = fp16(v)^T and vt_lo =
    fp16(v - fp16(v))^T for the V columns -- bit-equal to transposing the V thirds of the same projection's row-major [hi | lo] output
    (whole and ragged rows per sample); the q | k columns are unchanged.  Only the 8-wave kernel implements the tail: launches it cannot
    take (fewer than 256 rows) are refused, not computed some other way."""
    H = C // d
    M = B * rows
    x = rnd(f"vx{M}{C}", (M, C)) * 1.3
    w = rnd(f"vw{C}", (3 * C, C), 1 / math.sqrt(C))
    b = rnd(f"vb{C}", (3 * C,), 0.1)
    xh, xl = split(x)
    whi, wlo = split(w)
    a = torch.cat([xh, xl], 1).to(DEV)
    w2 = torch.cat([whi, wlo], 1).contiguous().to(DEV)
    ld = ops.vt_ld(rows)
    vt = torch.full((B, H, d, ld), 3.0, dtype=torch.float16, device=DEV)
    vtl = torch.full((B, H, d, ld), 3.0, dtype=torch.float16, device=DEV)
    out = torch.full((M, 6 * C), 7.0, dtype=torch.float16, device=DEV)
    if M < 256:
        with pytest.raises(Exception):
            ops.gemm(a, w2, out, b.to(DEV), EPI_BIAS, hilo_a=True, wsplit=2, hilo_out=True, vt=vt, vt_col0=2 * C, vt_rows=rows, vt_lo=vtl)
        return
    ops.set_option(30, 2)
    ops.set_option(5, 0)               # no split-K slices in the reference launch either (a launch with the transposed tail never splits K):
    ops.set_option(31, 0)              # same fp32 summation order on both sides, so the comparison is bitwise
    try:
        ref = torch.empty(M, 6 * C, dtype=torch.float16, device=DEV)
        ops.gemm(a, w2, ref, b.to(DEV), EPI_BIAS, hilo_a=True, wsplit=2, hilo_out=True)
        ops.set_option(30, 1)          # the tail itself needs no forcing: the dispatcher sends it to the 8-wave kernel
        ops.gemm(a, w2, out, b.to(DEV), EPI_BIAS, hilo_a=True, wsplit=2, hilo_out=True, vt=vt, vt_col0=2 * C, vt_rows=rows, vt_lo=vtl)
    finally:
        ops.set_option(30, 1)
        ops.set_option(5, -1)
        ops.set_option(31, 200)
    r = ref.cpu()
    o = out.cpu()
    assert torch.equal(o[:, :2 * C], r[:, :2 * C]) and torch.equal(o[:, 3 * C:5 * C], r[:, 3 * C:5 * C])          # q | k, hi and lo
    want_h = r[:, 2 * C:3 * C].view(B, rows, H, d).permute(0, 2, 3, 1)
    want_l = r[:, 5 * C:6 * C].view(B, rows, H, d).permute(0, 2, 3, 1)
    assert torch.equal(vt.cpu()[..., :rows], want_h) and torch.equal(vtl.cpu()[..., :rows], want_l)
    # and a plain output must not be given a residual tail
    with pytest.raises(Exception):
        ops.gemm(xh.contiguous().to(DEV), whi.contiguous().to(DEV), torch.empty(M, 3 * C, dtype=torch.float16, device=DEV), b.to(DEV), EPI_BIAS, vt=vt,
                 vt_col0=2 * C, vt_rows=rows, vt_lo=vtl)


@pytest.mark.parametrize("d,H,Nq,Nk,B", [(40, 8, 256, 256, 2), (40, 8, 300, 286, 1), (80, 8, 128, 77, 2), (160, 8, 64, 94, 2), (160, 8, 256, 256, 1),
                                         (16, 4, 256, 286, 1), (64, 2, 130, 10, 1),
                                         # Nq >= 512 at d = 32 / 40 / 48: the software-pipelined kernel (round 6): whole tiles, ragged query and key
                                         # ranges, one / two / many key tiles, the 77-key text context, the fuser's N + 30 keys
                                         (40, 8, 1024, 1024, 1), (40, 2, 640, 700, 2), (32, 4, 512, 130, 1), (48, 2, 512, 64, 1), (40, 1, 520, 77, 1),
                                         (40, 8, 4096, 4126, 1), (48, 3, 777, 1000, 1), (32, 2, 1024, 33, 2)])
def test_split_attention(d, H, Nq, Nk, B):
    """gl_attention with q_lo / k_lo / vt_lo: three-pass QK^T and P.V on hi + lo operands vs fp64 attention of the unrounded q, k, v;
    out_lo holds the residual of the output."""
    C = H * d
    q, k, v = rnd(f"aq{d}{Nq}", (B * Nq, C)) * 1.2, rnd(f"ak{d}{Nk}", (B * Nk, C)) * 1.2, rnd(f"av{d}{Nk}", (B * Nk, C))
    scale = d ** -0.5
    qh, ql = split(q)
    kh, kl = split(k)
    vh, vl = split(v)
    ld = ops.vt_ld(Nk)
    vt_h = torch.full((B, H, d, ld), float("nan"), dtype=torch.float16, device=DEV)
    vt_l = torch.full((B, H, d, ld), float("nan"), dtype=torch.float16, device=DEV)
    ops.transpose_v(vh.to(DEV), Nk * C, C, vt_h, B, H, d, Nk)
    ops.transpose_v(vl.to(DEV), Nk * C, C, vt_l, B, H, d, Nk)
    vt_h[..., Nk:] = float("nan")          # pad keys: never read
    vt_l[..., Nk:] = float("nan")
    out = torch.empty(B * Nq, 2 * C, dtype=torch.float16, device=DEV)
    ops.attention(qh.to(DEV), Nq * C, C, kh.to(DEV), Nk * C, C, vt_h, out, Nq * 2 * C, 2 * C, B, H, d, Nq, Nk, scale,
                  q_lo=ql.to(DEV), k_lo=kl.to(DEV), vt_lo=vt_l, out_lo=out[:, C:])
    f = lambda a, b_: (a.double() + b_.double())
    Q = f(qh, ql).view(B, -1, H, d).transpose(1, 2)
    K = f(kh, kl).view(B, -1, H, d).transpose(1, 2)
    V = f(vh, vl).view(B, -1, H, d).transpose(1, 2)
    want = (torch.softmax(Q @ K.transpose(-1, -2) * scale, -1) @ V).transpose(1, 2).reshape(B * Nq, C)
    got = out[:, :C].float() + out[:, C:].float()
    plain = torch.empty(B * Nq, C, dtype=torch.float16, device=DEV)
    vt0 = torch.zeros_like(vt_h)
    ops.transpose_v(vh.to(DEV), Nk * C, C, vt0, B, H, d, Nk)
    ops.attention(qh.to(DEV), Nq * C, C, kh.to(DEV), Nk * C, C, vt0, plain, Nq * C, C, B, H, d, Nq, Nk, scale)
    r, r1 = rel(got, want), rel(plain, want)
    print(f"[split attention d={d} Nq={Nq} Nk={Nk}] rel_l2 = {r:.2e} (single-fp16 kernel {r1:.2e})")
    assert torch.isfinite(out).all() and r < 1e-6 and r1 > 20 * r


# ------------------------------------------------------------------------------------------- engine
def _tiny_engine(split_weights):
    import dataclasses
    from layoutllm_t2i_amd.arch import TINY
    from layoutllm_t2i_amd.engine import UNetEngine
    from layoutllm_t2i_amd.weights import pack_state_dict
    cfg = dataclasses.replace(TINY, split_weights=split_weights)
    sd = recipe.state_dict(TINY, 0)
    P = pack_state_dict(sd, cfg, DEV, recipe.sd_first_conv(TINY, 0))
    return UNetEngine(P), cfg, sd


def test_strict_mode_needs_split_weights():
    from layoutllm_t2i_amd._lib import HipLibraryError
    from layoutllm_t2i_amd.arch import TINY
    eng, cfg, sd = _tiny_engine(False)
    inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(TINY, 2, 16, n_boxes=4, seed=99).items()}
    eng.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], 16)
    eng.set_option(50, 1)
    with pytest.raises(HipLibraryError):
        eng.forward(inp["x"].to(DEV), 481.0, 1.0, False, 1)


@pytest.mark.parametrize("fuser_scale,sd_conv", [(1.0, False), (0.0, True)])
def test_tiny_unet_strict_vs_oracle_and_default(fuser_scale, sd_conv):
    """The tiny UNet (fp32 recipe weights, NOT fp16-representable) through the C engine: a split_weights handle in its default mode
    equals the compact handle; strict with the third pass sits within north_star's tolerance of the fp32 oracle on the unrounded
    weights; strict without the third pass (key 51 = 0) within it of the oracle on fp16-rounded weight matrices ... except for the
    q projections' folded scale, whose residual only the third pass carries."""
    from layoutllm_t2i_amd.arch import TINY
    from oracle import unet_ref
    eng0, _, sd = _tiny_engine(False)
    eng, cfg, _ = _tiny_engine(True)
    inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(TINY, 2, 16, n_boxes=4, seed=99).items()}
    for e_ in (eng0, eng):
        e_.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], 16)
    x = inp["x"].to(DEV)
    a = eng0.forward(x, 481.0, fuser_scale, sd_conv, 1).clone()
    b = eng.forward(x, 481.0, fuser_scale, sd_conv, 1).clone()
    # the default mode of the split table differs from the compact table only where the fused FeedForward is not used
    assert rel(b, a) < 2e-4, rel(b, a)
    osd = {k: torch.from_numpy(np.asarray(v)).float() for k, v in sd.items()}
    if sd_conv:
        sdc = recipe.sd_first_conv(TINY, 0)
        osd = dict(osd)
        osd["input_blocks.0.0.weight"] = torch.from_numpy(np.asarray(sdc["weight"])).float()
        osd["input_blocks.0.0.bias"] = torch.from_numpy(np.asarray(sdc["bias"])).float()
    with torch.no_grad():
        ref = unet_ref.unet_forward(osd, TINY, inp["x"], torch.tensor([481, 481]), inp["context"], inp["relations"], inp["boxes"], inp["masks"],
                                    inp["positive_embeddings"], fuser_scale=fuser_scale)
    outside = lambda o: float(((o.float().cpu() - ref).abs() > 1e-4 + 1e-3 * ref.abs()).float().mean())
    eng.set_option(50, 1)
    s3 = eng.forward(x, 481.0, fuser_scale, sd_conv, 1).clone()
    eng.use_graphs = False
    s3e = eng.forward(x, 481.0, fuser_scale, sd_conv, 1).clone()
    eng.use_graphs = True
    assert torch.equal(s3, s3e), "graph replay == eager in strict mode"
    print(f"[tiny strict fuser={fuser_scale} sd_conv={sd_conv}] default rel_l2 {rel(b, ref):.2e} outside {outside(b) * 100:.1f} % | "
          f"strict (3 passes) rel_l2 {rel(s3, ref):.2e} outside {outside(s3) * 100:.2f} %")
    assert rel(s3, ref) < 5e-5 and outside(s3) < 0.01
    # the strict conditioning hoists are computed lazily (first strict forward after a default-mode gl_set_conditioning, as above): conditioning
    # set WHILE in strict mode, and key 51 changed with / without a new conditioning, must give the same bits
    eng.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], 16)
    assert torch.equal(eng.forward(x, 481.0, fuser_scale, sd_conv, 1), s3), "hoists at conditioning time == lazy hoists"
    eng.set_option(51, 0)
    s2_lazy = eng.forward(x, 481.0, fuser_scale, sd_conv, 1).clone()
    eng.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], 16)
    assert torch.equal(eng.forward(x, 481.0, fuser_scale, sd_conv, 1), s2_lazy), "key 51 changed: the hoists are recomputed without a new conditioning"
    eng.set_option(51, 1)
    assert torch.equal(eng.forward(x, 481.0, fuser_scale, sd_conv, 1), s3)
    eng.set_option(50, 0)
    b2 = eng.forward(x, 481.0, fuser_scale, sd_conv, 1).clone()
    assert torch.equal(b2, b), "switching strict off restores the default results"
