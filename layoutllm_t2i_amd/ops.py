"""Host-side wrappers: torch tensors (device memory + streams only) -> raw pointers -> C ABI.

Each function launches exactly one HIP kernel from libgligen_hip.so on torch's current stream.
There is no eager/PyTorch fallback: a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_GATE_RES, EPI_GEGLU, EPI_RES, EPI_ROWBIAS, EPI_SILU, OUT_F16_HILO, OUT_F16_ROWMAJOR, OUT_F32_NCHW,
                   OUT_F32_ROWMAJOR, AttnArgs, ConvArgs, GemmArgs, GnArgs, check)

F16 = torch.float16
F32 = torch.float32


_DEV = None


def _stream() -> int:
    """Stream of the device the last checked tensor lives on (``_req`` records it and asserts that all tensors of a
    call share it).  A tensor on cuda:N while another device is current therefore launches on cuda:N's current stream."""
    dev = _DEV if _DEV is not None else torch.cuda.current_device()
    if dev != torch.cuda.current_device():
        raise _lib.HipLibraryError(f"tensors live on cuda:{dev} but cuda:{torch.cuda.current_device()} is current: wrap the "
                                   "call in torch.cuda.device(...) (kernels launch on the current device)")
    _lib.init_device(dev)
    return torch.cuda.current_stream(dev).cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str, align: int = 16) -> None:
    if not t.is_cuda:
        raise _lib.HipLibraryError(f"{name}: tensor is not on the GPU (the HIP path has no CPU fallback)")
    global _DEV
    _DEV = t.device.index
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.data_ptr() % align:
        raise ValueError(f"{name}: pointer not {align}-byte aligned")


def _rows(t: torch.Tensor, name: str):
    """2-D row view: returns (rows, cols, row_stride) requiring unit column stride."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D tensor with contiguous columns, got shape {tuple(t.shape)} stride {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


_WS = {}
WORKSPACE_BYTES = 96 << 20


def _workspace(device) -> torch.Tensor:
    """Per-device split-K scratch (fp32 partial tiles); allocated once so captured graphs stay valid."""
    key = (device.type, device.index)
    t = _WS.get(key)
    if t is None:
        t = torch.empty(WORKSPACE_BYTES // 4, dtype=F32, device=device)
        _WS[key] = t
    return t


def _fill_epilogue(g: GemmArgs, epi, out, N_out, bias, res, gate, rowbias, rows_per_sample, nchw_hw, out16=None, hilo_out=False):
    """``out`` fp16 [M, N] (plain), fp32 [M, N] (residual stream; ``out16`` = optional fp16 copy) or, with ``nchw_hw``,
    fp32 NCHW; ``hilo_out``: fp16 [M, 2N] = [hi | lo] (GL_OUT_F16_HILO).  ``res`` may be fp16 or fp32 (residual stream)."""
    ws = _workspace(out.device)
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    g.bias = _ptr(bias)
    if bias is not None:
        _req(bias, F32, "bias")
    g.epi = epi
    if nchw_hw:
        _req(out, F32, "out")
        g.out_mode, g.hw = OUT_F32_NCHW, nchw_hw
        g.out, g.ldc = out.data_ptr(), 0
    else:
        f32out = out.dtype == F32
        _req(out, F32 if f32out else F16, "out", 16 if f32out else 8)
        r, c, ld = _rows(out, "out")
        if c != (2 * N_out if hilo_out else N_out):
            raise ValueError(f"out has {c} columns, expected {2 * N_out if hilo_out else N_out}")
        if hilo_out and f32out:
            raise ValueError("hilo_out writes fp16")
        g.out_mode, g.hw = (OUT_F16_HILO if hilo_out else OUT_F32_ROWMAJOR if f32out else OUT_F16_ROWMAJOR), 0
        g.out, g.ldc = out.data_ptr(), ld
        if out16 is not None:
            if not f32out:
                raise ValueError("out16 (fp16 copy) only goes with an fp32 out")
            _req(out16, F16, "out16", 8)
            r2, c2, ld2 = _rows(out16, "out16")
            if (r2, c2) != (r, c):
                raise ValueError("out16 shape mismatch")
            g.out2, g.ldc2 = out16.data_ptr(), ld2
    if res is not None:
        g.res_f32 = int(res.dtype == F32)
        _req(res, F32 if g.res_f32 else F16, "res", 16 if g.res_f32 else 8)
        g.res, g.ldres = res.data_ptr(), _rows(res, "res")[2]
    if gate is not None:
        _req(gate, F32, "gate", 4)
        g.gate = gate.data_ptr()
    if rowbias is not None:
        g.rowbias_f32 = int(rowbias.dtype == F32)
        _req(rowbias, F32 if g.rowbias_f32 else F16, "rowbias", 16 if g.rowbias_f32 else 8)
        g.rowbias, g.ld_rowbias, g.rows_per_sample = rowbias.data_ptr(), _rows(rowbias, "rowbias")[2], rows_per_sample


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias=None, epi: int = EPI_BIAS, res=None, gate=None,
         rowbias=None, rows_per_sample: int = 0, a2: Optional[torch.Tensor] = None, nchw_hw: int = 0, out16=None,
         vt: Optional[torch.Tensor] = None, vt_col0: int = 0, vt_rows: int = 0, hilo_a: bool = False, hilo_out: bool = False,
         wsplit: int = 0, vt_lo: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = [a | a2] @ w.T (+ epilogue).  a [M, K1], a2 [M, K2] (optional), w [N, K1+K2] fp16.
    ``hilo_a``: a is [M, 2K] = [hi | lo] of a split-fp16 activation and w [N, K] is used for both halves (gl_gemm_args.kwrap);
    ``hilo_out``: out is fp16 [M, 2N] = [hi | lo] of the result (GL_OUT_F16_HILO).
    ``wsplit``: w is [N, 2K] = [Whi | Wlo] rows of a split-fp16 weight: 1 = use Whi only (K or, with hilo_a, 2K products), 2 = with hilo_a the
    three-pass product xhi.Whi + xlo.Whi + xhi.Wlo (K index 3K; the third A segment is the hi half again, passed as the second source).
    ``vt`` [B, H, d, ldvt] fp16: columns [vt_col0, N) are written there as gl_attention's V^T operand (row m = sample
    m // vt_rows, key m % vt_rows) instead of to ``out`` (fused QKV projection)."""
    _req(a, F16, "a")
    _req(w, F16, "w")
    M, K1, lda = _rows(a, "a")
    N, K, ldw = _rows(w, "w")
    if ldw != K:
        raise ValueError("w must be contiguous")
    g = GemmArgs()
    g.a, g.lda = a.data_ptr(), lda
    if wsplit:
        if K % 2 or (wsplit == 2 and (not hilo_a or a2 is not None)):
            raise ValueError("wsplit: w must be [N, 2K]; the three-pass form needs hilo_a and no second source")
        K //= 2
        g.ldw = 2 * K
    if wsplit == 2:
        if K1 != 2 * K:
            raise ValueError(f"hilo_a: a has {K1} columns, expected 2 x K = {2 * K}")
        g.kwrap = K
        g.a2, g.lda2, g.ksplit = a.data_ptr(), lda, 2 * K
        K = 3 * K
    elif a2 is not None:
        _req(a2, F16, "a2")
        M2, K2, lda2 = _rows(a2, "a2")
        if M2 != M or K1 + K2 != K:
            raise ValueError("a2 shape mismatch")
        g.a2, g.lda2, g.ksplit = a2.data_ptr(), lda2, K1
    elif hilo_a:
        if K1 != 2 * K:
            raise ValueError(f"hilo_a: a has {K1} columns, expected 2 x K = {2 * K}")
        g.kwrap = K
        if not wsplit:
            g.ldw = K
        K = 2 * K
    elif K1 != K:
        raise ValueError(f"a has K={K1}, w has K={K}")
    g.w = w.data_ptr()
    g.M, g.N, g.K = M, N, K
    _fill_epilogue(g, epi, out, N // 2 if epi == EPI_GEGLU else N, bias, res, gate, rowbias, rows_per_sample, nchw_hw, out16, hilo_out)
    if vt is not None:
        _req(vt, F16, "vt")
        if vt.dim() != 4 or not vt.is_contiguous():
            raise ValueError("vt must be a contiguous [B, H, d, ldvt] tensor")
        g.vt, g.vt_col0, g.vt_rows = vt.data_ptr(), vt_col0, vt_rows
        g.vt_H, g.vt_d, g.vt_ld = vt.shape[1], vt.shape[2], vt.shape[3]
        if vt_lo is not None:              # ``hilo_out``: the fp16 residual of the V^T tail, same layout as ``vt`` (ABI 15)
            _req(vt_lo, F16, "vt_lo")
            if tuple(vt_lo.shape) != tuple(vt.shape) or not vt_lo.is_contiguous():
                raise ValueError("vt_lo must have vt's shape")
            g.vt_lo = vt_lo.data_ptr()
    check(_lib.lib().gl_gemm(C.byref(g), _stream()), "gl_gemm")
    return out


def conv3x3(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, B: int, Hin: int, Win: int, bias=None,
            stride: int = 1, upsample2x: bool = False, epi: int = EPI_BIAS, res=None, rowbias=None,
            rows_per_sample: int = 0, nchw_hw: int = 0, n_valid: Optional[int] = None, out16=None, in_split: int = 0,
            w_split: bool = False) -> torch.Tensor:
    """x [B*Hin*Win, Cin] fp16 NHWC; w [Cout, 9*Cin] fp16 (tap-major, channel-minor).
    ``in_split`` = 2: x is [B*Hin*Win, 2 Cin] = [hi | lo] pixel rows of a split-fp16 activation, both halves against the same weight;
    3 (with ``w_split``): plus the third pass hi.Wlo.  ``w_split``: w is [Cout, 18 Cin] = rows [Whi | Wlo] (gl_conv_args)."""
    _req(x, F16, "x")
    _req(w, F16, "w")
    rows, Cin, ldx = _rows(x, "x")
    if ldx != Cin or rows != B * Hin * Win:
        raise ValueError("x must be a contiguous [B*Hin*Win, Cin] matrix")
    if in_split:
        Cin //= 2
    N, K, _ = _rows(w, "w")
    if K != (18 if w_split else 9) * Cin:
        raise ValueError("w must be [Cout, 9*Cin] ([Cout, 18*Cin] with w_split)")
    if upsample2x:
        Hout, Wout = 2 * Hin, 2 * Win
    else:
        Hout, Wout = (Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1
    a = ConvArgs()
    a.inp = x.data_ptr()
    a.B, a.Hin, a.Win, a.Cin, a.Hout, a.Wout = B, Hin, Win, Cin, Hout, Wout
    a.stride, a.upsample2x = stride, int(upsample2x)
    a.in_split, a.w_split = int(in_split), int(bool(w_split))
    a.g.w = w.data_ptr()
    a.g.N = N if n_valid is None else n_valid
    _fill_epilogue(a.g, epi, out, a.g.N, bias, res, None, rowbias, rows_per_sample, nchw_hw, out16)
    check(_lib.lib().gl_conv3x3(C.byref(a), _stream()), "gl_conv3x3")
    return out


def vt_ld(Nk: int) -> int:
    """Row stride (keys) for a V^T buffer: Nk rounded up to the 64-key tile, plus one tile when that is a multiple of
    512 keys -- a 1 KiB-multiple row stride maps the d rows of every V^T tile onto the same L1 sets / channels
    (measured 307 vs 277 us at N = 4096, d = 40)."""
    n = (Nk + 63) // 64 * 64
    return n + 64 if n % 512 == 0 else n


def transpose_v(v: torch.Tensor, v_bstride: int, ldv: int, vt: torch.Tensor, B: int, H: int, d: int, Nk: int):
    """v: fp16 view whose element (b, key, h*d + c) sits at v.data_ptr + b*v_bstride + key*ldv + h*d + c.
    vt: [B, H, d, ldvt] fp16 contiguous, ldvt >= roundup(Nk, 64)."""
    _req(v, F16, "v", 2)
    _req(vt, F16, "vt")
    ldvt = vt.shape[-1]
    check(_lib.lib().gl_transpose_v(v.data_ptr(), v_bstride, ldv, vt.data_ptr(), ldvt, B, H, d, Nk, _stream()),
          "gl_transpose_v")
    return vt


def attention(q: torch.Tensor, q_bstride: int, ldq: int, k: torch.Tensor, k_bstride: int, ldk: int, vt: torch.Tensor,
              out: torch.Tensor, o_bstride: int, ldo: int, B: int, H: int, d: int, Nq: int, Nk: int, scale: float,
              q_prescaled: bool = False, q_lo=None, k_lo=None, vt_lo=None, out_lo=None):
    """``q_prescaled``: scale * log2(e) is already folded into q (weights.Q_FOLD does that to the packed q projections).
    ``q_lo`` / ``k_lo`` / ``vt_lo`` (all three, same layouts as q / k / vt): the fp16 residuals of split-fp16 operands -> the three-pass
    split attention kernel; ``out_lo`` then receives fp16(O - fp16(O))."""
    _req(q, F16, "q")
    _req(k, F16, "k")
    _req(vt, F16, "vt")
    _req(out, F16, "out", 8)
    a = AttnArgs()
    a.q, a.q_bstride, a.ldq = q.data_ptr(), q_bstride, ldq
    a.k, a.k_bstride, a.ldk = k.data_ptr(), k_bstride, ldk
    a.vt, a.ldvt = vt.data_ptr(), vt.shape[-1]
    a.out, a.o_bstride, a.ldo = out.data_ptr(), o_bstride, ldo
    a.B, a.H, a.d, a.Nq, a.Nk = B, H, d, Nq, Nk
    a.scale = scale
    a.q_prescaled = int(q_prescaled)
    for t_, n_ in ((q_lo, "q_lo"), (k_lo, "k_lo"), (vt_lo, "vt_lo"), (out_lo, "out_lo")):
        if t_ is not None:
            _req(t_, F16, n_, 8)
    a.q_lo, a.k_lo, a.vt_lo, a.out_lo = _ptr(q_lo), _ptr(k_lo), _ptr(vt_lo), _ptr(out_lo)
    check(_lib.lib().gl_attention(C.byref(a), _stream()), "gl_attention")
    return out


def gn_nchunk(HW: int) -> int:
    """pixel chunks per sample for the GroupNorm partial sums: 64 for the UNet's maps (<= 64x64), more for the
    VAE decoder's large maps so the statistics pass still fills the chip"""
    if HW <= 4096:
        return max(1, min(64, HW // 4))
    return min(512, HW // 512)


def groupnorm(x1: torch.Tensor, x2: Optional[torch.Tensor], B: int, HW: int, gamma: torch.Tensor, beta: torch.Tensor,
              eps: float, silu: bool, out: torch.Tensor, partial: torch.Tensor, out_lo: Optional[torch.Tensor] = None,
              raw: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm(32) of the channel concat [x1 | x2] (x2 may be None); x* are [B*HW, C*], both fp16 or both fp32 (the residual
    stream).  ``out`` fp16 [B*HW, C] (may be a column view of a wider buffer); ``out_lo`` (optional, same row stride): the
    fp16 residual of the normalised rows; ``raw`` (optional) fp16 [B*HW, >= 2C]: the INPUT as [hi | lo] (gl_groupnorm_ex)."""
    xf32 = x1.dtype == F32
    _req(x1, F32 if xf32 else F16, "x1")
    C1 = x1.shape[-1]
    C2 = 0
    if x2 is not None:
        _req(x2, F32 if xf32 else F16, "x2")
        C2 = x2.shape[-1]
    _req(gamma, F32, "gamma")
    _req(beta, F32, "beta")
    for t_, n_ in ((out, "out"), (out_lo, "out_lo"), (raw, "raw")):
        if t_ is not None and (t_.dtype != F16 or not t_.is_cuda or t_.stride(-1) != 1 or t_.data_ptr() % 16):
            raise _lib.HipLibraryError(f"{n_}: fp16 GPU rows, 16-byte aligned")
    _req(partial, F32, "partial")
    nchunk = gn_nchunk(HW)
    if partial.numel() < B * nchunk * 64:
        raise ValueError("partial buffer too small")
    a = GnArgs()
    a.x1, a.C1, a.x2, a.C2, a.x_f32, a.B, a.HW = x1.data_ptr(), C1, _ptr(x2), C2, int(xf32), B, HW
    a.gamma, a.beta, a.eps, a.silu = gamma.data_ptr(), beta.data_ptr(), eps, int(silu)
    a.out, a.ldo = out.data_ptr(), out.stride(0)
    if out_lo is not None:
        if out_lo.stride(0) != out.stride(0):
            raise ValueError("out_lo must have out's row stride")
        a.out_lo = out_lo.data_ptr()
    if raw is not None:
        a.raw, a.ldraw = raw.data_ptr(), raw.stride(0)
    a.partial, a.nchunk = partial.data_ptr(), nchunk
    check(_lib.lib().gl_groupnorm_ex(C.byref(a), _stream()), "gl_groupnorm_ex")
    return out


def layernorm(x: torch.Tensor, y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, B: int, rows_in: int,
              rows_out: Optional[int] = None, row_off: int = 0, eps: float = 1e-5, stats: Optional[torch.Tensor] = None,
              x2: Optional[torch.Tensor] = None, rows2: int = 0, y_lo: bool = False) -> torch.Tensor:
    """``y_lo``: y is fp16 [B*rows_out, >= 2C] and receives [hi | lo] rows (gl_layernorm x_f32 bit 3; fp32 inputs only).
    x [B*rows_in, C] fp16 or fp32 (residual stream) -> y fp16 (or fp32: the dtype of ``y`` decides) rows b*rows_out + row_off + i (y is [B*rows_out, C]);
    ``stats`` (optional fp32 [B*rows_in, 2]) receives (mean, rstd) per row.  ``x2`` (fp16 [B*rows2, C]): a second source
    whose rows follow x's rows inside every sample's block of y ([x ; objs] in one launch)."""
    xf32 = x.dtype == F32
    yf32 = y.dtype == F32
    _req(x, F32 if xf32 else F16, "x")
    _req(y, F32 if yf32 else F16, "y")
    _req(gamma, F32, "gamma")
    _req(beta, F32, "beta")
    _, Cc, ldx = _rows(x, "x")
    _, _, ldy = _rows(y, "y")
    rows_out = rows_in if rows_out is None else rows_out
    if stats is not None:
        _req(stats, F32, "stats", 8)
        if stats.numel() < 2 * B * rows_in:
            raise ValueError("stats buffer too small")
    ldx2 = 0
    x2f32 = False
    if x2 is not None:
        x2f32 = x2.dtype == F32
        _req(x2, F32 if x2f32 else F16, "x2")
        ldx2 = _rows(x2, "x2")[2]
    check(_lib.lib().gl_layernorm(x.data_ptr(), ldx, int(xf32) | (2 if yf32 else 0) | (4 if x2f32 else 0) | (8 if y_lo else 0), y.data_ptr(), ldy, gamma.data_ptr(), beta.data_ptr(), B, rows_in,
                                  rows_out, row_off, Cc, eps, _ptr(stats), _ptr(x2), ldx2, rows2, _stream()), "gl_layernorm")
    return y


def rela_pool(hid, B, H, W, Cc, rects, nvalid, poison, max_objs, feat, ln_gamma=None, ln_beta=None, ln_out=None, slots=0):
    """feat[b, i] = mean of hid over box i; with ``ln_out`` also LayerNorm(feat) (fused norm1 of rela_fuse).  ``slots``: rows per sample of
    feat / ln_out (0 = max_objs; every nvalid[b] must be <= slots)."""
    _req(hid, F16, "hid")
    _req(feat, F16, "feat")
    for t, n in ((rects, "rects"), (nvalid, "nvalid"), (poison, "poison")):
        _req(t, torch.int32, n, 4)
    if ln_out is not None:
        _req(ln_out, F16, "ln_out")
        _req(ln_gamma, F32, "ln_gamma")
        _req(ln_beta, F32, "ln_beta")
    check(_lib.lib().gl_rela_pool(hid.data_ptr(), B, H, W, Cc, rects.data_ptr(), nvalid.data_ptr(), poison.data_ptr(),
                                  max_objs, int(slots), feat.data_ptr(), _ptr(ln_gamma), _ptr(ln_beta), _ptr(ln_out), _stream()), "gl_rela_pool")
    return feat


def split_f32(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """fp32 rows [rows, C] -> fp16 [rows, 2C] = [hi | lo] (gl_split_f32)."""
    _req(x, F32, "x")
    _req(y, F16, "y")
    rows, Cc, ldx = _rows(x, "x")
    _, c2, ldy = _rows(y, "y")
    if c2 != 2 * Cc:
        raise ValueError("y must be [rows, 2C]")
    check(_lib.lib().gl_split_f32(x.data_ptr(), ldx, rows, Cc, y.data_ptr(), ldy, _stream()), "gl_split_f32")
    return y


def layernorm_stats(x: torch.Tensor, stats: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """per-row (mean, rstd) of fp32 rows [rows, C] -> stats fp32 [rows, 2] (gl_layernorm_stats: gl_layernorm's statistics alone)."""
    _req(x, F32, "x")
    _req(stats, F32, "stats", 8)
    rows, Cc, ldx = _rows(x, "x")
    if stats.numel() < 2 * rows:
        raise ValueError("stats buffer too small")
    check(_lib.lib().gl_layernorm_stats(x.data_ptr(), ldx, rows, Cc, eps, stats.data_ptr(), _stream()), "gl_layernorm_stats")
    return stats


def rela_pool_ln3(x, ln3_stats, ln3_gamma, ln3_beta, B, H, W, Cc, rects, nvalid, poison, max_objs, feat, ln_gamma=None, ln_beta=None, ln_out=None, slots=0):
    """rela_pool on the fp32 stream: feat[b, i] = mean over box i of LN3(x), LN3 re-evaluated in fp32 from ``ln3_stats`` (gl_rela_pool_ln3)."""
    _req(x, F32, "x")
    _req(ln3_stats, F32, "ln3_stats", 8)
    _req(ln3_gamma, F32, "ln3_gamma")
    _req(ln3_beta, F32, "ln3_beta")
    _req(feat, F16, "feat")
    for t, n in ((rects, "rects"), (nvalid, "nvalid"), (poison, "poison")):
        _req(t, torch.int32, n, 4)
    if ln_out is not None:
        _req(ln_out, F16, "ln_out")
        _req(ln_gamma, F32, "ln_gamma")
        _req(ln_beta, F32, "ln_beta")
    check(_lib.lib().gl_rela_pool_ln3(x.data_ptr(), ln3_stats.data_ptr(), ln3_gamma.data_ptr(), ln3_beta.data_ptr(), B, H, W, Cc, rects.data_ptr(),
                                      nvalid.data_ptr(), poison.data_ptr(), max_objs, int(slots), feat.data_ptr(), _ptr(ln_gamma), _ptr(ln_beta), _ptr(ln_out),
                                      _stream()), "gl_rela_pool_ln3")
    return feat


def rela_merge(x, hid, f, B, H, W, Cc, rects, nvalid, poison, max_objs, y, ln_stats=None, gamma=None, beta=None,
               ln2_gamma=None, ln2_beta=None, ln2_out=None, slots=0):
    """y = 0.5 * (x + hid + (1/max_objs) sum_i 1[p in rect_i] f_i); x / y fp16 or fp32 (same dtype).  With ``ln_stats``
    (+ gamma, beta) hid = LayerNorm(x) is re-evaluated in fp32 from the stored (mean, rstd) instead of read from ``hid``.
    With ``ln2_out`` (fp32 stream + ln_stats form) the launch also writes LayerNorm(y; ln2_gamma, ln2_beta) in fp16."""
    xf32 = x.dtype == F32
    for t, n in ((x, "x"), (y, "y")):
        _req(t, F32 if xf32 else F16, n)
    _req(f, F16, "f")
    if ln_stats is None:
        _req(hid, F16, "hid")
    else:
        for t, n in ((ln_stats, "ln_stats"), (gamma, "gamma"), (beta, "beta")):
            _req(t, F32, n, 8)
    if ln2_out is not None:
        _req(ln2_out, F16, "ln2_out")
        for t, n in ((ln2_gamma, "ln2_gamma"), (ln2_beta, "ln2_beta")):
            _req(t, F32, n, 8)
    check(_lib.lib().gl_rela_merge(x.data_ptr(), int(xf32), _ptr(hid), _ptr(ln_stats), _ptr(gamma), _ptr(beta), f.data_ptr(), B, H, W,
                                   Cc, rects.data_ptr(), nvalid.data_ptr(), poison.data_ptr(), max_objs, int(slots), y.data_ptr(), _ptr(ln2_gamma),
                                   _ptr(ln2_beta), _ptr(ln2_out), _stream()),
          "gl_rela_merge")
    return y


def ff_fused_supported(Cc: int) -> bool:
    return bool(_lib.lib().gl_ff_fused_supported(int(Cc)))


def ff_fused_applicable(Cc: int, M: int) -> bool:
    """the engine's own rule for taking the fused FeedForward (gl_ff_fused_applicable)"""
    return bool(_lib.lib().gl_ff_fused_applicable(int(Cc), int(M)))


def ff_fused(x, w1, b1, w2, b2, res, out, gate=None, hilo_out: bool = False):
    """out = res (+ | gate *) (GEGLU(x . w1^T + b1) . w2^T + b2) in one launch (gl_ff_fused; attention.py:38-62).
    x fp16 [M, C]; w1 / b1 the packed GEGLU operands [8C, C] / [8C]; w2 [C, 4C]; res fp32 or fp16 [M, C]; out fp16 or fp32."""
    _req(x, F16, "x")
    _req(w1, F16, "w1")
    _req(w2, F16, "w2")
    _req(b1, F32, "b1")
    _req(b2, F32, "b2")
    M, Cc = x.shape
    if w1.shape != (8 * Cc, Cc) or w2.shape != (Cc, 4 * Cc) or not w1.is_contiguous() or not w2.is_contiguous():
        raise ValueError("w1 must be [8C, C] (packed GEGLU layout) and w2 [C, 4C], contiguous")
    if res.dtype not in (F16, F32) or out.dtype not in (F16, F32) or not res.is_cuda or not out.is_cuda:
        raise _lib.HipLibraryError("res / out: fp16 or fp32 GPU tensors")
    a = _lib.FFArgs()
    a.x, a.ldx = x.data_ptr(), x.stride(0)
    a.w1, a.b1, a.w2, a.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    a.res, a.ldres, a.res_f32 = res.data_ptr(), res.stride(0), int(res.dtype == F32)
    a.gate = _ptr(gate)
    a.out, a.ldc, a.out_mode = out.data_ptr(), out.stride(0), (OUT_F16_HILO if hilo_out else OUT_F32_ROWMAJOR if out.dtype == F32 else OUT_F16_ROWMAJOR)
    a.M, a.C = M, Cc
    check(_lib.lib().gl_ff_fused(C.byref(a), _stream()), "gl_ff_fused")
    return out


def posnet_input(boxes, masks, emb, null_pos, null_xyxy, num_freqs, out):
    for t, n in ((boxes, "boxes"), (masks, "masks"), (emb, "emb"), (null_pos, "null_pos"), (null_xyxy, "null_xyxy")):
        _req(t, F32, n, 4)
        if not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous")
    _req(out, F16, "out")
    rows = boxes.shape[0] * boxes.shape[1]
    in_dim = emb.shape[-1]
    check(_lib.lib().gl_posnet_input(boxes.data_ptr(), masks.data_ptr(), emb.data_ptr(), null_pos.data_ptr(),
                                     null_xyxy.data_ptr(), rows, in_dim, num_freqs, out.data_ptr(), _stream()),
          "gl_posnet_input")
    return out


def timestep_embedding(t: torch.Tensor, dim: int, out: torch.Tensor):
    _req(t, F32, "t", 4)
    _req(out, F16, "out")
    check(_lib.lib().gl_timestep_embedding(t.data_ptr(), t.shape[0], dim, out.data_ptr(), _stream()),
          "gl_timestep_embedding")
    return out


def silu(x: torch.Tensor, y: torch.Tensor):
    _req(x, F16, "x")
    _req(y, F16, "y")
    check(_lib.lib().gl_silu_f16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "gl_silu_f16")
    return y


def cfg_combine(eps2b: torch.Tensor, guidance: float, e_out: torch.Tensor):
    _req(eps2b, F32, "eps2b")
    _req(e_out, F32, "e_out")
    check(_lib.lib().gl_cfg_combine(eps2b.data_ptr(), guidance, e_out.numel(), e_out.data_ptr(), _stream()),
          "gl_cfg_combine")
    return e_out


def plms_update(x, e, olds, coefs, div, sqrt_at, s1m, sqrt_aprev, dir_coef, x_prev):
    """x_prev from e' = (coefs[0]*e + sum coefs[1+j]*olds[j]) / div (plms.py:144-161)."""
    _req(x, F32, "x")
    _req(e, F32, "e")
    _req(x_prev, F32, "x_prev")
    ps = [None, None, None]
    cs = [0.0, 0.0, 0.0]
    for j, o in enumerate(olds):
        _req(o, F32, "old_eps")
        ps[j] = o.data_ptr()
        cs[j] = float(coefs[1 + j])
    check(_lib.lib().gl_plms_update(x.data_ptr(), e.data_ptr(), ps[0], ps[1], ps[2], float(coefs[0]), cs[0], cs[1],
                                    cs[2], float(div), float(sqrt_at), float(s1m), float(sqrt_aprev), float(dir_coef),
                                    x.numel(), x_prev.data_ptr(), _stream()), "gl_plms_update")
    return x_prev


def pack_latent(x: torch.Tensor, Cpad: int, reps: int, out: torch.Tensor, split: bool = False):
    """x fp32 [B, C, h, w] -> out fp16 [reps*B, h*w, Cpad]; ``split``: channels [hi | lo | hi] of x (first-conv weights [Whi | Whi | Wlo])."""
    _req(x, F32, "x")
    _req(out, F16, "out")
    B, Cc, h, w = x.shape
    check(_lib.lib().gl_pack_latent(x.data_ptr(), B, Cc, h * w, Cpad, reps, int(split), out.data_ptr(), _stream()), "gl_pack_latent")
    return out


def softmax_rows(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """in-place softmax(scale * x) over the last dim of a 2-D fp16 tensor"""
    _req(x, F16, "x")
    rows, n, ld = _rows(x, "x")
    check(_lib.lib().gl_softmax_rows(x.data_ptr(), rows, n, ld, scale, _stream()), "gl_softmax_rows")
    return x


def latent_affine_pack(z: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, pre: float, Cpad: int, out: torch.Tensor):
    """z fp32 [B, C, h, w] -> out fp16 [B*h*w, Cpad] = post_quant_conv(z * pre), zero-padded channels"""
    for t, n in ((z, "z"), (w, "w"), (bias, "bias")):
        _req(t, F32, n, 4)
    _req(out, F16, "out")
    B, Cc, h, ww = z.shape
    check(_lib.lib().gl_latent_affine_pack(z.data_ptr(), w.data_ptr(), bias.data_ptr(), pre, B, Cc, h * ww, Cpad,
                                           out.data_ptr(), _stream()), "gl_latent_affine_pack")
    return out


def set_option(key: int, value: int) -> None:
    """Tuning knob for A/B measurements (see gl_set_option in include/gligen_hip.h)."""
    check(_lib.lib().gl_set_option(key, value), "gl_set_option")


def gemm8_launch_count() -> int:
    """gl_gemm / gl_conv3x3 calls this process has served with the 8-wave kernel (gl_debug_read(9)); the tests of that kernel
    assert it moved, so a dispatcher that quietly stopped choosing it cannot pass them."""
    import ctypes
    n = ctypes.c_uint64(0)
    check(_lib.lib().gl_debug_read(9, ctypes.byref(n), 8), "gl_debug_read")
    return int(n.value)


# ------------------------------------------------------------------------------------------- CLIP towers (reward stage)
def clip_patchify(pixel_values: torch.Tensor, patch: int, Kpad: int, out: torch.Tensor) -> torch.Tensor:
    """pixel_values fp32 [B, 3, S, S] -> fp16 [B * (S/patch)^2, Kpad] patch rows (K index (c, i, j), zero-padded)."""
    _req(pixel_values, F32, "pixel_values")
    _req(out, F16, "out")
    B, c3, S, S2 = pixel_values.shape
    if c3 != 3 or S != S2 or not pixel_values.is_contiguous():
        raise ValueError("pixel_values must be a contiguous [B, 3, S, S] tensor")
    check(_lib.lib().gl_clip_patchify(pixel_values.data_ptr(), B, S, patch, Kpad, out.data_ptr(), _stream()), "gl_clip_patchify")
    return out


def clip_assemble(patch_emb: torch.Tensor, class_emb: torch.Tensor, pos_emb: torch.Tensor, B: int, T: int, x: torch.Tensor,
                  ln_gamma: Optional[torch.Tensor] = None, ln_beta: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    _req(patch_emb, F16, "patch_emb")
    for t, n in ((class_emb, "class_emb"), (pos_emb, "pos_emb"), (x, "x")):
        _req(t, F32, n)
    C_ = x.shape[-1]
    check(_lib.lib().gl_clip_assemble(patch_emb.data_ptr(), _rows(patch_emb, "patch_emb")[2], class_emb.data_ptr(), pos_emb.data_ptr(), B, T, C_,
                                      _ptr(ln_gamma), _ptr(ln_beta), eps, x.data_ptr(), _stream()), "gl_clip_assemble")
    return x


def clip_embed_tokens(ids: torch.Tensor, tok_emb: torch.Tensor, pos_emb: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    _req(ids, torch.int32, "ids", 4)
    for t, n in ((tok_emb, "tok_emb"), (pos_emb, "pos_emb"), (x, "x")):
        _req(t, F32, n)
    B, T = ids.shape
    check(_lib.lib().gl_clip_embed_tokens(ids.data_ptr(), tok_emb.data_ptr(), pos_emb.data_ptr(), B, T, x.shape[-1], tok_emb.shape[0],
                                          x.data_ptr(), _stream()), "gl_clip_embed_tokens")
    return x


def clip_gather_rows(x: torch.Tensor, rows: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _req(x, F32, "x")
    _req(rows, torch.int32, "rows", 4)
    _req(out, F32, "out")
    check(_lib.lib().gl_clip_gather_rows(x.data_ptr(), _rows(x, "x")[2], rows.data_ptr(), rows.numel(), x.shape[-1], out.data_ptr(), _stream()),
          "gl_clip_gather_rows")
    return out


def attention_small(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, ld: int, B: int, T: int, H: int, d: int, scale: float,
                    causal: bool, out: torch.Tensor) -> torch.Tensor:
    """q / k / v: fp16 views whose element (b, t, h*d + c) sits at data_ptr + (b*T + t)*ld + h*d + c; out fp16 [B*T, H*d]."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _req(t, F16, n, 2)
    _req(out, F16, "out", 2)
    check(_lib.lib().gl_attention_small(q.data_ptr(), k.data_ptr(), v.data_ptr(), ld, B, T, H, d, scale, int(causal), out.data_ptr(),
                                        _rows(out, "out")[2], _stream()), "gl_attention_small")
    return out


# ------------------------------------------------------------------------------------------- reward-stage image preprocessing
def image_to_u8(img: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """decoded image fp32 [B, 3, H, W] -> uint8 [B, H, W, 3] (interface.py:543-547 arithmetic)."""
    _req(img, F32, "img")
    B, c3, H, W = img.shape
    if c3 != 3 or not img.is_contiguous() or out.dtype != torch.uint8 or tuple(out.shape) != (B, H, W, 3) or not out.is_contiguous():
        raise ValueError("img must be contiguous fp32 [B, 3, H, W] and out contiguous uint8 [B, H, W, 3]")
    check(_lib.lib().gl_image_to_u8(img.data_ptr(), B, H, W, out.data_ptr(), _stream()), "gl_image_to_u8")
    return out


def _req_tables(bounds: torch.Tensor, coeffs: torch.Tensor, n_out: int, ksize: int) -> None:
    if bounds.dtype != torch.int32 or coeffs.dtype != torch.int32 or not bounds.is_cuda or not coeffs.is_cuda:
        raise ValueError("bounds / coeffs must be int32 tensors on the GPU")
    if tuple(bounds.shape) != (n_out, 2) or tuple(coeffs.shape) != (n_out, ksize) or not bounds.is_contiguous() or not coeffs.is_contiguous():
        raise ValueError("bounds must be [n_out, 2] and coeffs [n_out, ksize], contiguous")


def resample_h_u8(x: torch.Tensor, bounds: torch.Tensor, coeffs: torch.Tensor, ksize: int, out: torch.Tensor) -> torch.Tensor:
    """Pillow's horizontal 8-bit resampling pass: uint8 [B, H, W, 3] -> uint8 [B, H, Wout, 3]."""
    if x.dtype != torch.uint8 or out.dtype != torch.uint8 or not x.is_cuda or not out.is_cuda or not x.is_contiguous() or not out.is_contiguous():
        raise ValueError("x / out must be contiguous uint8 tensors on the GPU")
    B, H, W, _ = x.shape
    Wout = out.shape[2]
    if tuple(out.shape) != (B, H, Wout, 3):
        raise ValueError("out must be [B, H, Wout, 3]")
    _req_tables(bounds, coeffs, Wout, ksize)
    check(_lib.lib().gl_resample_h_u8(x.data_ptr(), B, H, W, bounds.data_ptr(), coeffs.data_ptr(), ksize, Wout, out.data_ptr(), _stream()),
          "gl_resample_h_u8")
    return out


def resample_v_norm(x: torch.Tensor, bounds: torch.Tensor, coeffs: torch.Tensor, ksize: int, Hout: int, top: int, left: int, crop_h: int,
                    crop_w: int, mean, std, out: Optional[torch.Tensor], out_u8: Optional[torch.Tensor] = None):
    """Pillow's vertical pass inside the centre-crop window + /255, (x - mean) / std, channels first."""
    import ctypes

    import numpy as np
    if x.dtype != torch.uint8 or not x.is_cuda or not x.is_contiguous():
        raise ValueError("x must be a contiguous uint8 tensor on the GPU")
    B, H, W, _ = x.shape
    _req_tables(bounds, coeffs, Hout, ksize)
    if out is not None and (out.dtype != F32 or tuple(out.shape) != (B, 3, crop_h, crop_w) or not out.is_contiguous()):
        raise ValueError("out must be contiguous fp32 [B, 3, crop_h, crop_w]")
    if out_u8 is not None and (out_u8.dtype != torch.uint8 or tuple(out_u8.shape) != (B, crop_h, crop_w, 3) or not out_u8.is_contiguous()):
        raise ValueError("out_u8 must be contiguous uint8 [B, crop_h, crop_w, 3]")
    m = np.ascontiguousarray(mean, dtype=np.float32)
    s_ = np.ascontiguousarray(std, dtype=np.float32)
    check(_lib.lib().gl_resample_v_norm(x.data_ptr(), B, H, W, bounds.data_ptr(), coeffs.data_ptr(), ksize, Hout, top, left, crop_h, crop_w,
                                        m.ctypes.data_as(ctypes.c_void_p), s_.ctypes.data_as(ctypes.c_void_p),
                                        out.data_ptr() if out is not None else None, out_u8.data_ptr() if out_u8 is not None else None,
                                        _stream()), "gl_resample_v_norm")
    return out
