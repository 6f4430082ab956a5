"""TEST INFRASTRUCTURE: the UNet forward as an op-by-op launch sequence written in Python over the op-level C ABI
(layoutllm_t2i_amd.ops).  The product's launch sequence lives in C++ behind gl_set_conditioning / gl_unet_forward
(csrc/engine.hip); this independent restatement drives the SAME kernels through the SAME op-level entry points, so the
two must agree bitwise (tests/test_gpu_engine.py) -- a check of the C++ orchestration (buffer pool, plan builder,
pointer / stride arithmetic, device-side box rectangles), not of the kernels.


Replaces ``UNetModel.forward`` (openaimodel.py:413-459) and everything below it.  Differences from
the reference that are *result-identical*  and where the time goes:

  * token-major fp16 activations end to end; concat / residual / time-embedding / GEGLU / gates are
    fused into GEMM and conv epilogues; attention never materialises the N x N score tensor;
  * everything that depends only on the conditioning is hoisted into ``set_conditioning`` (once per
    image instead of 102 times): PositionNet tokens, fuser.linear(objs) per layer, attn2 and
    rela_fuse K/V projections, integer box rectangles per resolution;
  * the gated-SA fuser is skipped outright when the sampler sets scale == 0 (exact identity);
  * the RelationCrossAttention injection runs in closed form (two small kernels + tiny GEMMs);
  * a whole forward is captured into a HIP graph per (fuser on/off, first-conv variant) and replayed;
  * the RESIDUAL STREAM is fp32: every block output (ResBlock sum, transformer-block x after each of its residual
    adds, proj_out + x_in, conv_in / down / up outputs) is accumulated and stored in fp32 by the producing GEMM / conv
    epilogue, with an fp16 copy only where a matrix-core consumer needs one (GroupNorm -> conv, 1x1 skip conv, down /
    up convs, proj_out).  LayerNorm reads the fp32 stream; rela_fuse's LN3 term is re-evaluated in fp32 inside
    rela_merge.  Only operands of matrix products are fp16 (tools/precision_sim.py: that floor is rel-L2 1.1e-3).

No torch op touches activations on the hot path: torch provides device memory, streams and graphs.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from layoutllm_t2i_amd import host, ops
from layoutllm_t2i_amd._lib import EPI_BIAS, EPI_GATE_RES, EPI_GEGLU, EPI_RES, EPI_ROWBIAS, EPI_SILU, init_device
from layoutllm_t2i_amd.arch import Block, Layer, UNetConfig
from layoutllm_t2i_amd.weights import CIN_PAD, PackedWeights

F16, F32 = torch.float16, torch.float32


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class PyRefEngine:
    def __init__(self, packed: PackedWeights):
        if not torch.cuda.is_available():
            raise RuntimeError("needs a GPU")
        init_device()
        self.P = packed
        self.cfg: UNetConfig = packed.cfg
        self.plan = packed.plan
        self.dev = packed.device
        self.W, self.S = packed.w, packed.s
        self._pool: Dict[Tuple, torch.Tensor] = {}
        self._gates: Dict[str, torch.Tensor] = {}
        self.st_layers: List[Layer] = self.plan.st_layers()
        # the engine's precision mode (gl_set_option keys 41 / 42, both on by default): split-fp16 1x1 convs + GroupNorm on the fp32
        # stream, and the ResBlock's first conv writing fp32
        self.precise, self.h1_f32 = True, True
        self.share_prefix = True              # gl_set_option 44
        self.rela_compact = True              # gl_set_option 43
        self.w3 = 1024                        # gl_set_option 45: rows threshold of the third pass (0 = off)
        self.in_split = True                  # gl_set_option 38
        # constant gates of rela_fuse, per-step gates of the fuser (scale * tanh(alpha))
        for l in self.st_layers:
            t = l.prefix + ".transformer_blocks.0"
            for k in ("tanh_attn", "tanh_dense"):
                self._gates[f"{t}.rela_fuse.{k}"] = torch.full((1,), self.S[f"{t}.rela_fuse.{k}"], dtype=F32, device=self.dev)
                self._gates[f"{t}.fuser.{k}"] = torch.zeros((1,), dtype=F32, device=self.dev)
        self._fuser_scale: Optional[float] = None
        self.cond: Optional[dict] = None
        self._graphs: Dict[Tuple, torch.cuda.CUDAGraph] = {}
        self.use_graphs = True

    # ------------------------------------------------------------------ memory
    def buf(self, tag: str, shape, dtype=F16, zero: bool = False) -> torch.Tensor:
        key = (tag, tuple(shape), dtype)
        t = self._pool.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.dev)
            self._pool[key] = t
        return t

    def pool_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._pool.values())

    # ------------------------------------------------------------------ conditioning (once per image)
    @torch.no_grad()
    def _w3(self, rows: int) -> int:
        """engine.hip Run::gemm: the third pass xhi.Wlo for launches of more than 1024 rows (key 45)"""
        return 2 if (self.w3 and rows > max(1024, int(self.w3))) else 1

    def set_conditioning(self, context, relations, boxes, masks, positive_embeddings, hw: int) -> None:
        """context [Bn,77,ctx], relations [Bn,R,ctx], boxes [Bn,30,4], masks [Bn,30],
        positive_embeddings [Bn,30,in_dim]: fp32 tensors (any device).  Null grounding = zeros
        (text_layout_tokinzer_input.py:47-62).  ``hw`` = latent side (64 for 512x512)."""
        dev, cfg, W = self.dev, self.cfg, self.W
        f32 = lambda t: torch.as_tensor(t, dtype=F32).to(dev).contiguous()
        context, relations = f32(context), f32(relations)
        boxes, masks, pe = f32(boxes), f32(masks), f32(positive_embeddings)
        Bn, mo = boxes.shape[0], boxes.shape[1]
        R = relations.shape[1]
        Lc = context.shape[1]
        c: dict = dict(Bn=Bn, mo=mo, R=R, Lc=Lc, hw=hw)
        # --- grounding tokens: PositionNet (text_grounding_net.py:26-43)
        pin = self.buf("pn.in", (Bn * mo, cfg.pos_in_dim + cfg.position_dim))
        ops.posnet_input(boxes, masks, pe, W["position_net.null_pos"], W["position_net.null_xyxy"], cfg.fourier_freqs, pin)
        h1 = ops.gemm(pin, W["position_net.linears.0.w"], self.buf("pn.h1", (Bn * mo, 512)), W["position_net.linears.0.b"], EPI_SILU)
        h2 = ops.gemm(h1, W["position_net.linears.2.w"], self.buf("pn.h2", (Bn * mo, 512)), W["position_net.linears.2.b"], EPI_SILU)
        objs = ops.gemm(h2, W["position_net.linears.4.w"], self.buf("pn.objs", (Bn * mo, cfg.pos_out_dim)), W["position_net.linears.4.b"])
        c["objs"] = objs
        ctx16 = self.buf("cond.ctx", (Bn * Lc, cfg.context_dim))
        ctx16.copy_(context.reshape(Bn * Lc, -1))
        rel16 = self.buf("cond.rel", (Bn * R, cfg.context_dim))
        rel16.copy_(relations.reshape(Bn * R, -1))
        H = cfg.num_heads
        for li, l in enumerate(self.st_layers):
            t = l.prefix + ".transformer_blocks.0"
            C, d = l.cin, l.d_head
            # fuser.linear(objs) (attention.py:228)
            c[f"objs.{li}"] = ops.gemm(objs, W[f"{t}.fuser.linear.w"], self.buf(f"hoist.objs.{li}", (Bn * mo, C)), W[f"{t}.fuser.linear.b"])
            c[f"objs32.{li}"] = ops.gemm(objs, W[f"{t}.fuser.linear.w"], self.buf(f"hoist.objs32.{li}", (Bn * mo, C), F32), W[f"{t}.fuser.linear.b"])
            # attn2 K/V of the text context (attention.py:124-125)
            kv = ops.gemm(ctx16, W[f"{t}.attn2.kv.w"], self.buf(f"hoist.kvctx.{li}", (Bn * Lc, 2 * C)))
            vt = self.buf(f"hoist.vtctx.{li}", (Bn, H, d, ops.vt_ld(Lc)))
            ops.transpose_v(kv[:, C:], Lc * 2 * C, 2 * C, vt, Bn, H, d, Lc)
            c[f"kvctx.{li}"], c[f"vtctx.{li}"] = kv, vt
            # rela_fuse K/V of the relation tokens (attention.py:348-349)
            kvr = ops.gemm(rel16, W[f"{t}.rela_fuse.attn.kv.w"], self.buf(f"hoist.kvrel.{li}", (Bn * R, 2 * C)))
            vtr = self.buf(f"hoist.vtrel.{li}", (Bn, H, d, ops.vt_ld(R)))
            ops.transpose_v(kvr[:, C:], R * 2 * C, 2 * C, vtr, Bn, H, d, R)
            c[f"kvrel.{li}"], c[f"vtrel.{li}"] = kvr, vtr
        # --- integer rectangles per resolution (host, fp32; attention.py:321-346)
        bx, mk = boxes.cpu().numpy(), masks.cpu().numpy()
        seen = set()
        nv_max = 0
        for level_hw in self._st_resolutions(hw):
            if level_hw in seen:
                continue
            seen.add(level_hw)
            rects, nvalid, poison = host.box_rects(bx, mk, level_hw, level_hw)
            nv_max = max(nv_max, int(nvalid.max()) if nvalid.size else 0)
            # pooled (address-stable) buffers, so captured graphs stay valid across images
            for nm, arr in (("rects", rects), ("nvalid", nvalid), ("poison", poison)):
                t = self.buf(f"cond.{nm}.{level_hw}", arr.shape, torch.int32)
                t.copy_(torch.from_numpy(arr))
                c[f"{nm}.{level_hw}"] = t
        # rows per sample of the relation chain (engine.hip: rel_slots): the largest nvalid, rounded up to 8
        c["ms"] = min(mo, max(8, (nv_max + 7) & ~7)) if self.rela_compact else mo
        self.cond = c
        torch.cuda.current_stream().synchronize()

    def _st_resolutions(self, hw: int) -> List[int]:
        out, cur = [], hw
        for b in self.plan.input_blocks:
            for l in b.layers:
                if l.kind == "down":
                    cur //= 2
                elif l.kind == "st":
                    out.append(cur)
        out.append(cur)  # middle
        return sorted(set(out), reverse=True)

    # ------------------------------------------------------------------ per-step scalars
    def set_fuser_scale(self, scale: float) -> None:
        """set_alpha_scale (interface.py:34-38): only GatedSelfAttentionDense is gated."""
        if self._fuser_scale == scale:
            return
        for l in self.st_layers:
            f = l.prefix + ".transformer_blocks.0.fuser"
            # fp32 product, like the reference's `scale * torch.tanh(alpha)` (python scalar x fp32 tensor) and like engine.hip
            self._gates[f + ".tanh_attn"].fill_(float(np.float32(scale) * np.float32(self.S[f + ".tanh_attn"])))
            self._gates[f + ".tanh_dense"].fill_(float(np.float32(scale) * np.float32(self.S[f + ".tanh_dense"])))
        self._fuser_scale = scale

    # ------------------------------------------------------------------ layers
    def _groupnorm(self, x1, x2, Bn, HW, p, eps, silu, tag, hilo=False, raw_tag=None):
        """x1 / x2 fp16 or fp32 (stream).  hilo: rows come out as [hi | lo] (2C wide); raw_tag: also the INPUT concat as [hi | lo]."""
        C = x1.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
        out = self.buf(tag, (Bn * HW, 2 * C if hilo else C))
        partial = self.buf("gn.partial", (Bn * 64 * 64,), F32)
        raw = self.buf(raw_tag, (Bn * HW, 2 * C)) if raw_tag else None
        ops.groupnorm(x1, x2, Bn, HW, self.W[p + ".g"], self.W[p + ".b"], eps, silu, out[:, :C], partial,
                      out_lo=out[:, C:] if hilo else None, raw=raw)
        return (out, raw) if raw_tag else out

    def _stream(self, tag, M, C, need_h=True):
        """fp32 residual-stream tensor + (where a down / up conv consumes it, or in the fp16-copy mode) its fp16 copy"""
        return self.buf(tag + ".f32", (M, C), F32), (self.buf(tag, (M, C)) if need_h else None)

    def _res_block(self, l: Layer, h, skip, Bn, side, emb_out, out_tag, need_h=True, out=None):
        """h = (fp32, fp16) stream pair; skip = (fp32, fp16) pair popped from the skip stack or None; out = (fp32, fp16 or None)
        row views to write into (the half-batch shared prefix writes the first half of full-size tensors)."""
        W, p = self.W, l.prefix
        HW = side * side
        h32, h16 = h
        split = None
        if self.precise:
            s32 = skip[0] if skip is not None else None
            if l.cin != l.cout:
                t, split = self._groupnorm(h32, s32, Bn, HW, p + ".in_layers.0", 1e-5, True, "rb.gn1", raw_tag="rb.split")
            else:
                t = self._groupnorm(h32, s32, Bn, HW, p + ".in_layers.0", 1e-5, True, "rb.gn1")
        else:
            s16 = skip[1] if skip is not None else None
            t = self._groupnorm(h16, s16, Bn, HW, p + ".in_layers.0", 1e-5, True, "rb.gn1")
        off = self.P.emb_offsets[p]
        h1f = self.precise and self.h1_f32
        h1 = ops.conv3x3(t, W[p + ".in_layers.2.w"], self.buf("rb.h1f" if h1f else "rb.h1", (Bn * HW, l.cout), F32 if h1f else F16), Bn, side, side,
                         W[p + ".in_layers.2.b"], epi=EPI_ROWBIAS, rowbias=emb_out[:, off:off + l.cout], rows_per_sample=HW)
        t2 = self._groupnorm(h1, None, Bn, HW, p + ".out_layers.0", 1e-5, True, "rb.gn2")
        if l.cin != l.cout:
            skb = self.buf("rb.skip.f32", (Bn * HW, l.cout), F32)
            if self.precise:
                sk = ops.gemm(split, W[p + ".skip_connection.w"], skb, W[p + ".skip_connection.b"], hilo_a=True, wsplit=self._w3(Bn * HW))
            else:
                sk = ops.gemm(h16, W[p + ".skip_connection.w"], skb, W[p + ".skip_connection.b"], a2=s16, wsplit=1)
        else:
            assert skip is None
            sk = h32
        o32, o16 = out if out is not None else self._stream(out_tag, Bn * HW, l.cout, need_h)
        ops.conv3x3(t2, W[p + ".out_layers.3.w"], o32, Bn, side, side, W[p + ".out_layers.3.b"], epi=EPI_RES, res=sk, out16=o16)
        return o32, o16

    def _self_attention(self, src, rows_per_b, Nq, Nk, C, d, wp, tagp, Bn=None):
        """src [Bn*rows_per_b, C] (already normalised) -> attention output [Bn*Nq, C] (before to_out)."""
        Bn, H = (self.cond["Bn"] if Bn is None else Bn), self.cfg.num_heads
        vt = self.buf(tagp + ".vt", (Bn, H, d, ops.vt_ld(max(Nk, rows_per_b))))
        if C % 32 == 0:      # same rule as engine.hip: V^T straight from the QKV GEMM epilogue (that form never splits K)
            qkv = ops.gemm(src, self.W[wp + ".qkv.w"], self.buf(tagp + ".qkv", (Bn * rows_per_b, 3 * C)), vt=vt, vt_col0=2 * C, vt_rows=rows_per_b)
        else:
            qkv = ops.gemm(src, self.W[wp + ".qkv.w"], self.buf(tagp + ".qkv", (Bn * rows_per_b, 3 * C)))
            ops.transpose_v(qkv[:, 2 * C:], rows_per_b * 3 * C, 3 * C, vt, Bn, H, d, Nk)
        att = self.buf(tagp + ".att", (Bn * Nq, C))
        ops.attention(qkv, rows_per_b * 3 * C, 3 * C, qkv[:, C:], rows_per_b * 3 * C, 3 * C, vt, att, Nq * C, C,
                      Bn, H, d, Nq, Nk, d ** -0.5, q_prescaled=True)
        return att

    def _feed_forward(self, xn, res, p, M, C, out, gate=None, hilo_out=False):
        W = self.W
        if ops.ff_fused_applicable(C, M):
            return ops.ff_fused(xn, W[p + ".ff1.w"], W[p + ".ff1.b"], W[p + ".ff2.w"], W[p + ".ff2.b"], res, out, gate=gate, hilo_out=hilo_out)
        hg = ops.gemm(xn, W[p + ".ff1.w"], self.buf("ff.h", (M, 4 * C)), W[p + ".ff1.b"], EPI_GEGLU)
        if gate is None:
            return ops.gemm(hg, W[p + ".ff2.w"], out, W[p + ".ff2.b"], EPI_RES, res=res, hilo_out=hilo_out)
        return ops.gemm(hg, W[p + ".ff2.w"], out, W[p + ".ff2.b"], EPI_GATE_RES, res=res, gate=gate)

    def _spatial_transformer(self, l: Layer, li: int, x_in, Bn, side, fuser_on, out_tag, need_h=True, share_half=False):
        """x_in = (fp32, fp16) stream pair.  Inside the block x lives in fp32 only (two ping-pong buffers).  share_half: x_in holds only
        the first Bn / 2 samples (shared cond / uncond prefix): GroupNorm .. attn1 run on those, then x_in and x are duplicated."""
        W, c, cfg = self.W, self.cond, self.cfg
        p = l.prefix
        t = p + ".transformer_blocks.0"
        C, d, H = l.cin, l.d_head, cfg.num_heads
        N = side * side
        M = Bn * N
        mo, R, Lc = c["mo"], c["R"], c["Lc"]
        xin32, xin16 = x_in
        xa, xb = self.buf("st.xa", (M, C), F32), self.buf("st.xb", (M, C), F32)
        B1 = Bn // 2 if share_half else Bn
        M1 = B1 * N
        xa1, xb1 = xa[:M1], xb[:M1]
        if self.precise:     # Normalize on the fp32 stream, [hi | lo] rows, both halves against proj_in's weight
            g0 = self._groupnorm(xin32[:M1], None, B1, N, p + ".norm", 1e-6, False, "st.gn", hilo=True)
            x = ops.gemm(g0, W[p + ".proj_in.w"], xa1, W[p + ".proj_in.b"], hilo_a=True, wsplit=self._w3(g0.shape[0]))
        else:
            g0 = self._groupnorm(xin16[:M1], None, B1, N, p + ".norm", 1e-6, False, "st.gn")
            x = ops.gemm(g0, W[p + ".proj_in.w"], xa1, W[p + ".proj_in.b"], wsplit=1)
        # --- attn1 (attention.py:395)
        n1 = ops.layernorm(x, self.buf("st.ln", (M1, C)), W[t + ".norm1.g"], W[t + ".norm1.b"], B1, N)
        att = self._self_attention(n1, N, N, N, C, d, t + ".attn1", "st.sa", Bn=B1)
        x = ops.gemm(att, W[t + ".attn1.o.w"], xb1, W[t + ".attn1.o.b"], EPI_RES, res=x)
        if share_half:
            xb[M1:].copy_(xb[:M1])
            xin32[M1:].copy_(xin32[:M1])
            if xin16 is not None:
                xin16[M1:].copy_(xin16[:M1])
        x = xb
        nxt = lambda cur: xb if cur is xa else xa
        # --- gated self-attention fuser over [x ; objs] (attention.py:226-234); exact identity at scale 0
        if fuser_on:
            f = t + ".fuser"
            rows = N + ((mo + 7) & ~7)          # same padding as engine.hip (pad rows are masked keys / unused queries)
            cat = self.buf("st.cat", (Bn * rows, C), zero=True)
            ops.layernorm(x, cat, W[f + ".norm1.g"], W[f + ".norm1.b"], Bn, N, rows, 0, x2=c[f"objs32.{li}" if self.precise else f"objs.{li}"], rows2=mo)
            att = self._self_attention(cat, rows, N, N + mo, C, d, f + ".attn", "st.fa")
            x = ops.gemm(att, W[f + ".attn.o.w"], nxt(x), W[f + ".attn.o.b"], EPI_GATE_RES, res=x,
                         gate=self._gates[f + ".tanh_attn"])
            n2 = ops.layernorm(x, self.buf("st.ln", (M, C)), W[f + ".norm2.g"], W[f + ".norm2.b"], Bn, N)
            x = self._feed_forward(n2, x, f + ".ff", M, C, nxt(x), gate=self._gates[f + ".tanh_dense"])
        # --- relation injection (attention.py:315-359, :398), closed form
        r = t + ".rela_fuse"
        rects, nvalid, poison = c[f"rects.{side}"], c[f"nvalid.{side}"], c[f"poison.{side}"]
        stats = self.buf("st.lnstats", (M, 2), F32)
        ms = c["ms"]
        Mo = Bn * ms
        fn = self.buf("rl.ln", (Mo, C))
        if self.precise:     # LayerNorm3 never materialised: statistics only, box means of LN3(x) in fp32 from the stream
            ops.layernorm_stats(x, stats)
            feat = ops.rela_pool_ln3(x, stats, W[r + ".norm3.g"], W[r + ".norm3.b"], Bn, side, side, C, rects, nvalid, poison, mo,
                                     self.buf("rl.feat", (Mo, C)), ln_gamma=W[r + ".norm1.g"], ln_beta=W[r + ".norm1.b"], ln_out=fn, slots=ms)
        else:
            hid = ops.layernorm(x, self.buf("st.hid", (M, C)), W[r + ".norm3.g"], W[r + ".norm3.b"], Bn, N, stats=stats)
            feat = ops.rela_pool(hid, Bn, side, side, C, rects, nvalid, poison, mo, self.buf("rl.feat", (Mo, C)),
                                 ln_gamma=W[r + ".norm1.g"], ln_beta=W[r + ".norm1.b"], ln_out=fn, slots=ms)
        q = ops.gemm(fn, W[r + ".attn.q.w"], self.buf("rl.q", (Mo, C)))
        kv = c[f"kvrel.{li}"]
        ar = self.buf("rl.att", (Mo, C))
        ops.attention(q, ms * C, C, kv, R * 2 * C, 2 * C, c[f"vtrel.{li}"], ar, ms * C, C, Bn, H, d, ms, R, d ** -0.5, q_prescaled=True)
        f1 = ops.gemm(ar, W[r + ".attn.o.w"], self.buf("rl.f1", (Mo, C)), W[r + ".attn.o.b"], EPI_GATE_RES, res=feat,
                      gate=self._gates[r + ".tanh_attn"])
        fn2 = ops.layernorm(f1, self.buf("rl.ln", (Mo, C)), W[r + ".norm2.g"], W[r + ".norm2.b"], Bn, ms)
        hg = ops.gemm(fn2, W[r + ".ff.ff1.w"], self.buf("rl.ffh", (Mo, 4 * C)), W[r + ".ff.ff1.b"], EPI_GEGLU)
        f2 = ops.gemm(hg, W[r + ".ff.ff2.w"], self.buf("rl.f2", (Mo, C)), W[r + ".ff.ff2.b"], EPI_GATE_RES, res=f1,
                      gate=self._gates[r + ".tanh_dense"])
        x = ops.rela_merge(x, None, f2, Bn, side, side, C, rects, nvalid, poison, mo, nxt(x), ln_stats=stats,
                           gamma=W[r + ".norm3.g"], beta=W[r + ".norm3.b"], slots=ms)
        # --- attn2: text cross-attention with hoisted K/V (attention.py:400)
        n = ops.layernorm(x, self.buf("st.ln", (M, C)), W[t + ".norm2.g"], W[t + ".norm2.b"], Bn, N)
        q2 = ops.gemm(n, W[t + ".attn2.q.w"], self.buf("st.q2", (M, C)))
        a2 = self.buf("st.att2", (M, C))
        ops.attention(q2, N * C, C, c[f"kvctx.{li}"], Lc * 2 * C, 2 * C, c[f"vtctx.{li}"], a2, N * C, C, Bn, H, d, N, Lc,
                      d ** -0.5, q_prescaled=True)
        x = ops.gemm(a2, W[t + ".attn2.o.w"], nxt(x), W[t + ".attn2.o.b"], EPI_RES, res=x)
        # --- GEGLU feed-forward (attention.py:401): the sum is only consumed by proj_out's matrix product -> fp16
        n3 = ops.layernorm(x, self.buf("st.ln", (M, C)), W[t + ".norm3.g"], W[t + ".norm3.b"], Bn, N)
        x16 = self._feed_forward(n3, x, t + ".ff", M, C, self.buf("st.x6", (M, 2 * C if self.precise else C)), hilo_out=self.precise)
        # --- proj_out + residual (attention.py:444-446)
        o32, o16 = self._stream(out_tag, M, C, need_h)
        ops.gemm(x16, W[p + ".proj_out.w"], o32, W[p + ".proj_out.b"], EPI_RES, res=xin32, out16=o16, hilo_a=self.precise,
                 wsplit=self._w3(x16.shape[0]) if self.precise else 1)
        return o32, o16

    # ------------------------------------------------------------------ one forward (eager launch sequence)
    def _launch_forward(self, x_lat: torch.Tensor, t_buf: torch.Tensor, reps: int, fuser_on: bool, sd_conv: bool,
                        eps_out: torch.Tensor, uniform_t: bool = False) -> None:
        W, cfg, c = self.W, self.cfg, self.cond
        Bn, side = c["Bn"], c["hw"]
        ib0 = self.plan.input_blocks
        # same rule as engine.hip (gl_set_option 44): the [cond ; uncond] halves share everything before the first conditioning-dependent op
        share = self.share_prefix and reps == 2 and uniform_t and Bn % 2 == 0 and len(ib0) > 1 and bool(ib0[1].layers) and ib0[1].layers[0].kind == "res"
        B0 = Bn // 2 if share else Bn
        mc = cfg.model_channels
        st_index = {l.prefix: i for i, l in enumerate(self.st_layers)}
        # time embedding (openaimodel.py:428-429) and all 22 emb_layers in one GEMM (:172-178, :220)
        te = ops.timestep_embedding(t_buf, mc, self.buf("te.sin", (Bn, mc)))
        e1 = ops.gemm(te, W["time_embed.0.w"], self.buf("te.e1", (Bn, 4 * mc)), W["time_embed.0.b"], EPI_SILU)
        e2 = ops.gemm(e1, W["time_embed.2.w"], self.buf("te.e2", (Bn, 4 * mc)), W["time_embed.2.b"], EPI_SILU)
        emb_out = ops.gemm(e2, W["emb_all.w"], self.buf("te.out32" if self.precise else "te.out", (Bn, self.P.emb_total), F32 if self.precise else F16),
                           W["emb_all.b"])
        # first conv on the zero-padded NHWC latent (openaimodel.py:299, :393-405)
        xin = ops.pack_latent(x_lat, CIN_PAD, 1 if share else reps, self.buf("in.x", (B0 * side * side, CIN_PAD)),
                              split=self.in_split and 3 * self.cfg.in_channels <= CIN_PAD)
        fc = "sd_first_conv" if sd_conv else "input_blocks.0.0"
        M0 = Bn * side * side
        Mh = B0 * side * side
        # fp16 copies of stream tensors only where a down / up conv reads them (same rule as engine.hip)
        first_kind = lambda blk: blk.layers[0].kind if blk is not None and blk.layers else None
        wants_h = lambda kind: (not self.precise) or kind in ("down", "up")
        ib = self.plan.input_blocks
        h = self._stream("skip.0", M0, mc, wants_h(first_kind(ib[1] if len(ib) > 1 else None)))
        ops.conv3x3(xin, W[fc + ".w"], h[0][:Mh], B0, side, side, W[fc + ".b"], out16=None if h[1] is None else h[1][:Mh])

        def dup(pair, rows):
            for t_ in pair:
                if t_ is not None:
                    t_[rows:].copy_(t_[:rows])
        if share:
            dup(h, Mh)
        skips: List[Tuple[Tuple[torch.Tensor, torch.Tensor], int]] = [(h, side)]

        def run_block(b: Block, h, side, bi: str, skip=None, nxt_block=None, half_first=False):
            half_pending = False
            for j, l in enumerate(b.layers):
                tag = f"{bi}.{j}"
                need_h = wants_h(b.layers[j + 1].kind if j + 1 < len(b.layers) else first_kind(nxt_block))
                if l.kind == "res":
                    if half_first and j == 0:
                        rows = B0 * side * side
                        full = self._stream(tag, Bn * side * side, l.cout, need_h)
                        hv = (h[0][:rows], None if h[1] is None else h[1][:rows])
                        self._res_block(l, hv, skip, B0, side, emb_out, tag, need_h, out=(full[0][:rows], None if full[1] is None else full[1][:rows]))
                        h = full
                        half_pending = True
                    else:
                        h = self._res_block(l, h, skip, Bn, side, emb_out, tag, need_h)
                    skip = None
                elif l.kind == "st":
                    h = self._spatial_transformer(l, st_index[l.prefix], h, Bn, side, fuser_on, tag, need_h, share_half=half_pending)
                    half_pending = False
                elif l.kind == "down":
                    o = self._stream(tag, Bn * (side // 2) ** 2, l.cout, need_h)
                    ops.conv3x3(h[1], W[l.prefix + ".w"], o[0], Bn, side, side, W[l.prefix + ".b"], stride=2, out16=o[1])
                    h = o
                    side //= 2
                elif l.kind == "up":
                    o = self._stream(tag, Bn * (side * 2) ** 2, l.cout, need_h)
                    ops.conv3x3(h[1], W[l.prefix + ".w"], o[0], Bn, side, side, W[l.prefix + ".b"], upsample2x=True, out16=o[1])
                    h = o
                    side *= 2
                if half_pending and not (j + 1 < len(b.layers) and b.layers[j + 1].kind == "st"):
                    dup(h, B0 * side * side)
                    half_pending = False
            return h, side

        ob = self.plan.output_blocks
        for i, b in enumerate(ib[1:], start=1):
            h, side = run_block(b, h, side, f"skip.{i}", nxt_block=ib[i + 1] if i + 1 < len(ib) else self.plan.middle, half_first=share and i == 1)
            skips.append((h, side))
        h, side = run_block(self.plan.middle, h, side, "mid", nxt_block=ob[0] if ob else None)
        for i, b in enumerate(ob):
            sk, sside = skips.pop()
            assert sside == side
            h, side = run_block(b, h, side, f"out.{i}", skip=sk, nxt_block=ob[i + 1] if i + 1 < len(ob) else None)
        g = self._groupnorm(h[0] if self.precise else h[1], None, Bn, side * side, "out.0", 1e-5, True, "fin.gn")
        ops.conv3x3(g, W["out.2.w"], eps_out, Bn, side, side, W["out.2.b"], nchw_hw=side * side)

    # ------------------------------------------------------------------ public forward
    @torch.no_grad()
    def forward(self, x_lat: torch.Tensor, t, fuser_scale: float = 1.0, sd_conv: bool = False, reps: int = 1,
                eps_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_lat fp32 [Bn/reps, 4, hw, hw] (NCHW, like the reference); returns eps fp32 [Bn, 4, hw, hw].
        With reps=2 the latent is shared by both CFG halves of a [cond ; uncond] conditioning batch."""
        c = self.cond
        if c is None:
            raise RuntimeError("call set_conditioning() first")
        Bn, side = c["Bn"], c["hw"]
        if x_lat.shape[0] * reps != Bn or x_lat.shape[-1] != side:
            raise ValueError(f"latent batch {tuple(x_lat.shape)} x reps {reps} does not match conditioning batch {Bn} @ {side}")
        if sd_conv and "sd_first_conv.w" not in self.W:
            raise RuntimeError("SD first-conv weights were not packed")
        t_buf = self.buf("in.t", (Bn,), F32)
        if torch.is_tensor(t):
            t_buf.copy_(t.to(F32).reshape(-1).expand(Bn) if t.numel() == 1 else t.to(F32))
        else:
            t_buf.fill_(float(t))
        self.set_fuser_scale(float(fuser_scale))
        fuser_on = fuser_scale != 0
        if eps_out is None:
            eps_out = self.buf("out.eps", (Bn, self.cfg.out_channels, side, side), F32)
        x_static = self.buf("in.xlat", tuple(x_lat.shape), F32)
        if x_lat.data_ptr() != x_static.data_ptr():
            x_static.copy_(x_lat)
        uniform_t = not torch.is_tensor(t)
        if not self.use_graphs:
            self._launch_forward(x_static, t_buf, reps, fuser_on, sd_conv, eps_out, uniform_t)
            return eps_out
        key = (Bn, side, c["R"], c["Lc"], c["mo"], c["ms"], fuser_on, sd_conv, reps, eps_out.data_ptr(), tuple(x_lat.shape), self.precise, self.h1_f32,
               self.share_prefix, self.w3, self.in_split, uniform_t)
        g = self._graphs.get(key)
        if g is None:
            # warm-up run allocates every pooled buffer, then capture the same launch sequence
            self._launch_forward(x_static, t_buf, reps, fuser_on, sd_conv, eps_out, uniform_t)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._launch_forward(x_static, t_buf, reps, fuser_on, sd_conv, eps_out, uniform_t)
            self._graphs[key] = g
        g.replay()
        return eps_out
