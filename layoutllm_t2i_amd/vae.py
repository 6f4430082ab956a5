"""VAE decode stage on the HIP kernels (the first "next" row after the denoiser).

Replaces ``AutoencoderKL.decode`` (GLIGEN/ldm/models/autoencoder.py:40-44) and ``Decoder.forward``
(GLIGEN/ldm/modules/diffusionmodules/model.py:535-568): 1/scale_factor, 1x1 post_quant_conv, conv_in,
mid (ResnetBlock, single-head AttnBlock, ResnetBlock), 4 up levels x 3 ResnetBlocks with nearest-2x
upsample + conv, GroupNorm(eps 1e-6) + swish + conv_out.  2.5 TFLOP per 512x512 image.

No new heavy kernels: every conv / 1x1 conv / GroupNorm+swish goes through the denoiser's
``gl_conv3x3`` / ``gl_gemm`` / ``gl_groupnorm_*`` (NHWC fp16, fp32 accumulate).  The mid attention has one
head of d = 512, beyond the flash kernel's register budget; it runs once per image as
Q.K^T (GEMM) -> row softmax -> P.V (GEMM against V^T), with 1/sqrt(C) folded into the q weights at
pack time so the fp16 logits stay small.

``VAEDecoder.decode(z)`` has the reference's contract: z fp32 [B, 4, h, w] -> fp32 [B, 3, 8h, 8w].  It is a thin caller
of the C engine (``gl_vae_create`` / ``gl_vae_load_weights`` / ``gl_vae_decode``, csrc/vae_engine.hip: plan, flat weight
layout, activation pool, one hipGraph per (batch, side)); ``decode_oplevel`` is the same launch sequence issued op by op
from Python -- the test mirror the C engine must equal bitwise, like tests/engine_pyref.py for the UNet.
"""
from __future__ import annotations

from typing import Dict, Mapping

import torch

from . import ops
from ._lib import EPI_BIAS, EPI_RES, init_device
from .arch import VAEConfig, vae_decoder_param_shapes
from .weights import CIN_PAD, _h, _t, pack_conv3x3

F16, F32 = torch.float16, torch.float32


class VAEDecoder:
    def __init__(self, state_dict: Mapping[str, object], cfg: VAEConfig = VAEConfig(), device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("VAEDecoder needs a GPU: there is no CPU fallback")
        init_device()
        self.cfg, self.device = cfg, torch.device(device)
        self.scale_factor = cfg.scale_factor
        need = vae_decoder_param_shapes(cfg)
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError(f"autoencoder state_dict is missing {len(missing)} decoder tensors, e.g. {missing[:3]}")
        g = lambda k: _t(state_dict[k], self.device)
        W: Dict[str, torch.Tensor] = {}
        for name, shp in need.items():
            if not name.endswith(".weight"):
                continue
            p = name[:-7]
            w = g(name)
            b = g(p + ".bias")
            if len(shp) == 1:                       # GroupNorm affine
                W[p + ".g"], W[p + ".b"] = w.contiguous(), b.contiguous()
            elif shp[2] == 3:                       # 3x3 conv
                W[p + ".w"] = pack_conv3x3(w, CIN_PAD if shp[1] < 64 else None)
                W[p + ".b"] = b.contiguous()
            elif p == "post_quant_conv":            # 1x1 on the 4-channel latent: applied in fp32 while packing
                W[p + ".w"], W[p + ".b"] = w.reshape(shp[0], shp[1]).contiguous(), b.contiguous()
            else:                                   # 1x1 conv = GEMM
                w2, b2 = w.reshape(shp[0], shp[1]), b
                if p.endswith("attn_1.q"):          # fold the softmax scale C^-0.5 (model.py:183) into q
                    s = float(shp[0]) ** -0.5
                    w2, b2 = w2 * s, b2 * s
                W[p + ".w"], W[p + ".b"] = _h(w2), b2.contiguous()
        self.W = W
        self._pool: Dict[tuple, torch.Tensor] = {}
        self._bind_engine()

    def _bind_engine(self):
        """Moves the packed tensors into the C engine's flat layout (gl_vae_weight_at) and rebinds ``W`` to views of it."""
        from . import _lib
        self.handle = _lib.create_vae(self.cfg)
        table, total = _lib.vae_weight_table(self.handle)
        flat = torch.zeros(total, dtype=torch.uint8, device=self.device)
        views = {}
        for name, off, nbytes, dtype, shape in table:
            td = F16 if dtype == 0 else F32
            dst = flat[off:off + nbytes].view(td).view(shape)
            src = self.W[name]
            if tuple(src.shape) != tuple(shape) or src.dtype != td:
                raise ValueError(f"{name}: packed {tuple(src.shape)} {src.dtype} != engine table {shape} {td}")
            dst.copy_(src)
            views[name] = dst
        extra = set(self.W) - set(views)
        if extra:
            raise KeyError(f"packer produced tensors the VAE engine does not know: {sorted(extra)[:3]}")
        self.W, self.flat = views, flat
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gl_vae_load_weights(self.handle, flat.data_ptr(), flat.numel(), torch.cuda.current_stream(self.device).cuda_stream),
                       "gl_vae_load_weights")
        self.use_graphs = True

    def set_option(self, key: int, value: int) -> None:
        """Override one gl_set_option knob for THIS decoder only (gl_vae_set_option)."""
        from . import _lib
        _lib.check(_lib.lib().gl_vae_set_option(self.handle, int(key), int(value)), "gl_vae_set_option")

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                from . import _lib
                _lib.lib().gl_vae_destroy(h)
            except Exception:
                pass

    @classmethod
    def from_packed(cls, W: Mapping[str, torch.Tensor], cfg: VAEConfig, device="cuda:0") -> "VAEDecoder":
        """A decoder around ALREADY PACKED tensors (``self.W`` of another instance, e.g. received through
        dist.broadcast_bundle): no state_dict, no repacking."""
        if not torch.cuda.is_available():
            raise RuntimeError("VAEDecoder needs a GPU: there is no CPU fallback")
        init_device()
        self = cls.__new__(cls)
        self.cfg, self.device, self.scale_factor = cfg, torch.device(device), cfg.scale_factor
        self.W = {k: v.to(self.device) for k, v in W.items()}
        self._pool = {}
        self._bind_engine()
        return self

    def buf(self, tag, shape, dtype=F16):
        key = (tag, tuple(shape), dtype)
        t = self._pool.get(key)
        if t is None:
            t = torch.empty(tuple(shape), dtype=dtype, device=self.device)
            self._pool[key] = t
        return t

    # ---- blocks
    def _gn(self, x, B, HW, p, silu, tag):
        C = x.shape[-1]
        nchunk = ops.gn_nchunk(HW)
        partial = self.buf("gn.partial", (B * nchunk * 64,), F32)
        return ops.groupnorm(x, None, B, HW, self.W[p + ".g"], self.W[p + ".b"], 1e-6, silu, self.buf(tag, (B * HW, C)), partial)

    def _resnet(self, p, x, B, side, cin, cout, tag):
        W, HW = self.W, side * side
        t = self._gn(x, B, HW, p + ".norm1", True, f"rn.gn.{cin}.{side}")
        h = ops.conv3x3(t, W[p + ".conv1.w"], self.buf(f"rn.h.{cout}.{side}", (B * HW, cout)), B, side, side, W[p + ".conv1.b"])
        t2 = self._gn(h, B, HW, p + ".norm2", True, f"rn.gn.{cout}.{side}")
        if cin != cout:
            sk = ops.gemm(x, W[p + ".nin_shortcut.w"], self.buf(f"rn.sk.{cout}.{side}", (B * HW, cout)), W[p + ".nin_shortcut.b"])
        else:
            sk = x
        return ops.conv3x3(t2, W[p + ".conv2.w"], self.buf(tag, (B * HW, cout)), B, side, side, W[p + ".conv2.b"],
                           epi=EPI_RES, res=sk)

    def _attn(self, p, x, B, side, C, tag):
        W, N = self.W, side * side
        M = B * N
        hn = self._gn(x, B, N, p + ".norm", False, "at.gn")
        q = ops.gemm(hn, W[p + ".q.w"], self.buf("at.q", (M, C)), W[p + ".q.b"])
        k = ops.gemm(hn, W[p + ".k.w"], self.buf("at.k", (M, C)), W[p + ".k.b"])
        v = ops.gemm(hn, W[p + ".v.w"], self.buf("at.v", (M, C)), W[p + ".v.b"])
        Np = (N + 63) // 64 * 64
        Hs = 4                                              # split C only for the transpose kernel's tile
        vt = self.buf("at.vt", (B, Hs, C // Hs, Np))
        ops.transpose_v(v, N * C, C, vt, B, Hs, C // Hs, N)
        vt2 = vt.view(B, C, Np)
        s = self.buf("at.s", (B, N, Np))
        if Np != N:
            s.zero_()
        o = self.buf("at.o", (M, C))
        for b in range(B):
            sb = s[b][:, :N]
            ops.gemm(q[b * N:(b + 1) * N], k[b * N:(b + 1) * N], sb)      # logits already carry C^-0.5
            ops.softmax_rows(sb, 1.0)
            ops.gemm(s[b], vt2[b], o[b * N:(b + 1) * N])                  # P [N, Np] . V^T[C, Np]^T
        return ops.gemm(o, W[p + ".proj_out.w"], self.buf(tag, (M, C)), W[p + ".proj_out.b"], EPI_RES, res=x)

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """AutoencoderKL.decode through the C engine (one hipGraph replay per call after the first)."""
        from . import _lib
        cfg = self.cfg
        z = z.to(self.device, F32).contiguous()
        B, zc, side, side_w = z.shape
        assert side == side_w and zc == cfg.z_channels
        oside = side * 2 ** (len(cfg.ch_mult) - 1)
        out = torch.empty(B, cfg.out_ch, oside, oside, dtype=F32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().gl_vae_decode(self.handle, z.data_ptr(), B, side, out.data_ptr(), int(self.use_graphs),
                                                torch.cuda.current_stream(self.device).cuda_stream), "gl_vae_decode")
        return out

    @torch.no_grad()
    def decode_oplevel(self, z: torch.Tensor) -> torch.Tensor:
        """The engine's launch sequence issued op by op from Python (test mirror; bitwise equal to ``decode``)."""
        cfg, W = self.cfg, self.W
        z = z.to(self.device, F32).contiguous()
        B, zc, side, side_w = z.shape
        assert side == side_w and zc == cfg.z_channels
        nres = len(cfg.ch_mult)
        ch = cfg.ch * cfg.ch_mult[nres - 1]
        xin = ops.latent_affine_pack(z, W["post_quant_conv.w"], W["post_quant_conv.b"], 1.0 / cfg.scale_factor, CIN_PAD,
                                     self.buf("in", (B * side * side, CIN_PAD)))
        h = ops.conv3x3(xin, W["decoder.conv_in.w"], self.buf("conv_in", (B * side * side, ch)), B, side, side, W["decoder.conv_in.b"])
        h = self._resnet("decoder.mid.block_1", h, B, side, ch, ch, "mid.1")
        h = self._attn("decoder.mid.attn_1", h, B, side, ch, "mid.a")
        h = self._resnet("decoder.mid.block_2", h, B, side, ch, ch, "mid.2")
        for lvl in reversed(range(nres)):
            cout = cfg.ch * cfg.ch_mult[lvl]
            for i in range(cfg.num_res_blocks + 1):
                h = self._resnet(f"decoder.up.{lvl}.block.{i}", h, B, side, ch, cout, f"up.{lvl}.{i}")
                ch = cout
            if lvl != 0:
                p = f"decoder.up.{lvl}.upsample.conv"
                h = ops.conv3x3(h, W[p + ".w"], self.buf(f"up.{lvl}.u", (B * 4 * side * side, ch)), B, side, side, W[p + ".b"],
                                upsample2x=True)
                side *= 2
        g = self._gn(h, B, side * side, "decoder.norm_out", True, "fin.gn")
        out = torch.empty(B, cfg.out_ch, side, side, dtype=F32, device=self.device)
        ops.conv3x3(g, W["decoder.conv_out.w"], out, B, side, side, W["decoder.conv_out.b"], nchw_hw=side * side)
        return out
