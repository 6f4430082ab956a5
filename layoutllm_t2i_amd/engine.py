"""Python face of the MI355X UNet engine.  The engine itself is C++ behind the forward-level C ABI of
``include/gligen_hip.h`` (``csrc/engine.hip``): block plan, packed-weight layout, activation pool, hoisted
conditioning, launch sequence and hipGraph capture / replay all live behind ``gl_create`` / ``gl_load_weights`` /
``gl_set_conditioning`` / ``gl_unet_forward`` / ``gl_plms_step``.  This class only turns torch tensors (device memory
and streams) into raw pointers for those calls -- it replaces ``UNetModel.forward`` (openaimodel.py:413-459) for the
Python callers (``model.UNetModel``, ``sampler.PLMSSampler``).

No torch op touches activations on the hot path, and there is no fallback: a missing library, a CPU tensor or a
missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import PlmsStepArgs, check, init_device
from .arch import UNetConfig
from .weights import PackedWeights

F16, F32 = torch.float16, torch.float32


class UNetEngine:
    def __init__(self, packed: PackedWeights):
        if not torch.cuda.is_available():
            raise RuntimeError("UNetEngine needs a GPU: the denoising path has no CPU fallback")
        if packed.flat is None or not packed.flat.is_cuda:
            raise _lib.HipLibraryError("packed weights must be a flat buffer on the GPU (PackedWeights.to_flat on a cuda device)")
        self.P = packed
        self.cfg: UNetConfig = packed.cfg
        self.plan = packed.plan
        self.dev = torch.device(packed.device)
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.W, self.S = packed.w, packed.s
        init_device(self.dev.index)
        self._lib = _lib.lib()
        self.handle = _lib.create_engine(self.cfg)
        with torch.cuda.device(self.dev):
            check(self._lib.gl_load_weights(self.handle, packed.flat.data_ptr(), packed.flat.numel(), int(packed.has_sd_conv), self._stream()),
                  "gl_load_weights")
        self._pool: Dict[Tuple, torch.Tensor] = {}
        self.cond: Optional[dict] = None
        self._cond_refs: Tuple = ()
        self.use_graphs = True

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                self._lib.gl_destroy(h)
            except Exception:
                pass

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.dev).cuda_stream

    def set_option(self, key: int, value: int) -> None:
        """Override one gl_set_option knob for THIS engine only (gl_set_handle_option): in effect while its entry points run,
        invisible to other engines and to op-level calls; its captured graphs are rebuilt on the next forward."""
        check(self._lib.gl_set_handle_option(self.handle, int(key), int(value)), "gl_set_handle_option")

    def clear_options(self) -> None:
        check(self._lib.gl_clear_handle_options(self.handle), "gl_clear_handle_options")

    # ------------------------------------------------------------------ torch-side scratch (sampler state, outputs)
    def buf(self, tag: str, shape, dtype=F16, zero: bool = False) -> torch.Tensor:
        key = (tag, tuple(shape), dtype)
        t = self._pool.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.dev)
            self._pool[key] = t
        return t

    def pool_bytes(self) -> int:
        """bytes of the engine's activation pool (C side) + the torch-side sampler buffers"""
        return int(self._lib.gl_pool_bytes(self.handle)) + sum(t.numel() * t.element_size() for t in self._pool.values())

    def num_launches(self) -> int:
        """kernel launches of one forward (as counted by the last eager / capture pass)"""
        return int(self._lib.gl_num_launches(self.handle))

    # ------------------------------------------------------------------ conditioning (once per image)
    @torch.no_grad()
    def set_conditioning(self, context, relations, boxes, masks, positive_embeddings, hw: int) -> None:
        """context [Bn,77,ctx], relations [Bn,R,ctx], boxes [Bn,30,4], masks [Bn,30],
        positive_embeddings [Bn,30,in_dim]: fp32 tensors (any device).  Null grounding = zeros
        (text_layout_tokinzer_input.py:47-62).  ``hw`` = latent side (64 for 512x512)."""
        dev = self.dev
        f32 = lambda t: torch.as_tensor(t, dtype=F32).to(dev).contiguous()
        context, relations = f32(context), f32(relations)
        boxes, masks, pe = f32(boxes), f32(masks), f32(positive_embeddings)
        Bn, mo = boxes.shape[0], boxes.shape[1]
        if mo != self.cfg.max_objs:
            raise ValueError(f"{mo} grounding slots, the model was built for {self.cfg.max_objs}")
        if context.shape[-1] != self.cfg.context_dim or relations.shape[-1] != self.cfg.context_dim or pe.shape[-1] != self.cfg.pos_in_dim:
            raise ValueError("conditioning feature dims do not match the model config")
        if not (context.shape[0] == relations.shape[0] == masks.shape[0] == pe.shape[0] == Bn):
            raise ValueError("conditioning batch sizes differ")
        R, Lc = relations.shape[1], context.shape[1]
        with torch.cuda.device(dev):
            check(self._lib.gl_set_conditioning(self.handle, context.data_ptr(), relations.data_ptr(), boxes.data_ptr(), masks.data_ptr(),
                                                pe.data_ptr(), Bn, Lc, R, hw, self._stream()), "gl_set_conditioning")
        self._cond_refs = (context, relations, boxes, masks, pe)      # read asynchronously by the launched kernels
        self.cond = dict(Bn=Bn, mo=mo, R=R, Lc=Lc, hw=hw)

    # ------------------------------------------------------------------ forward
    def _check_x(self, x_lat: torch.Tensor, reps: int):
        c = self.cond
        if c is None:
            raise RuntimeError("call set_conditioning() first")
        Bn, side = c["Bn"], c["hw"]
        if x_lat.shape[0] * reps != Bn or x_lat.shape[-1] != side or x_lat.shape[1] != self.cfg.in_channels:
            raise ValueError(f"latent batch {tuple(x_lat.shape)} x reps {reps} does not match conditioning batch {Bn} @ {side}")
        if not x_lat.is_cuda or x_lat.dtype != F32 or not x_lat.is_contiguous():
            raise _lib.HipLibraryError("latent must be a contiguous fp32 GPU tensor (the HIP path has no CPU fallback)")
        return Bn, side

    @torch.no_grad()
    def forward(self, x_lat: torch.Tensor, t, fuser_scale: float = 1.0, sd_conv: bool = False, reps: int = 1,
                eps_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_lat fp32 [Bn/reps, 4, hw, hw] (NCHW, like the reference); returns eps fp32 [Bn, 4, hw, hw].
        With reps=2 the latent is shared by both CFG halves of a [cond ; uncond] conditioning batch."""
        Bn, side = self._check_x(x_lat, reps)
        if sd_conv and not self.P.has_sd_conv:
            raise RuntimeError("SD first-conv weights were not packed")
        if eps_out is None:
            eps_out = self.buf("out.eps", (Bn, self.cfg.out_channels, side, side), F32)
        t_dev, t_host, keep = None, 0.0, None
        if torch.is_tensor(t):
            if t.numel() == 1:
                t_host = float(t.reshape(-1)[0])
            else:
                keep = t.to(self.dev, F32).reshape(-1).contiguous()
                if keep.numel() != Bn:
                    raise ValueError("timesteps must have one entry per sample")
                t_dev = keep.data_ptr()
        else:
            t_host = float(t)
        with torch.cuda.device(self.dev):
            check(self._lib.gl_unet_forward(self.handle, x_lat.data_ptr(), t_dev, t_host, reps, float(fuser_scale), int(bool(sd_conv)),
                                            eps_out.data_ptr(), int(self.use_graphs), self._stream()), "gl_unet_forward")
        return eps_out

    @torch.no_grad()
    def plms_step(self, x_eval: torch.Tensor, x_base: torch.Tensor, x_out: torch.Tensor, e_out: torch.Tensor, e_terms: List[torch.Tensor],
                  coefs, div: float, t: float, reps: int, guidance: float, fuser_scale: float, sd_conv: bool, sqrt_at: float, s1m: float,
                  sqrt_aprev: float, dir_coef: float) -> torch.Tensor:
        """One gl_plms_step: UNet(x_eval, t) -> CFG combine into ``e_out`` -> e' = sum coefs[j] * e_terms[j] / div ->
        x_out = PLMS / DDIM(sigma 0) update of x_base (plms.py:110-163)."""
        self._check_x(x_eval, reps)
        a = PlmsStepArgs()
        for tns, name in ((x_eval, "x_eval"), (x_base, "x_base"), (x_out, "x_out"), (e_out, "e_out"), *[(e, "e_term") for e in e_terms]):
            if not tns.is_cuda or tns.dtype != F32 or not tns.is_contiguous() or tns.shape != x_eval.shape:
                raise _lib.HipLibraryError(f"{name}: need a contiguous fp32 GPU tensor of the latent's shape")
        a.x_eval, a.x_base, a.x_out, a.e_out = x_eval.data_ptr(), x_base.data_ptr(), x_out.data_ptr(), e_out.data_ptr()
        if not 1 <= len(e_terms) <= 4 or len(coefs) != len(e_terms):
            raise ValueError("1..4 eps terms with one coefficient each")
        for j, (e, c) in enumerate(zip(e_terms, coefs)):
            a.e_terms[j] = e.data_ptr()
            a.coef[j] = float(c)
        a.n_terms, a.div = len(e_terms), float(div)
        a.t, a.reps, a.guidance, a.fuser_scale, a.sd_conv = float(t), int(reps), float(guidance), float(fuser_scale), int(bool(sd_conv))
        a.sqrt_at, a.s1m, a.sqrt_aprev, a.dir_coef = float(sqrt_at), float(s1m), float(sqrt_aprev), float(dir_coef)
        a.use_graph = int(self.use_graphs)
        if sd_conv and not self.P.has_sd_conv:
            raise RuntimeError("SD first-conv weights were not packed")
        with torch.cuda.device(self.dev):
            check(self._lib.gl_plms_step(self.handle, C.byref(a), self._stream()), "gl_plms_step")
        return x_out
