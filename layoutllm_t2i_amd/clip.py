"""CLIP towers of the reward stage on the MI355X kernels (BASELINE.json configs[4]).

The reference's ``Reward_Model`` (models/policy.py:36-43, :106-113) keeps a HuggingFace ``transformers.CLIPModel``
(``openai/clip-vit-large-patch14``) and calls ``get_text_features`` on the captions and ``get_image_features`` on the 16
decoded rollout images and on the ground-truth images.  ``ClipTowers`` takes that model's ``state_dict()`` (transformers'
key names) and offers the same two calls, returning the UNNORMALISED projected features [B, projection_dim] fp32 like
transformers 4.19.2 (the version the reference pins) does:

* ``get_image_features(pixel_values)``  -- ``pixel_values`` [B, 3, S, S] fp32 as ``CLIPProcessor`` produces them (the
  resize / crop / normalise step stays the caller's, as in the reference);
* ``get_text_features(input_ids, attention_mask=None)`` -- pooled at ``input_ids.argmax(-1)``; a right-padding mask cannot
  change that row under the causal mask, so it is accepted and ignored.

Per encoder layer: LayerNorm (fp32 stream in, fp16 out) -> fused q|k|v projection (one gl_gemm) -> attention
(gl_attention for the vision tower's 257 bidirectional tokens, gl_attention_small for the text tower's <= 77 causal
ones) -> out projection + residual into the fp32 stream -> LayerNorm -> fc1 with the SiLU epilogue -> fc2 + residual.
quick_gelu(x) = x sigmoid(1.702 x) = silu(1.702 x) / 1.702, so fc1 is packed pre-scaled by 1.702 and fc2 by 1 / 1.702:
no new epilogue.  The residual stream is fp32 (CLIP's streams carry large outlier channels), matrix operands are fp16.
No fallback: a missing GPU / library raises.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import torch

from . import ops
from ._lib import EPI_BIAS, EPI_RES, EPI_SILU, init_device

F16, F32 = torch.float16, torch.float32
QG = 1.702


class _Tower:
    def __init__(self, sd, prefix: str, heads: int, device):
        self.heads = heads
        self.layers = []
        f = lambda k: torch.as_tensor(sd[k]).detach().to(device, F32).contiguous()
        h = lambda t: t.to(F16).contiguous()
        i = 0
        while f"{prefix}.encoder.layers.{i}.layer_norm1.weight" in sd:
            p = f"{prefix}.encoder.layers.{i}"
            L = dict(
                ln1=(f(p + ".layer_norm1.weight"), f(p + ".layer_norm1.bias")), ln2=(f(p + ".layer_norm2.weight"), f(p + ".layer_norm2.bias")),
                wqkv=h(torch.cat([f(p + f".self_attn.{n}_proj.weight") for n in "qkv"], 0)),
                bqkv=torch.cat([f(p + f".self_attn.{n}_proj.bias") for n in "qkv"], 0).contiguous(),
                wo=h(f(p + ".self_attn.out_proj.weight")), bo=f(p + ".self_attn.out_proj.bias"),
                w1=h(f(p + ".mlp.fc1.weight") * QG), b1=(f(p + ".mlp.fc1.bias") * QG).contiguous(),
                w2=h(f(p + ".mlp.fc2.weight") / QG), b2=f(p + ".mlp.fc2.bias"))
            self.layers.append(L)
            i += 1
        if not self.layers:
            raise KeyError(f"no {prefix}.encoder.layers.* tensors in the state dict")
        self.C = self.layers[0]["wo"].shape[0]
        self.d = self.C // heads
        if self.C % 64 or self.d % 8 or self.d > 64 or self.layers[0]["w1"].shape[0] % 64:
            raise NotImplementedError(f"hidden {self.C} / head dim {self.d}: need hidden % 64 == 0 and head dim % 8 == 0, <= 64")


class ClipTowers:
    def __init__(self, state_dict: Mapping[str, object], vision_heads: int = 16, text_heads: int = 12, device="cuda:0", eps: float = 1e-5):
        """``state_dict``: ``transformers.CLIPModel.state_dict()``; heads default to ViT-L/14's (16 vision, 12 text)."""
        if not torch.cuda.is_available():
            raise RuntimeError("ClipTowers needs a GPU: the reward stage has no CPU fallback")
        init_device()
        self.device, self.eps = torch.device(device), eps
        sd = state_dict
        f = lambda k: torch.as_tensor(sd[k]).detach().to(self.device, F32).contiguous()
        # a CLIPTextModel state dict (the conditioning encoder, text_encoder.py) has no vision tower and no projections
        self.vis = _Tower(sd, "vision_model", vision_heads, self.device) if "vision_model.embeddings.patch_embedding.weight" in sd else None
        self.txt = _Tower(sd, "text_model", text_heads, self.device)
        if self.vis is not None:
            pw = f("vision_model.embeddings.patch_embedding.weight")                       # [C, 3, P, P]
            self.patch = pw.shape[-1]
            K = 3 * self.patch * self.patch
            self.Kpad = (K + 63) // 64 * 64
            w = torch.zeros(pw.shape[0], self.Kpad, device=self.device, dtype=F32)
            w[:, :K] = pw.reshape(pw.shape[0], K)
            self.patch_w = w.to(F16).contiguous()
            self.cls = f("vision_model.embeddings.class_embedding")
            self.vpos = f("vision_model.embeddings.position_embedding.weight")
            self.pre_ln = (f("vision_model.pre_layrnorm.weight"), f("vision_model.pre_layrnorm.bias"))
            self.post_ln = (f("vision_model.post_layernorm.weight"), f("vision_model.post_layernorm.bias"))
            self.vproj = f("visual_projection.weight").to(F16).contiguous()
        self.tok = f("text_model.embeddings.token_embedding.weight")
        self.tpos = f("text_model.embeddings.position_embedding.weight")
        self.final_ln = (f("text_model.final_layer_norm.weight"), f("text_model.final_layer_norm.bias"))
        self.tproj = f("text_projection.weight").to(F16).contiguous() if "text_projection.weight" in sd else None
        self._pool: Dict[tuple, torch.Tensor] = {}
        # one captured launch sequence per (tower, batch, length): the towers are ~200 (vision) / ~100 (text) small launches whose
        # host-side issue through ctypes costs more than their GPU time at rollout batch sizes
        self.use_graphs = True
        self._graphs: Dict[tuple, tuple] = {}

    @staticmethod
    def _tower_of(tag: str) -> str:
        """graph keys "v" / "t" / "h" and pool tags "v.*" / "t.*": which tower a captured graph or a pooled buffer belongs to"""
        return "v" if str(tag).startswith("v") else "t"

    def buf(self, tag, shape, dtype=F16):
        key = (tag, tuple(shape), dtype)
        t = self._pool.get(key)
        if t is None:
            t = torch.empty(tuple(shape), dtype=dtype, device=self.device)
            self._pool[key] = t
        return t

    # ------------------------------------------------------------------ encoder (shared by both towers)
    def _encode(self, tw: _Tower, x: torch.Tensor, B: int, T: int, causal: bool, tag: str) -> torch.Tensor:
        """x: fp32 stream [B*T, C] (overwritten ping-pong style); returns the final stream."""
        C, H, d, M = tw.C, tw.heads, tw.d, B * T
        y = self.buf(tag + ".x2", (M, C), F32)
        hbuf = self.buf(tag + ".h", (M, C))
        qkv = self.buf(tag + ".qkv", (M, 3 * C))
        att = self.buf(tag + ".att", (M, C))
        ff = self.buf(tag + ".ff", (M, tw.layers[0]["w1"].shape[0]))
        vt = None if causal else self.buf(tag + ".vt", (B, H, d, ops.vt_ld(T)))
        for L in tw.layers:
            ops.layernorm(x, hbuf, L["ln1"][0], L["ln1"][1], B, T, eps=self.eps)
            ops.gemm(hbuf, L["wqkv"], qkv, L["bqkv"], EPI_BIAS)
            q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
            if causal:
                ops.attention_small(q, k, v, 3 * C, B, T, H, d, d ** -0.5, True, att)
            else:
                ops.transpose_v(v, T * 3 * C, 3 * C, vt, B, H, d, T)
                ops.attention(q, T * 3 * C, 3 * C, k, T * 3 * C, 3 * C, vt, att, T * C, C, B, H, d, T, T, d ** -0.5)
            ops.gemm(att, L["wo"], y, L["bo"], EPI_RES, res=x)
            ops.layernorm(y, hbuf, L["ln2"][0], L["ln2"][1], B, T, eps=self.eps)
            ops.gemm(hbuf, L["w1"], ff, L["b1"], EPI_SILU)
            ops.gemm(ff, L["w2"], x, L["b2"], EPI_RES, res=y)
        return x

    def _pool_project(self, x: torch.Tensor, rows: torch.Tensor, ln, proj: torch.Tensor, tag: str) -> torch.Tensor:
        B, C = rows.numel(), x.shape[-1]
        g = ops.clip_gather_rows(x, rows, self.buf(tag + ".g", (B, C), F32))
        n = ops.layernorm(g, self.buf(tag + ".n", (B, C)), ln[0], ln[1], B, 1, eps=self.eps)
        out = torch.empty(B, proj.shape[0], dtype=F32, device=self.device)
        return ops.gemm(n, proj, out, None, EPI_BIAS)

    # ------------------------------------------------------------------ the two calls of models/policy.py:108-113
    def _replay(self, key, static_in: torch.Tensor, fn):
        """Run ``fn(static_in)`` through a captured graph: the first call for a key runs it eagerly (allocates every pooled
        buffer) and captures it on a side stream; later calls copy the input into the static buffer and replay."""
        g = self._graphs.get(key)
        if g is None:
            # a host that keeps changing batch sizes: do not hoard graphs, nor the pooled buffers they were captured on (keyed by shape, never
            # reused by another shape: without this the pool grows monotonically over a rollout).  Eviction is per TOWER (key[0]: "v" vision,
            # "t" / "h" text): a run of new text shapes must not throw away the vision tower's graph and buffers, and vice versa
            mine = [k for k in self._graphs if self._tower_of(k[0]) == self._tower_of(key[0])]
            if len(mine) >= 8:
                torch.cuda.synchronize(self.device)
                for k in mine:
                    del self._graphs[k]
                pre = self._tower_of(key[0]) + "."
                for pk in [pk for pk in self._pool if str(pk[0]).startswith(pre)]:
                    del self._pool[pk]
            buf = static_in.clone()
            fn(buf)                                             # warm-up: pool allocations, LDS attributes
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = fn(buf)
            g = (graph, buf, out)
            self._graphs[key] = g
        graph, buf, out = g
        buf.copy_(static_in)
        graph.replay()
        return out.clone()

    def _image_features(self, px: torch.Tensor) -> torch.Tensor:
        B, _, S, _ = px.shape
        nps = S // self.patch
        T = nps * nps + 1
        C = self.vis.C
        pr = ops.clip_patchify(px, self.patch, self.Kpad, self.buf("v.patch", (B * nps * nps, self.Kpad)))
        pe = ops.gemm(pr, self.patch_w, self.buf("v.pe", (B * nps * nps, C)), None, EPI_BIAS)
        x = ops.clip_assemble(pe, self.cls, self.vpos, B, T, self.buf("v.x", (B * T, C), F32), self.pre_ln[0], self.pre_ln[1], self.eps)
        x = self._encode(self.vis, x, B, T, False, "v")
        rows = (torch.arange(B, device=self.device, dtype=torch.int32) * T).contiguous()
        return self._pool_project(x, rows, self.post_ln, self.vproj, "v")

    @torch.no_grad()
    def get_image_features(self, pixel_values: torch.Tensor) -> torch.Tensor:
        if self.vis is None:
            raise RuntimeError("this ClipTowers was built from a text-only state dict (no vision_model.* tensors)")
        px = torch.as_tensor(pixel_values).to(self.device, F32).contiguous()
        B, _, S, S2 = px.shape
        nps = S // self.patch
        if S != S2 or S % self.patch or nps * nps + 1 > self.vpos.shape[0]:
            raise ValueError(f"pixel_values {tuple(px.shape)} do not fit patch {self.patch} / {self.vpos.shape[0]} positions")
        with torch.cuda.device(self.device):
            if self.use_graphs:
                return self._replay(("v", B, S), px, self._image_features)
            return self._image_features(px)

    def _text_features(self, ids32: torch.Tensor) -> torch.Tensor:
        B, T = ids32.shape
        x = ops.clip_embed_tokens(ids32, self.tok, self.tpos, self.buf("t.x", (B * T, self.txt.C), F32))
        x = self._encode(self.txt, x, B, T, True, "t")
        rows = (torch.arange(B, device=self.device) * T + ids32.argmax(dim=-1)).to(torch.int32).contiguous()
        return self._pool_project(x, rows, self.final_ln, self.tproj, "t")

    def _text_hidden(self, ids32: torch.Tensor) -> torch.Tensor:
        """``CLIPTextModel.forward``: final_layer_norm of EVERY row, fp32 [B * T + B, C]: the B * T rows of last_hidden_state followed
        by the B pooled rows (the row at ids.argmax(-1): the first eos of a row padded with the eos id)."""
        B, T = ids32.shape
        C = self.txt.C
        x = ops.clip_embed_tokens(ids32, self.tok, self.tpos, self.buf("t.x", (B * T, C), F32))
        x = self._encode(self.txt, x, B, T, True, "t")
        out = self.buf("t.lhs", (B * T + B, C), F32)
        ops.layernorm(x, out[:B * T], self.final_ln[0], self.final_ln[1], B, T, eps=self.eps)
        rows = (torch.arange(B, device=self.device) * T + ids32.argmax(dim=-1)).to(torch.int32).contiguous()
        ops.clip_gather_rows(out[:B * T], rows, out[B * T:])
        return out

    def _bucket_ids(self, ids: torch.Tensor):
        """Token rows [B, T] -> [B8, Tb] int32 on the device, B rounded up to a multiple of 8 and T up to 16 / 32 / 48 / 64 / the position
        table; a row is padded with ITS OWN largest id (the eos id for tokenizer output), pad rows repeat row 0: the conditioning path calls
        with B = the number of relation / grounding phrases and T = the longest phrase, both different on every call, and every
        new (B, T) would cost an eager warm-up, a synchronize, a capture and its own set of pooled buffers (~1.5 MB per row).
        Under the causal mask the pad columns cannot change a real row, argmax(-1) still finds each row's FIRST eos, and the pad
        rows are dropped by the caller."""
        B, T = ids.shape
        tmax = min(int(self.tpos.shape[0]), 128)
        Tb = next((t for t in (16, 32, 48, 64) if T <= t <= tmax), tmax)      # (48 / 64: a 33-token phrase no longer pays for the whole 77-row table)
        B8 = (B + 7) // 8 * 8
        if B8 == B and Tb == T:
            return ids.to(torch.int32).contiguous()
        ids32 = ids.to(torch.int32)
        out = ids32.max(dim=-1, keepdim=True).values.expand(B, Tb).contiguous()
        out[:, :T] = ids32
        if B8 != B:
            out = torch.cat([out, out[:1].expand(B8 - B, Tb)], 0).contiguous()
        return out

    @torch.no_grad()
    def text_hidden_states(self, input_ids: torch.Tensor):
        """(last_hidden_state fp32 [B, T, C], pooler_output fp32 [B, C]) of ``transformers.CLIPTextModel`` for token rows
        [B, T] (FrozenCLIPEmbedder.forward, GLIGEN/ldm/modules/encoders/modules.py:163-170; the text branch of get_clip_feature,
        GLIGEN/interface.py:132-139)."""
        ids = torch.as_tensor(input_ids).to(self.device)
        B, T = ids.shape
        if T > self.tpos.shape[0] or T > 128:
            raise ValueError(f"sequence length {T} exceeds the position table ({self.tpos.shape[0]}) / the short-attention kernel (128)")
        with torch.cuda.device(self.device):
            ids32 = self._bucket_ids(ids)
            Bb, Tb = ids32.shape
            out = self._replay(("h", Bb, Tb), ids32, self._text_hidden) if self.use_graphs else self._text_hidden(ids32).clone()
        return out[:Bb * Tb].view(Bb, Tb, -1)[:B, :T], out[Bb * Tb:Bb * Tb + B]

    @torch.no_grad()
    def get_text_features(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.tproj is None:
            raise RuntimeError("this ClipTowers was built from a CLIPTextModel state dict (no text_projection): use text_hidden_states")
        ids = torch.as_tensor(input_ids).to(self.device)
        B, T = ids.shape
        if T > self.tpos.shape[0] or T > 128:
            raise ValueError(f"sequence length {T} exceeds the position table ({self.tpos.shape[0]}) / the short-attention kernel (128)")
        with torch.cuda.device(self.device):
            ids32 = self._bucket_ids(ids)
            if self.use_graphs:
                return self._replay(("t",) + tuple(ids32.shape), ids32, self._text_features)[:B]
            return self._text_features(ids32)[:B]
