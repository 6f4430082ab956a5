"""Conditioning text encoder on the MI355X kernels.

The reference encodes every prompt, the empty prompt and every relation phrase with ``FrozenCLIPEmbedder``
(GLIGEN/ldm/modules/encoders/modules.py:144-184: HuggingFace ``CLIPTokenizer`` + ``CLIPTextModel`` of
``openai/clip-vit-large-patch14``, rows padded to 77 tokens, ``last_hidden_state`` [B, 77, 768] and optionally
``pooler_output``), and every grounding phrase with ``CLIPModel(...).text_model_output.pooler_output``
(GLIGEN/interface.py:114-141, called once per phrase, :446-448).  Both are the same 12-layer causal text tower; this module
runs it on the HIP kernels (``clip.ClipTowers`` in its text-only form: fused q|k|v GEMM, causal short attention, fp32 residual
stream, fp32 ``final_layer_norm`` rows) behind the two call surfaces:

* ``HipCLIPTextEncoder.encode(texts, return_pooler_output=False)`` / ``__call__`` -- FrozenCLIPEmbedder.forward / .encode;
* ``HipCLIPTextEncoder.pooler_output(input_ids)`` -- what interface.get_clip_features_batched reads for the phrases (the
  caller's ``clip_processor`` keeps tokenising them: ``padding=True``, pads = eos id, pooled at the first eos).

Tokenisation stays host-side string work and stays the tokenizer's (a ``transformers.CLIPTokenizer``, or any callable with its
call contract).  Weights come from a state dict in any of the three key spellings that occur: ``transformer.text_model.*``
(FrozenCLIPEmbedder.state_dict(), i.e. ``saved_ckpt["text_encoder"]`` of a GLIGEN checkpoint, interface.py:88),
``text_model.*`` (CLIPModel / CLIPTextModel of transformers 4.x) or bare ``embeddings.* / encoder.*`` (CLIPTextModel of 5.x).
No fallback: a missing GPU / library raises.
"""
from __future__ import annotations

from typing import Mapping, Optional, Sequence

import torch


def normalise_text_state_dict(sd: Mapping[str, object]) -> dict:
    """-> ``text_model.*`` keys (only the text tower's tensors)."""
    out = {}
    for k, v in sd.items():
        if k.startswith("transformer."):
            k = k[len("transformer."):]
        if k.startswith(("embeddings.", "encoder.", "final_layer_norm.")):
            k = "text_model." + k
        if k.startswith("text_model.") and not k.endswith("position_ids"):
            out[k] = v
    return out


class HipCLIPTextEncoder:
    def __init__(self, state_dict: Mapping[str, object], tokenizer, device="cuda:0", heads: Optional[int] = None, max_length: int = 77):
        from .clip import ClipTowers
        sd = normalise_text_state_dict(state_dict)
        if "text_model.embeddings.token_embedding.weight" not in sd:
            raise KeyError("no CLIP text tower in the state dict (expected transformer.text_model.* / text_model.* / embeddings.* keys)")
        hidden = int(sd["text_model.embeddings.token_embedding.weight"].shape[1])
        self.heads = heads if heads is not None else hidden // 64          # CLIP text towers use 64-wide heads (ViT-L/14: 768 / 12)
        self.towers = ClipTowers(sd, text_heads=self.heads, device=device)
        self._sd = sd                      # kept (host or device, as given) for load_all_models_sharded's weight broadcast
        self.tokenizer, self.max_length = tokenizer, max_length
        self.device = self.towers.device
        self.hidden = hidden

    @staticmethod
    def accepts(state_dict: Mapping[str, object]) -> bool:
        return "text_model.embeddings.token_embedding.weight" in normalise_text_state_dict(state_dict)

    def towers_state_dict(self) -> dict:
        """``text_model.*`` fp32 tensors this encoder was built from (what the sharded loader broadcasts)."""
        return {k: torch.as_tensor(v).detach().float() for k, v in self._sd.items()}

    # FrozenCLIPEmbedder is an nn.Module: callers do .to(device).eval() on it (interface.py:86)
    def to(self, device):
        # "cuda", "cuda:0", torch.device("cuda", 0) and 0 all name the same device: compare resolved (type, index) pairs
        want, have = torch.device(device), torch.device(self.device)
        idx = lambda dv: dv.index if dv.index is not None else (torch.cuda.current_device() if dv.type == "cuda" and torch.cuda.is_available() else 0)
        if want.type != have.type or idx(want) != idx(have):
            raise RuntimeError(f"HipCLIPTextEncoder lives on {self.device}; build it on the target device")
        return self

    def eval(self):
        return self

    def tokenize(self, texts: Sequence[str]) -> torch.Tensor:
        """encoders/modules.py:160-161: truncation to max_length, padding to max_length -> ids [B, max_length] (CPU)."""
        be = self.tokenizer(list(texts), truncation=True, max_length=self.max_length, return_length=True, return_overflowing_tokens=False,
                            padding="max_length", return_tensors="pt")
        return be["input_ids"]

    @torch.no_grad()
    def encode_ids(self, input_ids: torch.Tensor, return_pooler_output: bool = False):
        z, pooled = self.towers.text_hidden_states(input_ids)
        return (z, pooled) if return_pooler_output else z

    @torch.no_grad()
    def encode(self, text, return_pooler_output: bool = False):
        """encoders/modules.py:159-174."""
        if isinstance(text, str):
            text = [text]
        return self.encode_ids(self.tokenize(text), return_pooler_output)

    __call__ = encode
    forward = encode

    @torch.no_grad()
    def encode_one_token(self, text, return_pooler_output: bool = True):
        """encoders/modules.py:176-184: ONE string tokenised without padding or truncation; ``pooler_output`` [1, hidden] (what
        GroundingNetInput.prepare's ``labels`` branch stores per box, text_layout_tokinzer_input.py:36) or ``last_hidden_state``
        [1, T, hidden]."""
        ids = self.tokenizer(text=text, padding=False, return_tensors="pt")["input_ids"]      # padding=False is CLIPTokenizer's default
        z, pooled = self.towers.text_hidden_states(ids)
        return pooled if return_pooler_output else z

    @torch.no_grad()
    def pooler_output(self, input_ids: torch.Tensor) -> torch.Tensor:
        """[B, T] token rows (any padding after the first eos) -> [B, hidden]: ``text_model_output.pooler_output`` of
        GLIGEN/interface.py:138-139."""
        return self.towers.text_hidden_states(input_ids)[1]
