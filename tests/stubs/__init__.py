"""Test doubles for the modules that sit OUTSIDE the denoising path (the CLIP text side, sng_parser and
checkpoints are unobtainable offline): a deterministic text encoder with FrozenCLIPEmbedder's call contract, a tiny
randomly-initialised HF CLIP + whitespace processor, a rule-based ``sng_parser`` and a synthetic GLIGEN checkpoint
writer with the reference's container layout (interface.py:79-94)."""
from __future__ import annotations

import hashlib
import os
import sys
import types

import numpy as np
import torch


class StubTextEncoder:
    """``encode(list[str], return_pooler_output=False) -> [n, 77, 768]`` (encoders/modules.py:159-184): a pure
    function of each string, so the test's own restatement of the conditioning gets identical tensors."""

    def __init__(self, dim: int = 768, max_length: int = 77):
        self.dim, self.max_length, self.device = dim, max_length, "cpu"
        self.loaded = None

    def to(self, device):
        self.device = device
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd, strict=True):
        self.loaded = dict(sd)

    def _one(self, text: str) -> torch.Tensor:
        key = int.from_bytes(hashlib.sha256(text.encode()).digest()[:8], "little")
        bits = np.random.Philox(key=key).random_raw(4 * self.max_length * self.dim)
        u = (bits >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
        return torch.from_numpy(u.reshape(4, self.max_length, self.dim).sum(0) * np.float32(np.sqrt(3.0 / 4.0)))

    def encode(self, texts, return_pooler_output=False):
        z = torch.stack([self._one(t) for t in texts]).to(self.device)
        if return_pooler_output:
            return z, z[:, -1, :].clone()
        return z

    __call__ = encode


class ToyProcessor:
    """whitespace tokenizer with CLIP's conventions: BOS first, EOS = highest id, padded with EOS, mask 0 on pads"""
    BOS, EOS = 98, 99

    def __call__(self, text=None, return_tensors="pt", padding=True, images=None):
        if isinstance(text, str):
            text = [text]
        rows = [[self.BOS] + [1 + (sum(map(ord, w)) % 90) for w in t.split()] + [self.EOS] for t in text]
        n = max(map(len, rows))
        ids = torch.tensor([r + [self.EOS] * (n - len(r)) for r in rows])
        am = torch.tensor([[1] * len(r) + [0] * (n - len(r)) for r in rows])
        return {"input_ids": ids, "attention_mask": am}


class ToyTokenizer:
    """``CLIPTokenizer``'s call contract as FrozenCLIPEmbedder uses it (encoders/modules.py:160-161): rows truncated / padded
    to ``max_length`` with the EOS id; same word hashing as ToyProcessor."""
    BOS, EOS = 98, 99

    def __call__(self, text, truncation=True, max_length=77, return_length=True, return_overflowing_tokens=False, padding="max_length",
                 return_tensors="pt"):
        if isinstance(text, str):
            text = [text]
        rows = [([self.BOS] + [1 + (sum(map(ord, w)) % 90) for w in t.split()])[:max_length - 1] + [self.EOS] for t in text]
        if padding is False:          # CLIPTokenizer's default, as FrozenCLIPEmbedder.encode_one_token calls it (encoders/modules.py:177)
            n = max(map(len, rows))
            return {"input_ids": torch.tensor([r + [self.EOS] * (n - len(r)) for r in rows], dtype=torch.long)}
        return {"input_ids": torch.tensor([r + [self.EOS] * (max_length - len(r)) for r in rows], dtype=torch.long)}


def toy_text_tower_state_dict(seed: int = 0, hidden: int = 768, heads: int = 12, layers: int = 2, inter: int = 256, vocab: int = 100,
                              positions: int = 77):
    """``text_model.*`` fp32 tensors of a small CLIP text tower (transformers' key names) with recipe weights: LN gains ~1,
    projections ~ 1/sqrt(fan_in), so every term of the tower matters."""
    from layoutllm_t2i_amd import recipe
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sd = {}

    def lin(name, n, k, scale=1.0):
        sd[name + ".weight"] = T(recipe.normal("tt." + name + ".w", (n, k), seed)) * float(scale / np.sqrt(k))
        sd[name + ".bias"] = T(recipe.normal("tt." + name + ".b", (n,), seed)) * 0.1

    def ln(name):
        sd[name + ".weight"] = 1.0 + 0.1 * T(recipe.normal("tt." + name + ".g", (hidden,), seed))
        sd[name + ".bias"] = 0.05 * T(recipe.normal("tt." + name + ".b", (hidden,), seed))
    sd["text_model.embeddings.token_embedding.weight"] = T(recipe.normal("tt.tok", (vocab, hidden), seed)) * 0.5
    sd["text_model.embeddings.position_embedding.weight"] = T(recipe.normal("tt.pos", (positions, hidden), seed)) * 0.3
    for i in range(layers):
        p = f"text_model.encoder.layers.{i}"
        for n in "qkv":
            lin(f"{p}.self_attn.{n}_proj", hidden, hidden)
        lin(f"{p}.self_attn.out_proj", hidden, hidden, 0.7)
        ln(f"{p}.layer_norm1")
        ln(f"{p}.layer_norm2")
        lin(f"{p}.mlp.fc1", inter, hidden)
        lin(f"{p}.mlp.fc2", hidden, inter, 0.7)
    ln("text_model.final_layer_norm")
    return sd


def toy_clip(text_heads: int = 4):
    from transformers import CLIPConfig, CLIPModel
    rng = torch.random.get_rng_state()
    torch.manual_seed(0)
    cfg = CLIPConfig(text_config=dict(hidden_size=768, intermediate_size=128, num_hidden_layers=2, num_attention_heads=text_heads,
                                      vocab_size=100, max_position_embeddings=16, eos_token_id=99, bos_token_id=98,
                                      pad_token_id=99),
                     vision_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                        image_size=224, patch_size=112), projection_dim=768)
    m = CLIPModel(cfg).eval()
    torch.random.set_rng_state(rng)           # building the toy model must not disturb the caller's global RNG
    return m


def install_fake_sng_parser():
    """``sng_parser.parse(prompt)`` -> {'entities': [{'lemma_head'}], 'relations': [{'subject','relation','object'}]}
    (the fields interface.py:225-236 reads).  Rule: clauses split on ' and '; 'A <words> B' with >= 3 words is one relation."""
    mod = types.ModuleType("sng_parser")

    def parse(prompt):
        ents, rels = [], []
        for clause in prompt.split(" and "):
            w = clause.split()
            if len(w) >= 3:
                ents.append({"lemma_head": w[0]})
                ents.append({"lemma_head": w[-1]})
                rels.append({"subject": len(ents) - 2, "relation": " ".join(w[1:-1]), "object": len(ents) - 1})
        return {"entities": ents, "relations": rels}
    mod.parse = parse
    sys.modules["sng_parser"] = mod
    return mod


def write_synthetic_checkpoint(path, cfg, vae_cfg, seed=0, max_relations=10, with_sd_conv=True, clip_text_tower=False):
    """A checkpoint with the reference's container: {model, autoencoder, text_encoder, diffusion, config_dict._content}
    (interface.py:79-94) holding recipe weights of ``cfg`` / ``vae_cfg``; the text encoder node targets StubTextEncoder, or,
    with ``clip_text_tower``, is FrozenCLIPEmbedder's with a ``transformer.text_model.*`` state dict of a small CLIP text tower
    (what a real GLIGEN checkpoint holds) and a ToyTokenizer node."""
    from layoutllm_t2i_amd import recipe
    t = lambda d: {k: torch.tensor(np.asarray(v, dtype=np.float32)) for k, v in d.items()}     # 0-d gates stay 0-d
    content = {
        "model": {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                  "params": {"image_size": cfg.image_size, "in_channels": cfg.in_channels, "model_channels": cfg.model_channels,
                             "out_channels": cfg.out_channels, "num_res_blocks": cfg.num_res_blocks,
                             "attention_resolutions": list(cfg.attention_resolutions), "channel_mult": list(cfg.channel_mult),
                             "num_heads": cfg.num_heads, "context_dim": cfg.context_dim, "fuser_type": "gatedSA",
                             "use_checkpoint": True, "transformer_depth": 1,
                             "grounding_tokenizer": {"target": "ldm.modules.diffusionmodules.text_grounding_net.PositionNet",
                                                     "params": {"in_dim": cfg.pos_in_dim, "out_dim": cfg.pos_out_dim}}}},
        "autoencoder": {"target": "ldm.models.autoencoder.AutoencoderKL",
                        "params": {"scale_factor": vae_cfg.scale_factor, "embed_dim": vae_cfg.embed_dim,
                                   "ddconfig": {"double_z": True, "z_channels": vae_cfg.z_channels, "resolution": 256, "in_channels": 3,
                                                "out_ch": vae_cfg.out_ch, "ch": vae_cfg.ch, "ch_mult": list(vae_cfg.ch_mult),
                                                "num_res_blocks": vae_cfg.num_res_blocks, "attn_resolutions": [], "dropout": 0.0}}},
        "text_encoder": ({"target": "ldm.modules.encoders.modules.FrozenCLIPEmbedder",
                          "params": {"max_length": 77, "num_attention_heads": 12, "tokenizer": {"target": "stubs.ToyTokenizer"}}}
                         if clip_text_tower else {"target": "stubs.StubTextEncoder", "params": {}}),
        "diffusion": {"target": "ldm.models.diffusion.ldm.LatentDiffusion",
                      "params": {"linear_start": 0.00085, "linear_end": 0.012, "timesteps": 1000}},
        "grounding_tokenizer_input": {"target": "grounding_input.text_layout_tokinzer_input.GroundingNetInput"},
        "max_relations": max_relations,
    }
    ckpt = {"model": t(recipe.state_dict(cfg, seed)), "autoencoder": t(recipe.vae_state_dict(vae_cfg, seed)),
            "text_encoder": ({"transformer." + k: v for k, v in toy_text_tower_state_dict(seed).items()} if clip_text_tower
                             else {"dummy": torch.zeros(1)}),
            "diffusion": {}, "config_dict": {"_content": content}}
    torch.save(ckpt, path)
    if with_sd_conv:
        fc = recipe.sd_first_conv(cfg, seed)
        torch.save({"weight": torch.from_numpy(fc["weight"]), "bias": torch.from_numpy(fc["bias"])},
                   os.path.join(os.path.dirname(os.path.abspath(path)), "SD_input_conv_weight_bias.pth"))
    return ckpt
