#!/usr/bin/env python3
"""Build-container script: golden vectors for the CONDITIONING text encoder -- tests/golden/clip_text_tiny.npz.

The reference's ``FrozenCLIPEmbedder`` (GLIGEN/ldm/modules/encoders/modules.py:144-184) wraps ``transformers.CLIPTextModel``
and returns ``last_hidden_state`` [B, 77, 768] (+ ``pooler_output``) of prompts padded to ``max_length``; the grounding
phrases go through ``CLIPModel(...).text_model_output.pooler_output`` (GLIGEN/interface.py:114-141).  Both are the SAME text
tower of a third-party dependency that is not under /root/reference (transformers, pinned 4.19.2 in env_docker/Dockerfile:3),
so the pin is transformers' OWN implementation as installed here: the text-tower weights already stored in
tests/golden/clip_tiny.npz (tools/make_clip_goldens.py) are loaded into a ``CLIPTextModel`` of the same small config and run on
token rows padded the way CLIP's tokenizer pads them (pad id == eos id == the largest id, so ``argmax`` must pick the FIRST one).
Stored: the token ids and the two outputs (weights are not duplicated).

    python tools/make_clip_text_goldens.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel
    z = np.load(os.path.join(ROOT, "tests", "golden", "clip_tiny.npz"))
    VOCAB, T = z["w:text_model.embeddings.token_embedding.weight"].shape[0], z["w:text_model.embeddings.position_embedding.weight"].shape[0]
    cfg = CLIPTextConfig(vocab_size=VOCAB, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=int(z["heads"]),
                         max_position_embeddings=T, hidden_act="quick_gelu", eos_token_id=2, bos_token_id=0, pad_token_id=1)
    model = CLIPTextModel(cfg).eval()
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:text_model.")}
    if not any(k.startswith("text_model.") for k in model.state_dict()):        # transformers >= 5 flattens CLIPTextModel's keys
        sd = {k[len("text_model."):]: v for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(3, VOCAB - 1, (5, T), generator=g)
    ids[:, 0] = 0
    for b, L in enumerate((T - 1, 5, 9, 2, 1)):          # <bos> tokens <eos> then pads that EQUAL the eos id (max_length padding)
        ids[b, L:] = VOCAB - 1
    with torch.no_grad():
        out = model(input_ids=ids)
    lhs, pooled = out.last_hidden_state, out.pooler_output
    first_eos = (ids == VOCAB - 1).int().argmax(-1)
    assert torch.equal(pooled, lhs[torch.arange(ids.shape[0]), first_eos]), "pooler_output = hidden state at the FIRST eos"
    dst = os.path.join(ROOT, "tests", "golden", "clip_text_tiny.npz")
    np.savez_compressed(dst, input_ids=ids.numpy(), last_hidden_state=lhs.numpy(), pooler_output=pooled.numpy(),
                        transformers_version=np.asarray(transformers.__version__))
    print("wrote", dst, os.path.getsize(dst), "bytes; |lhs|max", float(lhs.abs().max()))


if __name__ == "__main__":
    main()
