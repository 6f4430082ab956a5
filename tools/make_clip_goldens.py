#!/usr/bin/env python3
"""Build-container script: golden vectors for the CLIP towers of the reward stage.

The towers live in a third-party dependency of the reference (``transformers.CLIPModel``, models/policy.py:36-43; pinned
4.19.2 in env_docker/Dockerfile:3), not under /root/reference, so the pin is transformers' OWN implementation as installed
here: a small random-init CLIPConfig (head dim 64 like ViT-L/14, quick_gelu, eos = the largest token id) is run through
``get_image_features`` / ``get_text_features`` and weights + inputs + outputs are stored in tests/golden/clip_tiny.npz.

    python tools/make_clip_goldens.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def feats(out):
    return out if torch.is_tensor(out) else out.pooler_output        # transformers >= 5 wraps the projected features


def main():
    import transformers
    from transformers import CLIPConfig, CLIPModel
    torch.manual_seed(1234)
    VOCAB = 99
    cfg = CLIPConfig(
        text_config=dict(vocab_size=VOCAB, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                         max_position_embeddings=16, hidden_act="quick_gelu", eos_token_id=2, bos_token_id=0, pad_token_id=1),
        vision_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=42,
                           patch_size=14, hidden_act="quick_gelu"),
        projection_dim=64)
    model = CLIPModel(cfg).eval()
    with torch.no_grad():                      # default init is tiny (std 0.02): widen it so every term matters
        for n, p in model.named_parameters():
            if p.dim() >= 2:
                p.mul_(4.0)
            elif "bias" in n or "class_embedding" in n:
                p.add_(torch.randn_like(p) * 0.2)
    g = torch.Generator().manual_seed(7)
    pixel_values = torch.randn(3, 3, 42, 42, generator=g)
    # rows: <bos> tokens ... <eos = largest id> <pad...>; argmax(-1) is the EOS position (transformers 4.19.2 pooling)
    ids = torch.randint(3, VOCAB - 1, (4, 12), generator=g)
    ids[:, 0] = 0
    for b, L in enumerate((11, 5, 8, 3)):
        ids[b, L] = VOCAB - 1
        ids[b, L + 1:] = 1
    am = (ids != 1).long()
    with torch.no_grad():
        img = feats(model.get_image_features(pixel_values=pixel_values))
        txt = feats(model.get_text_features(input_ids=ids, attention_mask=am))
        txt_nomask = feats(model.get_text_features(input_ids=ids))
    assert torch.allclose(txt, txt_nomask, atol=1e-6), "right padding must not change the pooled text feature"
    out = {"w:" + k: v.detach().numpy() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    out.update(pixel_values=pixel_values.numpy(), input_ids=ids.numpy(), image_features=img.numpy(), text_features=txt.numpy(),
               heads=np.asarray(2), transformers_version=np.asarray(transformers.__version__))
    dst = os.path.join(ROOT, "tests", "golden", "clip_tiny.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(out), "arrays; |img|max", float(img.abs().max()), "|txt|max", float(txt.abs().max()))


if __name__ == "__main__":
    main()
