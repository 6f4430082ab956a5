"""CLIP towers of the reward stage.  CPU: the oracle (oracle/clip_ref.py) against outputs of transformers' own
CLIPModel on a small random-init config (tests/golden/clip_tiny.npz, tools/make_clip_goldens.py).  GPU: the HIP towers
(layoutllm_t2i_amd/clip.py) against that golden and, at ViT-L/14 layer sizes, against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from layoutllm_t2i_amd import recipe
from oracle import clip_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_tiny.npz")
T = torch.from_numpy


def golden():
    z = np.load(GOLD)
    sd = {k[2:]: T(z[k]) for k in z.files if k.startswith("w:")}
    return z, sd


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def test_oracle_matches_transformers_clip_golden():
    z, sd = golden()
    heads = int(z["heads"])
    with torch.no_grad():
        img = clip_ref.image_features(sd, T(z["pixel_values"]), heads)
        txt = clip_ref.text_features(sd, T(z["input_ids"]), heads)
    assert img.shape == z["image_features"].shape and txt.shape == z["text_features"].shape
    assert float(np.abs(img.numpy() - z["image_features"]).max()) < 5e-5 * float(np.abs(z["image_features"]).max())
    assert float(np.abs(txt.numpy() - z["text_features"]).max()) < 5e-5 * float(np.abs(z["text_features"]).max())


def test_oracle_text_pooling_ignores_right_padding():
    """the pooled row (EOS = largest id) is causal: tokens after it, whatever they are, cannot change it"""
    z, sd = golden()
    ids = T(z["input_ids"]).clone()
    base = clip_ref.text_features(sd, ids, int(z["heads"]))
    eos = ids.argmax(-1)
    for b in range(ids.shape[0]):
        ids[b, int(eos[b]) + 1:] = 5
    assert torch.allclose(clip_ref.text_features(sd, ids, int(z["heads"])), base, atol=1e-6)


def vit_l_like_sd(layers_v=2, layers_t=2, seed=0):
    """ViT-L/14 layer SIZES (vision 1024 / 16 heads / 4096, 257 tokens; text 768 / 12 heads / 3072, 77 positions; projection
    768) with few layers and recipe weights scaled like a trained model (LN gains ~1, projections ~1/sqrt(fan_in))."""
    sd = {}

    def lin(name, n, k, bias=True, scale=1.0):
        sd[name + ".weight"] = T(recipe.normal(name + ".w", (n, k), seed)) * (scale / np.sqrt(k))
        if bias:
            sd[name + ".bias"] = T(recipe.normal(name + ".b", (n,), seed)) * 0.1

    def ln(name, c):
        sd[name + ".weight"] = 1.0 + 0.2 * T(recipe.normal(name + ".g", (c,), seed))
        sd[name + ".bias"] = 0.1 * T(recipe.normal(name + ".b", (c,), seed))

    def tower(prefix, n, c, inter):
        for i in range(n):
            p = f"{prefix}.encoder.layers.{i}"
            ln(p + ".layer_norm1", c)
            ln(p + ".layer_norm2", c)
            for w in "qkv":
                lin(p + f".self_attn.{w}_proj", c, c)
            lin(p + ".self_attn.out_proj", c, c, scale=0.5)
            lin(p + ".mlp.fc1", inter, c)
            lin(p + ".mlp.fc2", c, inter, scale=0.5)
    tower("vision_model", layers_v, 1024, 4096)
    tower("text_model", layers_t, 768, 3072)
    sd["vision_model.embeddings.patch_embedding.weight"] = T(recipe.normal("pe.w", (1024, 3, 14, 14), seed)) * (1.0 / np.sqrt(588))
    sd["vision_model.embeddings.class_embedding"] = T(recipe.normal("cls", (1024,), seed))
    sd["vision_model.embeddings.position_embedding.weight"] = T(recipe.normal("vpos", (257, 1024), seed)) * 0.5
    ln("vision_model.pre_layrnorm", 1024)
    ln("vision_model.post_layernorm", 1024)
    lin("visual_projection", 768, 1024, bias=False)
    sd["text_model.embeddings.token_embedding.weight"] = T(recipe.normal("tok", (1000, 768), seed))
    sd["text_model.embeddings.position_embedding.weight"] = T(recipe.normal("tpos", (77, 768), seed)) * 0.5
    ln("text_model.final_layer_norm", 768)
    lin("text_projection", 768, 768, bias=False)
    return sd


@pytest.mark.gpu
def test_hip_towers_match_transformers_golden():
    from layoutllm_t2i_amd.clip import ClipTowers
    z, sd = golden()
    heads = int(z["heads"])
    tw = ClipTowers(sd, vision_heads=heads, text_heads=heads)
    img = tw.get_image_features(T(z["pixel_values"]))
    txt = tw.get_text_features(T(z["input_ids"]), attention_mask=(T(z["input_ids"]) != 1).long())
    ri, rt = rel(img, T(z["image_features"])), rel(txt, T(z["text_features"]))
    print(f"[clip_tiny vs transformers] image rel_l2={ri:.3e} text rel_l2={rt:.3e}")
    assert img.shape == z["image_features"].shape and txt.dtype == torch.float32
    assert ri < 3e-3 and rt < 3e-3, (ri, rt)            # fp16 matrix operands incl. fp16-rounded weights, fp32 stream
    # captured-graph replays (the default) == the eager launch sequence, bitwise, also on new inputs of the same shape
    px2 = T(z["pixel_values"]).flip(0) * 0.5
    ids2 = T(z["input_ids"]).flip(0)
    g_img, g_txt, g_img2, g_txt2 = img, txt, tw.get_image_features(px2), tw.get_text_features(ids2)
    # (text rows are bucketed: batch rounded up to 8, length to 16 / 32 / 48 / 64 / the position table -- clip.ClipTowers._bucket_ids)
    assert ("v", px2.shape[0], px2.shape[-1]) in tw._graphs and ("t", 8, 16) in tw._graphs and len(tw._graphs) == 2
    tw.use_graphs = False
    assert torch.equal(tw.get_image_features(T(z["pixel_values"])), g_img) and torch.equal(tw.get_text_features(T(z["input_ids"])), g_txt)
    assert torch.equal(tw.get_image_features(px2), g_img2) and torch.equal(tw.get_text_features(ids2), g_txt2)
    assert not torch.equal(g_img, g_img2)


@pytest.mark.gpu
def test_hip_towers_at_vit_l14_sizes_vs_oracle():
    """257 vision tokens x 1024 x 16 heads (d = 64) and 77 causal text tokens x 768 x 12 heads, ragged batch sizes; oracle
    on the fp16-rounded weights (isolates arithmetic from weight quantisation)."""
    from layoutllm_t2i_amd.clip import ClipTowers
    sd = vit_l_like_sd()
    sdh = {k: (v.half().float() if v.dim() >= 2 else v) for k, v in sd.items()}
    tw = ClipTowers(sd)
    px = T(recipe.normal("px", (3, 3, 224, 224), 1))
    ids = torch.randint(1, 998, (5, 77), generator=torch.Generator().manual_seed(3))
    for b, L in enumerate((76, 10, 33, 5, 50)):
        ids[b, L] = 999
        ids[b, L + 1:] = 0
    img, txt = tw.get_image_features(px), tw.get_text_features(ids)
    torch.set_num_threads(min(32, max(1, os.cpu_count() or 1)))
    with torch.no_grad():
        ri = rel(img, clip_ref.image_features(sdh, px.half().float(), 16))
        rt = rel(txt, clip_ref.text_features(sdh, ids, 12))
    print(f"[ViT-L/14-size layers vs oracle] image rel_l2={ri:.3e} text rel_l2={rt:.3e}")
    assert img.shape == (3, 768) and txt.shape == (5, 768)
    assert ri < 2e-3 and rt < 2e-3, (ri, rt)
    # deterministic, and a sample does not depend on its batch neighbours
    assert torch.equal(img, tw.get_image_features(px))
    assert rel(tw.get_image_features(px[1:2]), img[1:2]) < 1e-3


@pytest.mark.gpu
def test_reward_model_forward_vs_oracle_pipeline():
    """Reward_Model.forward's GPU part end to end (models/policy.py:106-124,135): towers -> F.normalize -> similarities ->
    aesthetic MLP -> reward, HIP (RewardModel) vs oracle (clip_ref + reward_ref) on the transformers-generated tiny CLIP."""
    from layoutllm_t2i_amd.reward import RewardModel
    from oracle import reward_ref
    z, sd = golden()
    heads, D = int(z["heads"]), 64
    shapes = {0: (1024, D), 2: (128, 1024), 4: (64, 128), 6: (16, 64), 7: (1, 16)}
    aes = {}
    for li, (n, k) in shapes.items():
        aes[f"layers.{li}.weight"] = T(recipe.uniform(f"aes.{li}.w", (n, k), 2)) * float(np.sqrt(3.0 / k))
        aes[f"layers.{li}.bias"] = T(recipe.uniform(f"aes.{li}.b", (n,), 2)) * 0.1
    rm = RewardModel(sd, aes, vision_heads=heads, text_heads=heads)
    ids = T(z["input_ids"])[:3]
    pred, gt = T(z["pixel_values"]), T(recipe.normal("gtpx", (3, 3, 42, 42), 4))
    miou, laysim = torch.tensor([0.1, 0.5, 0.3]), torch.tensor([0.2, 0.0, 0.9])
    out = rm(ids, pred, gt, miou=miou, laysim=laysim)
    with torch.no_grad():
        ref = reward_ref.reward_scores(aes, clip_ref.text_features(sd, ids, heads), clip_ref.image_features(sd, pred, heads),
                                       clip_ref.image_features(sd, gt, heads), miou, laysim)
    for k in ("sims_ti", "sims_ii", "aes_reward", "reward"):
        d = float((out[k].cpu() - ref[k]).abs().max())
        print(f"[reward model] {k}: max|diff|={d:.3e} |ref|max={float(ref[k].abs().max()):.3f}")
        assert d < 5e-3 * max(1.0, float(ref[k].abs().max())), (k, d)


@pytest.mark.gpu
def test_reward_model_from_images_equals_the_pixel_values_path():
    """forward_images: decoded fp32 predictions (GPU) and mixed-size ground-truth images go through the HIP preprocessing
    (uint8 conversion, Pillow-exact bicubic resize, crop, normalise) -- the rewards must equal, BITWISE, forward() on the
    pixel_values the CPU restatement of the reference's feature extractor produces for the same images."""
    from layoutllm_t2i_amd.reward import RewardModel
    from oracle import clip_preprocess_ref as ppref
    z, sd = golden()
    heads, D = int(z["heads"]), 64
    shapes = {0: (1024, D), 2: (128, 1024), 4: (64, 128), 6: (16, 64), 7: (1, 16)}
    aes = {}
    for li, (n, k) in shapes.items():
        aes[f"layers.{li}.weight"] = T(recipe.uniform(f"aes.{li}.w", (n, k), 2)) * float(np.sqrt(3.0 / k))
        aes[f"layers.{li}.bias"] = T(recipe.uniform(f"aes.{li}.b", (n,), 2)) * 0.1
    rm = RewardModel(sd, aes, vision_heads=heads, text_heads=heads, image_size=42)
    ids = T(z["input_ids"])[:3]
    dec = T(recipe.normal("decoded", (3, 3, 96, 96), 9)) * 0.8
    rng = np.random.default_rng(5)
    gts = [rng.integers(0, 256, s_, dtype=np.uint8) for s_ in ((60, 80, 3), (96, 96, 3), (60, 80, 3))]
    out = rm.forward_images(ids, dec.to("cuda:0"), gts)
    px_pred = np.stack([ppref.clip_feature_extractor(u, 42, 42)[0] for u in ppref.decoded_to_u8(dec)])
    px_gt = np.stack([ppref.clip_feature_extractor(g, 42, 42)[0] for g in gts])
    want = rm(ids, T(px_pred), T(px_gt))
    for k in ("sims_ti", "sims_ii", "aes_reward", "reward"):
        assert torch.equal(out[k], want[k]), k
    # a decoded fp32 batch that happens to have the crop size is still an IMAGE batch (it used to be taken for ready
    # pixel_values by its shape); ready pixel_values are passed through only when the caller says so
    dec42 = T(recipe.normal("decoded42", (3, 3, 42, 42), 9)) * 0.8
    out42 = rm.forward_images(ids, dec42.to("cuda:0"), gts)
    px42 = np.stack([ppref.clip_feature_extractor(u, 42, 42)[0] for u in ppref.decoded_to_u8(dec42)])
    want42 = rm(ids, T(px42), T(px_gt))
    ready = rm.forward_images(ids, T(px42), T(px_gt), pred_is_pixel_values=True, gt_is_pixel_values=True)
    for k in ("sims_ti", "sims_ii", "aes_reward", "reward"):
        assert torch.equal(out42[k], want42[k]) and torch.equal(ready[k], want42[k]), k


# ------------------------------------------------------------------------------------------------ conditioning text encoder
GOLD_TEXT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_text_tiny.npz")


def test_oracle_text_hidden_states_match_transformers_cliptextmodel_golden():
    """oracle/clip_ref.text_hidden_states against transformers' own CLIPTextModel (what FrozenCLIPEmbedder wraps,
    encoders/modules.py:144-174) on rows padded with the eos id: last_hidden_state and pooler_output (first eos)."""
    z, sd = golden()
    zt = np.load(GOLD_TEXT)
    ids = T(zt["input_ids"])
    with torch.no_grad():
        lhs, pooled = clip_ref.text_hidden_states(sd, ids, int(z["heads"]))
    scale = float(np.abs(zt["last_hidden_state"]).max())
    assert float(np.abs(lhs.numpy() - zt["last_hidden_state"]).max()) < 5e-5 * scale
    assert float(np.abs(pooled.numpy() - zt["pooler_output"]).max()) < 5e-5 * scale
    first_eos = (ids == ids.max()).int().argmax(-1)
    assert torch.equal(pooled, lhs[torch.arange(ids.shape[0]), first_eos])


@pytest.mark.gpu
@pytest.mark.parametrize("spelling", ["text_model", "transformer.text_model", "bare"])
def test_hip_text_encoder_vs_transformers_cliptextmodel_golden(spelling):
    """text_encoder.HipCLIPTextEncoder (the HIP text tower behind FrozenCLIPEmbedder's call surface) against the CLIPTextModel
    golden, built from each of the three state-dict key spellings that occur in the wild."""
    from layoutllm_t2i_amd.text_encoder import HipCLIPTextEncoder
    z, sd = golden()
    zt = np.load(GOLD_TEXT)
    tsd = {k: v for k, v in sd.items() if k.startswith("text_model.")}
    if spelling == "transformer.text_model":
        tsd = {"transformer." + k: v for k, v in tsd.items()}
    elif spelling == "bare":
        tsd = {k[len("text_model."):]: v for k, v in tsd.items()}
    enc = HipCLIPTextEncoder(tsd, None, "cuda:0", heads=int(z["heads"]), max_length=zt["input_ids"].shape[1])
    ids = T(zt["input_ids"])
    lhs, pooled = enc.encode_ids(ids, return_pooler_output=True)
    assert lhs.dtype == torch.float32 and lhs.shape == zt["last_hidden_state"].shape and pooled.shape == zt["pooler_output"].shape
    rl, rp = rel(lhs, T(zt["last_hidden_state"])), rel(pooled, T(zt["pooler_output"]))
    print(f"[hip text encoder vs CLIPTextModel, {spelling}] last_hidden_state rel_l2={rl:.3e} pooler_output rel_l2={rp:.3e}")
    assert rl < 3e-3 and rp < 3e-3, (rl, rp)            # fp16 matrix operands of 4x-scaled weights, as for the reward towers (1.4e-3)
    first_eos = (ids == ids.max()).int().argmax(-1)
    assert torch.equal(pooled.cpu(), lhs.cpu()[torch.arange(ids.shape[0]), first_eos])
    # graph replay == eager, and the phrase path (rows padded only to the longest phrase) gives the same pooled rows
    enc.towers.use_graphs = False
    lhs2, pooled2 = enc.encode_ids(ids, return_pooler_output=True)
    enc.towers.use_graphs = True
    assert torch.equal(lhs, lhs2) and torch.equal(pooled, pooled2)
    L = int(first_eos.max()) + 1
    short = enc.pooler_output(ids[:, :L])
    assert rel(short, pooled) < 2e-3                     # other GEMM row counts: fp32 summation order only


@pytest.mark.gpu
def test_hip_text_encoder_encode_contract_and_oracle_at_vit_l_sizes():
    """``encode(list[str], return_pooler_output)`` (FrozenCLIPEmbedder.forward, encoders/modules.py:159-174) through a tokenizer
    with CLIPTokenizer's call contract, at ViT-L/14's text sizes (768 wide, 12 heads, 77 positions) against the oracle."""
    import stubs
    from layoutllm_t2i_amd.text_encoder import HipCLIPTextEncoder
    sd = stubs.toy_text_tower_state_dict(1)
    tok = stubs.ToyTokenizer()
    enc = HipCLIPTextEncoder({"transformer." + k: v for k, v in sd.items()}, tok, "cuda:0")
    assert enc.heads == 12 and enc.hidden == 768
    texts = ["cat sitting on mat and dog under a tree", "", "a quiet empty street", "PAD"]
    z = enc.encode(texts)
    z2, pooled = enc(texts, return_pooler_output=True)
    assert z.shape == (4, 77, 768) and torch.equal(z, z2) and pooled.shape == (4, 768) and z.is_cuda
    ids = tok(texts)["input_ids"]
    with torch.no_grad():
        want, wpool = clip_ref.text_hidden_states(sd, ids, 12)
    rl, rp = rel(z, want), rel(pooled, wpool)
    print(f"[hip text encoder, 768 x 12 heads x 77] rel_l2={rl:.3e} pooled rel_l2={rp:.3e}")
    assert rl < 2e-3 and rp < 2e-3, (rl, rp)
    one = enc.encode("a quiet empty street")                                  # a single string is a batch of one
    assert one.shape == (1, 77, 768) and rel(one[0], z[2]) < 1e-3             # (other GEMM row counts: fp32 summation order only)


@pytest.mark.gpu
def test_encode_one_token_and_grounding_input_labels_branch():
    """FrozenCLIPEmbedder.encode_one_token (encoders/modules.py:176-184) on the HIP tower, and the branch of
    GroundingNetInput.prepare that calls it when a batch carries ``labels`` instead of ``text_embeddings``
    (text_layout_tokinzer_input.py:29-39): (i) on the CLIPTextModel golden's weights, one unpadded row per call -> that row's
    golden ``pooler_output``; (ii) at ViT-L/14's text sizes through prepare(), against the oracle tower."""
    import stubs
    from layoutllm_t2i_amd.model import GroundingNetInput
    from layoutllm_t2i_amd.text_encoder import HipCLIPTextEncoder
    z, sd = golden()
    zt = np.load(GOLD_TEXT)
    ids = T(zt["input_ids"])
    first_eos = (ids == ids.max()).int().argmax(-1)

    class RowTokenizer:                   # "3" -> golden row 3 cut after its first eos (no padding), CLIPTokenizer's call contract
        def __call__(self, text=None, padding=False, return_tensors="pt", **kw):
            r = int(text)
            return {"input_ids": ids[r:r + 1, :int(first_eos[r]) + 1]}
    enc = HipCLIPTextEncoder({k: v for k, v in sd.items() if k.startswith("text_model.")}, RowTokenizer(), "cuda", heads=int(z["heads"]),
                             max_length=ids.shape[1])
    assert enc.to("cuda") is enc and enc.to(torch.device("cuda", torch.cuda.current_device())) is enc and enc.to("cuda:0") is enc
    for r in range(ids.shape[0]):
        pooled = enc.encode_one_token(str(r))
        assert pooled.shape == (1, zt["pooler_output"].shape[1])
        assert rel(pooled[0], T(zt["pooler_output"][r])) < 3e-3, r
        lhs = enc.encode_one_token(str(r), return_pooler_output=False)
        assert lhs.shape == (1, int(first_eos[r]) + 1, zt["pooler_output"].shape[1])
        assert rel(lhs[0], T(zt["last_hidden_state"][r, :int(first_eos[r]) + 1])) < 3e-3, r
    # (ii) prepare() with labels: 768-wide tower (in_dim is hard-coded 768 there), two samples with 2 and 1 boxes
    tsd = stubs.toy_text_tower_state_dict(2)
    tok = stubs.ToyTokenizer()
    enc768 = HipCLIPTextEncoder(tsd, tok, "cuda:0")
    boxes = torch.zeros(2, 30, 4, device="cuda:0")
    masks = torch.zeros(2, 30, device="cuda:0")
    masks[0, :2] = 1
    masks[1, :1] = 1
    labels = ["red apple|a dog", "tall tree|unused"]
    g = GroundingNetInput()
    out = g.prepare(dict(boxes=boxes, masks=masks, labels=labels), text_encoder=enc768)
    pe = out["positive_embeddings"]
    assert pe.shape == (2, 30, 768) and out["boxes"] is boxes and out["masks"] is masks
    for (b, i), text in {(0, 0): "red apple", (0, 1): "a dog", (1, 0): "tall tree"}.items():
        with torch.no_grad():
            want = clip_ref.text_hidden_states(tsd, tok(text, padding=False)["input_ids"], 12)[1][0]
        assert rel(pe[b, i], want) < 2e-3, (b, i)
    assert float(pe[1, 1:].abs().max()) == 0.0 and float(pe[0, 2:].abs().max()) == 0.0
    null = g.get_null_input()
    assert null["positive_embeddings"].shape == (2, 30, 768) and float(null["positive_embeddings"].abs().max()) == 0.0
