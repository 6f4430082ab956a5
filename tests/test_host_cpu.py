"""CPU tests (no GPU): C-ABI library loads and exports what include/gligen_hip.h declares, host logic
(box rectangles, schedules, packing layouts, sampler tables), and loud failure without a GPU."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import golden_cases as gc
from layoutllm_t2i_amd import _lib, arch, host, recipe
from layoutllm_t2i_amd.arch import TINY, UNetConfig
from layoutllm_t2i_amd.weights import geglu_interleave, pack_conv3x3, pack_state_dict
from oracle import plms_ref, unet_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------------------------------- C ABI
def _ensure_built():
    if not os.path.exists(_lib.LIB_PATH):
        from layoutllm_t2i_amd.csrc.build import build
        build(verbose=False)


def test_library_exports_every_declared_symbol():
    _ensure_built()
    hdr = open(os.path.join(ROOT, "include", "gligen_hip.h")).read()
    declared = set(re.findall(r"\b(?:int|int64_t)\s+(gl_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 19, declared
    l = _lib.lib()                      # also checks ABI version + struct sizes
    for name in declared:
        assert hasattr(l, name), f"{name} declared in gligen_hip.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert l.gl_abi_version() == _lib.ABI_VERSION == 15
    assert l.gl_sizeof_gemm_args() == ctypes.sizeof(_lib.GemmArgs)
    assert l.gl_sizeof_conv_args() == ctypes.sizeof(_lib.ConvArgs)
    assert l.gl_sizeof_attn_args() == ctypes.sizeof(_lib.AttnArgs)
    assert l.gl_sizeof_unet_config() == ctypes.sizeof(_lib.UNetConfigC)
    assert l.gl_sizeof_weight_info() == ctypes.sizeof(_lib.WeightInfo)
    assert l.gl_sizeof_plms_step_args() == ctypes.sizeof(_lib.PlmsStepArgs)


def test_engine_handle_plan_and_weight_table_without_a_gpu():
    """gl_create builds the block plan and the packed-weight table on the host: the table must name exactly the tensors
    the Python packer produces (same shapes / dtypes), in 256-byte aligned non-overlapping slots, for the real config,
    the tiny one and a one-level variant; bad configs are rejected; compute entry points refuse an unloaded handle."""
    _ensure_built()
    l = _lib.lib()
    for cfg in (UNetConfig(), TINY, UNetConfig(image_size=32, model_channels=640, channel_mult=(1,), attention_resolutions=(1,), num_res_blocks=1)):
        h = _lib.create_engine(cfg)
        table, total = _lib.weight_table(h)
        names = [t[0] for t in table]
        assert len(set(names)) == len(names)
        end = 0
        for name, off, nbytes, dtype, shape in table:
            assert off % 256 == 0 and off >= end
            n = int(np.prod(shape))
            assert nbytes == n * (2 if dtype == 0 else 4)
            end = off + nbytes
        assert total >= end and total % 256 == 0
        if cfg is TINY:
            P = pack_state_dict(recipe.state_dict(TINY, 0), TINY, "cpu", recipe.sd_first_conv(TINY, 0))
            assert set(P.w) == set(names) and P.flat.numel() == total
            for name, off, nbytes, dtype, shape in table:
                assert tuple(P.w[name].shape) == tuple(shape)
                assert P.w[name].data_ptr() == P.flat.data_ptr() + off          # views into the flat buffer
        if cfg == UNetConfig():
            assert 2.4e9 < total < 2.7e9 and len(names) == 1102
        # the split_weights table (strict mode): the same names, every matrix twice as wide ([Whi | Wlo] rows; the first conv and the
        # three kinds of 1x1 conv keep their own split forms), and the packer fills it: Whi == the compact table's matrix, Whi + Wlo ~ W
        import dataclasses
        hs = _lib.create_engine(dataclasses.replace(cfg, split_weights=True))
        table_s, total_s = _lib.weight_table(hs)
        assert [t[0] for t in table_s] == names and total < total_s < 2 * total
        for (name, _, _, dtype, shape), (_, _, _, dtype_s, shape_s) in zip(table, table_s):
            assert dtype == dtype_s
            if dtype == 0 and len(shape) == 2 and not name.startswith(("input_blocks.0.0.", "sd_first_conv.")) \
                    and not name.endswith((".skip_connection.w", ".proj_in.w", ".proj_out.w")):
                assert tuple(shape_s) == (shape[0], 2 * shape[1]), name
            else:
                assert tuple(shape_s) == tuple(shape), name
        assert l.gl_destroy(hs) == 0
        if cfg is TINY:
            Ps = pack_state_dict(recipe.state_dict(TINY, 0), dataclasses.replace(TINY, split_weights=True), "cpu", recipe.sd_first_conv(TINY, 0))
            assert Ps.flat.numel() == total_s
            for name, _, _, dtype, shape in table:
                if tuple(Ps.w[name].shape) != tuple(shape):
                    k = shape[1]
                    assert torch.equal(Ps.w[name][:, :k], P.w[name]), name
                    assert float(Ps.w[name][:, k:].float().abs().max()) < 2.0 ** -10 * float(P.w[name].float().abs().max()), name
            sd0 = recipe.state_dict(TINY, 0)
            w32 = torch.from_numpy(np.asarray(sd0["output_blocks.2.0.in_layers.2.weight"])).float()
            from layoutllm_t2i_amd.weights import pack_conv3x3
            got = Ps.w["output_blocks.2.0.in_layers.2.w"].float()
            k = got.shape[1] // 2
            want = pack_conv3x3(w32).float() + pack_conv3x3(w32 - w32.half().float()).float()
            assert torch.equal(got[:, :k] + got[:, k:], want)
        # nothing loaded / no conditioning: the compute entry points must refuse, not crash
        assert l.gl_unet_forward(h, None, None, 0.0, 1, 1.0, 0, None, 1, None) == -1
        assert l.gl_set_conditioning(h, None, None, None, None, None, 1, 77, 10, 16, None) == -1
        assert l.gl_destroy(h) == 0
    bad = _lib.unet_config_c(TINY)
    bad.model_channels = 48
    hh = ctypes.c_void_p()
    assert l.gl_create(ctypes.byref(bad), ctypes.byref(hh)) == -2
    bad = _lib.unet_config_c(TINY)
    bad.n_levels = 9
    assert l.gl_create(ctypes.byref(bad), ctypes.byref(hh)) == -1


def test_bad_arguments_are_rejected_without_a_gpu():
    _ensure_built()
    l = _lib.lib()
    g = _lib.GemmArgs()
    assert l.gl_gemm(ctypes.byref(g), None) == -1          # null pointers -> GL_ERR_BAD_ARG, no launch
    a = _lib.AttnArgs()
    assert l.gl_attention(ctypes.byref(a), None) == -1
    assert l.gl_layernorm(None, 0, 0, None, 0, None, None, 1, 1, 1, 0, 64, 1e-5, None, None, 0, 0, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from layoutllm_t2i_amd import ops
    from layoutllm_t2i_amd.model import UNetModel
    with pytest.raises((_lib.HipLibraryError, RuntimeError, AssertionError)):
        UNetModel(TINY, recipe.state_dict(TINY, 0), device="cuda:0")
    with pytest.raises(_lib.HipLibraryError):
        ops.gemm(torch.zeros(64, 64, dtype=torch.float16), torch.zeros(64, 64, dtype=torch.float16),
                 torch.zeros(64, 64, dtype=torch.float16))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "layoutllm_t2i_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), fn


# ------------------------------------------------------------------------------------------- arch / packing
def test_param_table_full_model():
    shapes = arch.param_shapes(UNetConfig())
    assert len(shapes) == 1238
    assert arch.count_params(UNetConfig()) == 1_261_457_796
    assert shapes["input_blocks.0.0.weight"] == (320, 4, 3, 3)
    assert shapes["output_blocks.11.1.transformer_blocks.0.rela_fuse.attn.to_k.weight"] == (320, 768)
    plan = arch.build_plan(UNetConfig())
    assert len(plan.res_layers()) == 22 and len(plan.st_layers()) == 16


def test_geglu_interleave_layout():
    t = torch.arange(2 * 64).float()
    p = geglu_interleave(t)
    assert p[:32].tolist() == list(range(0, 32)) and p[32:64].tolist() == list(range(64, 96))
    assert p[64:96].tolist() == list(range(32, 64)) and p[96:].tolist() == list(range(96, 128))


def test_conv_pack_layout():
    w = (torch.arange(2 * 128 * 9) % 1021).float().reshape(2, 128, 3, 3)   # exactly representable in fp16
    p = pack_conv3x3(w).float().reshape(2, 2, 3, 3, 64)          # [Cout, channel block, ky, kx, channel]
    for cb in range(2):
        for ky in range(3):
            for kx in range(3):
                assert torch.equal(p[:, cb, ky, kx, :], w[:, cb * 64:(cb + 1) * 64, ky, kx])
    w4 = torch.arange(2 * 3 * 9).float().reshape(2, 3, 3, 3) + 1
    p4 = pack_conv3x3(w4, cin_pad=64).float().reshape(2, 1, 3, 3, 64)
    for ky in range(3):
        for kx in range(3):
            assert torch.equal(p4[:, 0, ky, kx, :3], w4[:, :, ky, kx]) and float(p4[:, 0, ky, kx, 3:].abs().max()) == 0


def test_pack_state_dict_cpu_tiny():
    P = pack_state_dict(recipe.state_dict(TINY, 0), TINY, "cpu", recipe.sd_first_conv(TINY, 0))
    assert P.w["emb_all.w"].shape == (P.emb_total, TINY.time_embed_dim)
    assert P.emb_total == sum(l.cout for l in P.plan.res_layers())
    assert P.w["input_blocks.0.0.w"].shape == (64, 9 * 64)
    t = "input_blocks.1.1.transformer_blocks.0"
    assert P.w[t + ".attn1.qkv.w"].shape == (192, 64) and P.w[t + ".attn2.kv.w"].shape == (128, 768)
    assert abs(P.s[t + ".fuser.tanh_attn"]) > 0.25
    with pytest.raises(KeyError):
        pack_state_dict({}, TINY, "cpu")


# ------------------------------------------------------------------------------------------- host logic
def _closed_form(hid, f, rects, nvalid, poison, mo):
    B, H, W, C = hid.shape
    acc = torch.zeros_like(hid)
    for b in range(B):
        for i in range(int(nvalid[b])):
            t, bo, l, r = [int(v) for v in rects[b, i]]
            acc[b, t:bo, l:r] += f[b, i]
    out = hid + acc / mo
    for b in range(B):
        if poison[b]:
            out[b] = float("nan")
    return out


@pytest.mark.parametrize("variant,hw", [("normal", 8), ("degenerate", 8), ("null", 8), ("clamp", 8), ("maskgap", 16), ("empty_slice", 8)])
def test_box_rects_closed_form_equals_reference(variant, hw):
    """host.box_rects + the closed form used by the HIP kernels reproduce the reference's
    RelationCrossAttention goldens (break rule, clamping, python slices, NaN poison)."""
    name = {"normal": "rela_normal", "degenerate": "rela_degenerate", "null": "rela_null", "clamp": "rela_clamp",
            "maskgap": "rela_maskgap", "empty_slice": "rela_empty_slice"}[variant]
    case = next(c for c in gc.CASES if c["name"] == name)
    inp = {a: torch.from_numpy(v) for a, v in gc.case_inputs(case).items()}
    C, heads, mo = case["C"], case["heads"], 30
    sd = {"r." + n: torch.from_numpy(np.asarray(recipe.tensor(f"golden.{name}.{n}", s, 0)))
          for n, s in arch.rela_params("", C, gc.CTX).items()}
    rects, nvalid, poison = host.box_rects(inp["boxes"].numpy(), inp["masks"].numpy(), hw, hw)
    B = inp["x"].shape[0]
    hid = torch.nn.functional.layer_norm(inp["x"], (C,), sd["r.norm3.weight"], sd["r.norm3.bias"]).view(B, hw, hw, C)
    feat = torch.zeros(B, mo, C)
    for b in range(B):
        for i in range(int(nvalid[b])):
            t, bo, l, r = [int(v) for v in rects[b, i]]
            feat[b, i] = hid[b, t:bo, l:r].reshape(-1, C).mean(0)
    with torch.no_grad():
        f = feat + torch.tanh(sd["r.alpha_attn"]) * unet_ref.attention(
            sd, "r.attn", torch.nn.functional.layer_norm(feat, (C,), sd["r.norm1.weight"], sd["r.norm1.bias"]),
            inp["relations"], inp["relations"], heads)
        f = f + torch.tanh(sd["r.alpha_dense"]) * unet_ref.feed_forward(
            sd, "r.ff", torch.nn.functional.layer_norm(f, (C,), sd["r.norm2.weight"], sd["r.norm2.bias"]))
    out = _closed_form(hid, f, rects, nvalid, poison, mo).view(B, hw * hw, C).numpy()
    ref = np.load(os.path.join(GOLD, name + ".npz"))["out"]
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4, equal_nan=True)


def test_box_rects_edge_semantics():
    b = np.zeros((1, 30, 4), np.float32)
    m = np.zeros((1, 30), np.float32)
    b[0, 0] = (0.999, 0.0, 1.0, 1.0)      # x0 = int(63.936) = 63, x1 = 64 -> 1 px wide at w = 64
    b[0, 1] = (0.5, 0.5, 0.5, 0.9)        # zero width -> break
    b[0, 2] = (0.1, 0.1, 0.9, 0.9)
    m[0, :3] = 1
    rects, nvalid, poison = host.box_rects(b, m, 64, 64)
    assert nvalid[0] == 1 and poison[0] == 0 and tuple(rects[0, 0]) == (0, 64, 63, 64)
    b[0, 0] = (-0.5, 0.0, 0.5, 0.5)       # negative x0: int(-32.0) = -32 -> python slice(-32, 32) on w = 64 == [32, 32) -> empty
    rects, nvalid, poison = host.box_rects(b, m, 64, 64)
    assert nvalid[0] == 1 and poison[0] == 1


def test_schedule_and_alpha_match_reference_goldens():
    for S in (10, 50):
        g = np.load(os.path.join(GOLD, f"schedule_s{S}.npz"))
        acp = host.alphas_cumprod()
        np.testing.assert_array_equal(acp, g["alphas_cumprod"])
        s = host.make_schedule(S, acp)
        np.testing.assert_array_equal(s["ddim_timesteps"], g["ddim_timesteps"])
        np.testing.assert_array_equal(s["ddim_alphas"], g["ddim_alphas"])
        np.testing.assert_array_equal(s["ddim_alphas_prev"], g["ddim_alphas_prev"])
        np.testing.assert_array_equal(s["ddim_sqrt_one_minus_alphas"], g["ddim_sqrt_one_minus_alphas"])
    g = np.load(os.path.join(GOLD, "alpha_gen.npz"))
    assert host.alpha_generator(50, [0.3, 0.0, 0.7]) == list(g["a50"])
    assert host.alpha_generator(20, [0.5, 0.25, 0.25]) == pytest.approx(list(g["a20"]))
    assert host.alpha_generator(7, None) == list(g["a7"])
    # 102 UNet evaluations per image at S = 50 (step 0 twice, cond + uncond)
    assert 2 * (50 + 1) == 102


def test_interface_surface_matches_reference_names():
    import inspect
    from layoutllm_t2i_amd import interface as I
    from layoutllm_t2i_amd.sampler import PLMSSampler
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(I.generate_batch_images) == ["all_models", "captions", "labels", "bboxes", "clip_model", "clip_processor", "device"]
    assert sig(I.generate_one_image) == ["all_models", "caption", "label", "bbox", "clip_model", "clip_processor", "device"]
    assert sig(I.run_batch_images) == ["all_models", "args", "meta", "starting_noise", "clip_model", "clip_processor", "device"]
    assert sig(I.run_one_image) == sig(I.run_batch_images)
    # the reference's positional parameters, plus this build's optional keyword (default None = GLIGEN_STRICT from the environment)
    assert sig(I.load_all_models) == ["ckpt", "device", "strict"] and sig(I.load_ckpt) == ["ckpt_path", "device", "strict"]
    assert inspect.signature(I.load_all_models).parameters["strict"].default is None and inspect.signature(I.load_ckpt).parameters["strict"].default is None
    assert sig(PLMSSampler.sample) == ["self", "S", "shape", "input", "uc", "guidance_scale", "mask", "x0"]
    assert sig(PLMSSampler.__init__) == ["self", "diffusion", "model", "schedule", "alpha_generator_func", "set_alpha_scale"]
    assert I.convert_xywh_to_ltrb([0.1, 0.2, 0.3, 0.4]) == pytest.approx([0.1, 0.2, 0.4, 0.6])

    class M:
        fuser_scale = 1
    m = M()
    I.set_alpha_scale(m, 0)
    assert m.fuser_scale == 0


# ---------------------------------------------------------------- conditioning prep batching
class _ToyProcessor:
    """whitespace tokenizer with CLIP's conventions: BOS first, EOS = highest id, padded with EOS, mask 0 on pads"""
    BOS, EOS = 98, 99

    def __call__(self, text, return_tensors="pt", padding=True):
        import torch
        if isinstance(text, str):
            text = [text]
        rows = [[self.BOS] + [1 + (sum(map(ord, w)) % 90) for w in t.split()] + [self.EOS] for t in text]
        n = max(map(len, rows))
        ids = torch.tensor([r + [self.EOS] * (n - len(r)) for r in rows])
        am = torch.tensor([[1] * len(r) + [0] * (n - len(r)) for r in rows])
        return {"input_ids": ids, "attention_mask": am}


def _toy_clip():
    import torch
    from transformers import CLIPConfig, CLIPModel
    torch.manual_seed(0)
    cfg = CLIPConfig(text_config=dict(hidden_size=768, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                      vocab_size=100, max_position_embeddings=16, eos_token_id=99, bos_token_id=98,
                                      pad_token_id=99),
                     vision_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                        image_size=224, patch_size=112), projection_dim=768)
    return CLIPModel(cfg).eval()


def test_batched_phrase_features_equal_per_phrase_reference_semantics():
    """prepare_batch_multiple with ONE padded CLIP forward for the whole batch must reproduce the reference's
    one-forward-per-phrase loop (interface.py:424-475): same boxes / masks, embeddings equal to fp32 rounding."""
    import torch
    from layoutllm_t2i_amd import interface as itf
    model, proc = _toy_clip(), _ToyProcessor()
    calls = {"n": 0}
    fwd = model.text_model.forward

    def counting(*a, **k):
        calls["n"] += 1
        return fwd(*a, **k)
    model.text_model.forward = counting              # the text tower is what interface.py:136-139 reads ('before' projection)
    meta = {"phrases": [["a red dog", "tree", "a very tall old tree"], ["tree", "sky"]],
            "locations": [[[0.1, 0.1, 0.5, 0.5], [0.2, 0.3, 0.9, 0.8], [0.0, 0.0, 1.0, 1.0]], [[0.3, 0.3, 0.6, 0.6], [0.0, 0.0, 1.0, 0.4]]]}
    with torch.no_grad():
        out = itf.prepare_batch_multiple(meta, model, proc, batch=2, max_objs=30, device="cpu")
        assert calls["n"] == 1                       # 4 distinct phrases, one forward (reference: 5 forwards)
        for b, phrases in enumerate(meta["phrases"]):
            for i, ph in enumerate(phrases):
                ref = itf.get_clip_feature(model, proc, ph, "cpu")          # the reference's per-phrase call
                torch.testing.assert_close(out["text_embeddings"][b, i:i + 1], ref, rtol=1e-5, atol=1e-5)
                # ... and, independently of interface.py, HuggingFace's own text tower on the UNPADDED phrase: the pooled hidden state
                # before text_projection (transformers 4.19.2's text_model_output.pooler_output, interface.py:115,139)
                ids = proc(text=ph)["input_ids"]
                hf = fwd(input_ids=ids).pooler_output
                torch.testing.assert_close(out["text_embeddings"][b, i:i + 1], hf, rtol=1e-5, atol=1e-5)
            n = len(phrases)
            assert out["masks"][b].tolist() == [1.0] * n + [0.0] * (30 - n)
            assert out["text_masks"][b].tolist() == [1.0] * n + [0.0] * (30 - n)
            assert torch.equal(out["boxes"][b, :n], torch.tensor(meta["locations"][b]))
            assert float(out["text_embeddings"][b, n:].abs().max()) == 0.0
        assert out["image_masks"].abs().max() == 0 and out["image_embeddings"].abs().max() == 0
        one = itf.prepare_batch({"phrases": meta["phrases"][0], "locations": meta["locations"][0]}, model, proc, batch=3, device="cpu")
        torch.testing.assert_close(one["text_embeddings"][2], out["text_embeddings"][0], rtol=1e-5, atol=1e-5)


def test_relation_phrases_batched_equals_per_prompt():
    """all prompts' triplets in one encode call == the reference's prompt-by-prompt encode (interface.py:221-252)"""
    import torch
    from layoutllm_t2i_amd import interface as itf

    class Enc:                                   # deterministic stand-in for FrozenCLIPEmbedder.encode
        calls = 0

        def encode(self, texts, return_pooler_output=False):
            Enc.calls += 1
            pooled = torch.stack([torch.full((768,), float(sum(map(ord, t)) % 97)) + torch.arange(768) * 1e-3 for t in texts])
            z = pooled[:, None, :].repeat(1, 77, 1)
            return (z, pooled) if return_pooler_output else z

    table = {"a cat on a mat": ["cat on mat"], "nothing here": [], "a dog under a tree near a car": ["dog under tree", "tree near car"]}
    parse = lambda prompt, max_relas: ((["PAD"] + table[prompt] * 2)[:max_relas] if table[prompt] else [])
    prompts = list(table)
    enc = Enc()
    out = itf.prepare_relation_phrases_batch(prompts, 4, enc, "cpu", parse=parse)
    assert Enc.calls == 1 and out.shape == (3, 4, 768)
    for i, p in enumerate(prompts):
        one = itf.prepare_relation_phrases_batch([p], 4, enc, "cpu", parse=parse)
        assert torch.equal(out[i], one[0])
        n = len(parse(p, 4))
        assert float(out[i, n:].abs().max() if n < 4 else 0.0) == 0.0
    assert float(out[1].abs().max()) == 0.0                       # no relation -> all-zero rows (interface.py:241-243)
    assert torch.equal(out[2, 0], out[2, 0]) and out[2, 3, 0] == out[2, 1, 0]   # PAD, r1, r2, r1 (truncated at 4)


def test_text_encoder_state_dict_spellings_and_tokenize_conditioning_host_logic():
    """SURVEY 8f-2 host side (no GPU): the three state-dict key spellings of a CLIP text tower normalise to ``text_model.*``; the
    sharded entry's string half (tokenize_conditioning) turns prompts / relation phrases / grounding phrases into token rows and
    index tables that reproduce the reference flow's structure: 'PAD' + every relation twice truncated to max_relations
    (interface.py:226-243), boxes / masks padded to 30 slots (:460-475), one row per DISTINCT phrase."""
    import torch
    import stubs
    from layoutllm_t2i_amd import interface as itf
    from layoutllm_t2i_amd.text_encoder import HipCLIPTextEncoder, normalise_text_state_dict
    sd = stubs.toy_text_tower_state_dict(0, hidden=64, heads=1, layers=1, inter=64, vocab=100, positions=16)
    for spell in (lambda k: k, lambda k: "transformer." + k, lambda k: k[len("text_model."):]):
        n = normalise_text_state_dict({spell(k): v for k, v in sd.items()})
        assert set(n) == set(sd) and all(torch.equal(n[k], sd[k]) for k in sd)
    assert HipCLIPTextEncoder.accepts({"transformer." + k: v for k, v in sd.items()}) and not HipCLIPTextEncoder.accepts({"dummy": torch.zeros(1)})
    extra = dict(sd, **{"text_model.embeddings.position_ids": torch.arange(16)[None], "vision_model.x": torch.zeros(1), "text_projection.weight": torch.zeros(2, 2)})
    assert set(normalise_text_state_dict(extra)) == set(sd)            # only the text tower's tensors, no position_ids buffer

    class Enc:                                   # tokenizer side of HipCLIPTextEncoder only (no GPU here)
        max_length, tokenizer = 77, stubs.ToyTokenizer()
        tokenize = HipCLIPTextEncoder.tokenize
    stubs.install_fake_sng_parser()
    prompts = ["cat sitting on mat and dog under a tree", "a quiet empty street", "bird on wire"]
    phrases = [["cat", "mat", "a big dog"], None, ["bird", "bird"]]
    boxes = [[[0.1, 0.1, 0.5, 0.55], [0.05, 0.6, 0.95, 0.95], [0.55, 0.2, 0.9, 0.7]], [[0.0, 0.5, 1.0, 1.0]], [[0.1, 0.2, 0.3, 0.4], [0.6, 0.2, 0.8, 0.4]]]
    tok = itf.tokenize_conditioning((None, None, Enc(), None, {"max_relations": 4}), prompts, phrases, boxes, stubs.ToyProcessor())
    assert tok["cap_ids"].shape == (3, 77) and tok["uc_ids"].shape == (1, 77) and int(tok["uc_ids"][0, 0]) == 98 and int(tok["uc_ids"][0, 1]) == 99
    # prompt 0 has two relations -> PAD, r1, r2, r1 (truncated to 4); prompts 1 and 2 one each -> PAD, r, r (the stub parser's rule)
    assert tok["rel_owner"].tolist() == [0, 0, 0, 0, 1, 1, 1, 2, 2, 2] and tok["rel_slot"].tolist() == [0, 1, 2, 3, 0, 1, 2, 0, 1, 2]
    assert tok["rel_ids"].shape == (10, 77) and torch.equal(tok["rel_ids"][0], tok["rel_ids"][4])         # both start with "PAD"
    assert torch.equal(tok["rel_ids"][5], tok["rel_ids"][6]) and torch.equal(tok["rel_ids"][8], tok["rel_ids"][9])
    assert torch.equal(tok["rel_ids"][1], tok["rel_ids"][3]) and not torch.equal(tok["rel_ids"][1], tok["rel_ids"][2])
    assert tok["phrase_ids"].shape[0] == 4                                                                  # cat, mat, a big dog, bird
    assert tok["phrase_index"][0, :4].tolist() == [0, 1, 2, -1] and tok["phrase_index"][1, :2].tolist() == [-1, -1]
    assert tok["phrase_index"][2, :3].tolist() == [3, 3, -1]
    assert tok["masks"].sum(-1).tolist() == [3.0, 1.0, 2.0] and torch.equal(tok["boxes"][2, 1], torch.tensor(boxes[2][1]))
    assert tok["boxes"].shape == (3, 30, 4) and float(tok["boxes"][1, 1:].abs().max()) == 0.0
