"""Reward-stage image preprocessing (models/policy.py:108-111): the oracle restatement pinned to Pillow and to the
installed transformers PIL-backend CLIP processor (CPU), and the HIP kernels against both (GPU): bit-exact uint8 resampling,
bit-exact fp32 pixel_values."""
import numpy as np
import pytest
import torch
from PIL import Image

from layoutllm_t2i_amd import preprocess as pp
from oracle import clip_preprocess_ref as ref

SIZES = [(512, 512), (480, 640), (333, 500), (100, 150), (224, 300), (64, 64), (427, 640), (640, 427), (225, 224), (1024, 768)]


def _img(h, w, seed, smooth=False):
    rng = np.random.default_rng(seed)
    if not smooth:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    chans = [127.5 + 127.5 * np.sin(xx / (7.0 + c) + yy / (11.0 - c) + c) for c in range(3)]
    return np.clip(np.stack(chans, -1) + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w", SIZES)
def test_oracle_resize_equals_pillow_bit_exact(h, w):
    oh, ow = ref.resized_shape(h, w, 224)
    for smooth in (False, True):
        img = _img(h, w, h * 7 + w, smooth)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC))
        assert np.array_equal(ref.pil_resize_bicubic(img, oh, ow), want)


def test_oracle_pipeline_equals_the_hf_pil_processor_bit_exact():
    hf = pytest.importorskip("transformers")
    try:
        from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil as Proc
    except Exception:                                   # older layouts: the default class is the PIL one
        Proc = hf.CLIPImageProcessor
    proc = Proc()
    for (h, w) in [(512, 512), (480, 640), (640, 427), (100, 150)]:
        img = _img(h, w, 3 * h + w, True)
        want = proc(images=[Image.fromarray(img)], return_tensors="np")["pixel_values"][0]
        got, _ = ref.clip_feature_extractor(img)
        assert got.dtype == np.float32 and np.array_equal(got, want), float(np.abs(got - want).max())


@pytest.mark.parametrize("n_in,n_out", [(512, 224), (640, 298), (150, 336), (64, 224), (300, 300), (1024, 298), (225, 224)])
def test_product_tables_equal_the_oracle(n_in, n_out):
    b, k = pp.pil_bicubic_tables(n_in, n_out)
    assert b.dtype == np.int32 and k.dtype == np.int32
    if n_in == n_out:
        assert np.array_equal(b[:, 0], np.arange(n_out)) and (b[:, 1] == 1).all() and (k == 1 << 22).all()
        return
    rb, rk, ksize = ref.precompute_coeffs(n_in, n_out)
    assert k.shape == (n_out, ksize) and np.array_equal(b, rb) and np.array_equal(k, ref.normalize_coeffs_8bpc(rk))
    assert (np.abs(k.sum(1) - (1 << 22)) <= k.shape[1]).all()         # rows sum to one up to per-tap rounding


def test_resized_shape_rules():
    assert pp.resized_shape(512, 512, 224) == (224, 224)
    assert pp.resized_shape(480, 640, 224) == (224, 298)
    assert pp.resized_shape(640, 427, 224) == (335, 224)
    assert pp.resized_shape(224, 300, 224) == (224, 300)
    for (h, w) in SIZES:
        assert pp.resized_shape(h, w, 224) == ref.resized_shape(h, w, 224)


def test_decoded_to_u8_reference_arithmetic():
    x = torch.tensor([-2.0, -1.0, -0.999, 0.0, 0.00392, 0.5, 0.9999, 1.0, 3.0]).view(1, 1, 3, 3).repeat(1, 3, 1, 1)
    u = ref.decoded_to_u8(x)
    assert u.shape == (1, 3, 3, 3) and u[0, 0, 0, 0] == 0 and u[0, 2, 2, 0] == 255 and u[0, 1, 0, 0] == 127


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("h,w", SIZES)
def test_hip_resample_equals_pillow_bit_exact(h, w):
    P = pp.ClipImagePreprocessor()
    imgs = np.stack([_img(h, w, 5 * h + w + i, smooth=bool(i & 1)) for i in range(3)])
    px, u8 = P(torch.from_numpy(imgs).to("cuda:0"), return_u8=True)
    oh, ow = pp.resized_shape(h, w, 224)
    top, left = (oh - 224) // 2, (ow - 224) // 2
    for i in range(3):
        want = np.asarray(Image.fromarray(imgs[i]).resize((ow, oh), resample=Image.BICUBIC))[top:top + 224, left:left + 224]
        assert np.array_equal(u8[i].cpu().numpy(), want), (h, w, i)
        ref_px, ref_u8 = ref.clip_feature_extractor(imgs[i])
        assert np.array_equal(ref_u8, want)
        assert np.array_equal(px[i].cpu().numpy(), ref_px), float(np.abs(px[i].cpu().numpy() - ref_px).max())


@pytest.mark.gpu
def test_hip_decoded_to_u8_and_whole_path_bit_exact():
    P = pp.ClipImagePreprocessor()
    g = torch.Generator().manual_seed(3)
    dec = torch.randn(4, 3, 256, 256, generator=g) * 0.7
    dec[0, :, :4, :4] = torch.tensor([-1.0, 1.0, 0.0, 0.5]).view(1, 1, 4)          # exact boundaries
    dec[1, 0, 0, :8] = torch.tensor([-1.0000001, 1.0000001, 0.003921568, 0.003921569, 0.99607843, 0.9960785, -0.0, 1e-9])
    u = P.to_u8(dec.to("cuda:0"))
    want = ref.decoded_to_u8(dec)
    assert np.array_equal(u.cpu().numpy(), want)
    px = P.from_decoded(dec.to("cuda:0")).cpu().numpy()
    for i in range(4):
        assert np.array_equal(px[i], ref.clip_feature_extractor(want[i])[0])


@pytest.mark.gpu
def test_hip_mixed_sizes_keep_their_order_and_match_the_hf_processor():
    hf = pytest.importorskip("transformers")
    try:
        from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil as Proc
    except Exception:
        Proc = hf.CLIPImageProcessor
    shapes = [(480, 640), (512, 512), (480, 640), (333, 500), (512, 512)]
    pil = [Image.fromarray(_img(h, w, 17 * i + h, True)) for i, (h, w) in enumerate(shapes)]
    got = pp.ClipImagePreprocessor().from_pil(pil).cpu().numpy()
    want = Proc()(images=pil, return_tensors="np")["pixel_values"]
    assert got.shape == want.shape == (5, 3, 224, 224) and np.array_equal(got, want)
