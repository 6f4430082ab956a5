"""ctypes binding of libgligen_hip.so (the C ABI declared in include/gligen_hip.h).

The product path has NO fallback: if the HIP library is missing or its ABI does not match, importing
this module's ``lib()`` raises.  Struct layouts mirror include/gligen_hip.h and are size-checked
against the library at load time.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GLIGEN_HIP_LIB: measurement hook -- the file name of an A/B build of the same library next to the product one
# (csrc/build.py --variant); never a path outside this package, never a fallback: a missing file raises like the product library
LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("GLIGEN_HIP_LIB", "") or "libgligen_hip.so"))
ABI_VERSION = 15

# enum gl_epilogue / gl_out_mode
EPI_BIAS, EPI_SILU, EPI_GEGLU, EPI_RES, EPI_GATE_RES, EPI_ROWBIAS = range(6)
OUT_F16_ROWMAJOR, OUT_F32_NCHW, OUT_F32_ROWMAJOR, OUT_F16_HILO = 0, 1, 2, 3

vp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float
fp = C.c_void_p  # float* passed as raw address


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", vp), ("lda", i32),
        ("a2", vp), ("lda2", i32), ("ksplit", i32),
        ("w", vp),
        ("bias", vp),
        ("M", i32), ("N", i32), ("K", i32),
        ("epi", i32),
        ("out_mode", i32),
        ("out", vp), ("ldc", i32),
        ("res", vp), ("ldres", i32),
        ("gate", vp),
        ("rowbias", vp), ("ld_rowbias", i32), ("rows_per_sample", i32),
        ("hw", i32),
        ("workspace", vp), ("workspace_bytes", i64),
        ("res_f32", i32),
        ("out2", vp), ("ldc2", i32),
        ("vt", vp), ("vt_col0", i32), ("vt_rows", i32), ("vt_d", i32), ("vt_ld", i32), ("vt_H", i32),
        ("ldw", i32), ("kwrap", i32),
        ("rowbias_f32", i32),
        ("vt_lo", vp),
    ]


class GnArgs(C.Structure):
    """gl_gn_args"""
    _fields_ = [
        ("x1", vp), ("C1", i32),
        ("x2", vp), ("C2", i32),
        ("x_f32", i32),
        ("B", i32), ("HW", i32),
        ("gamma", vp), ("beta", vp), ("eps", f32), ("silu", i32),
        ("out", vp), ("ldo", i32),
        ("out_lo", vp),
        ("raw", vp), ("ldraw", i32),
        ("partial", vp), ("nchunk", i32),
    ]


class ConvArgs(C.Structure):
    _fields_ = [
        ("inp", vp),
        ("B", i32), ("Hin", i32), ("Win", i32), ("Cin", i32),
        ("Hout", i32), ("Wout", i32),
        ("stride", i32),
        ("upsample2x", i32),
        ("g", GemmArgs),
        ("in_split", i32), ("w_split", i32),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", vp), ("q_bstride", i64), ("ldq", i32),
        ("k", vp), ("k_bstride", i64), ("ldk", i32),
        ("vt", vp), ("ldvt", i32),
        ("out", vp), ("o_bstride", i64), ("ldo", i32),
        ("B", i32), ("H", i32), ("d", i32), ("Nq", i32), ("Nk", i32),
        ("scale", f32),
        ("q_prescaled", i32),
        ("q_lo", vp), ("k_lo", vp), ("vt_lo", vp), ("out_lo", vp),
    ]


class UNetConfigC(C.Structure):
    """gl_unet_config"""
    _fields_ = [
        ("in_channels", i32), ("model_channels", i32), ("out_channels", i32), ("num_res_blocks", i32),
        ("n_levels", i32), ("channel_mult", i32 * 8),
        ("n_attn_res", i32), ("attention_resolutions", i32 * 8),
        ("num_heads", i32), ("context_dim", i32),
        ("pos_in_dim", i32), ("pos_out_dim", i32), ("fourier_freqs", i32),
        ("max_objs", i32),
        ("split_weights", i32),
    ]


class WeightInfo(C.Structure):
    """gl_weight_info"""
    _fields_ = [("name", C.c_char * 160), ("offset", i64), ("nbytes", i64), ("dtype", i32), ("ndim", i32), ("shape", i64 * 4)]


class VaeConfigC(C.Structure):
    """gl_vae_config"""
    _fields_ = [("ch", i32), ("ch_mult", i32 * 8), ("n_mult", i32), ("num_res_blocks", i32), ("z_channels", i32), ("out_ch", i32),
                ("embed_dim", i32), ("scale_factor", f32)]


class PlmsStepArgs(C.Structure):
    """gl_plms_step_args"""
    _fields_ = [
        ("x_eval", vp), ("x_base", vp), ("x_out", vp),
        ("e_out", vp),
        ("e_terms", vp * 4), ("coef", f32 * 4), ("n_terms", i32), ("div", f32),
        ("t", f32), ("reps", i32), ("guidance", f32), ("fuser_scale", f32), ("sd_conv", i32),
        ("sqrt_at", f32), ("s1m", f32), ("sqrt_aprev", f32), ("dir_coef", f32),
        ("use_graph", i32),
    ]


class RewardArgs(C.Structure):
    """gl_reward_args"""
    _fields_ = [("txt", vp), ("img_pred", vp), ("img_gt", vp), ("B", i32), ("D", i32),
                ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("w3", vp), ("b3", vp), ("w4", vp), ("b4", vp), ("w5", vp), ("b5", vp),
                ("sims_ti", vp), ("sims_ii", vp), ("aesthetic", vp), ("partial_reward", vp)]


class FFArgs(C.Structure):
    """gl_ff_args"""
    _fields_ = [("x", vp), ("ldx", i32), ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("res", vp), ("ldres", i32), ("res_f32", i32),
                ("gate", vp), ("out", vp), ("ldc", i32), ("out_mode", i32), ("M", i32), ("C", i32)]


# name -> (restype, argtypes); every symbol include/gligen_hip.h declares
PROTOTYPES = {
    "gl_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "gl_conv3x3": (i32, [C.POINTER(ConvArgs), vp]),
    "gl_attention": (i32, [C.POINTER(AttnArgs), vp]),
    "gl_transpose_v": (i32, [vp, i64, i32, vp, i32, i32, i32, i32, i32, vp]),
    "gl_groupnorm_stats": (i32, [vp, i32, vp, i32, i32, i32, fp, i32, vp]),
    "gl_groupnorm_apply": (i32, [vp, i32, vp, i32, i32, i32, fp, i32, fp, fp, f32, i32, vp, vp]),
    "gl_groupnorm": (i32, [vp, i32, vp, i32, i32, i32, fp, fp, f32, i32, vp, fp, i32, vp]),
    "gl_groupnorm_launches": (i32, [i32, i32]),
    "gl_groupnorm_ex": (i32, [C.POINTER(GnArgs), vp]),
    "gl_groupnorm_launches_ex": (i32, [i32, i32, i32]),
    "gl_sizeof_gn_args": (i32, []),
    "gl_layernorm": (i32, [vp, i32, i32, vp, i32, fp, fp, i32, i32, i32, i32, i32, f32, fp, vp, i32, i32, vp]),
    "gl_layernorm_stats": (i32, [fp, i32, i32, i32, f32, fp, vp]),
    "gl_split_f32": (i32, [fp, i32, i64, i32, vp, i32, vp]),
    "gl_timestep_embedding_f32": (i32, [fp, i32, i32, fp, vp]),
    "gl_posnet_input_f32": (i32, [fp, fp, fp, fp, fp, i32, i32, i32, fp, vp]),
    "gl_rela_pool_ln3": (i32, [fp, fp, fp, fp, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp, fp, fp, vp, vp]),
    "gl_rela_pool": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp, fp, fp, vp, vp]),
    "gl_rela_merge": (i32, [vp, i32, vp, fp, fp, fp, vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp, fp, fp, vp, vp]),
    "gl_posnet_input": (i32, [fp, fp, fp, fp, fp, i32, i32, i32, vp, vp]),
    "gl_timestep_embedding": (i32, [fp, i32, i32, vp, vp]),
    "gl_silu_f16": (i32, [vp, vp, i64, vp]),
    "gl_cfg_combine": (i32, [fp, f32, i64, fp, vp]),
    "gl_plms_update": (i32, [fp, fp, fp, fp, fp, f32, f32, f32, f32, f32, f32, f32, f32, f32, i64, fp, vp]),
    "gl_pack_latent": (i32, [fp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "gl_latent_affine_pack": (i32, [fp, fp, fp, f32, i32, i32, i32, i32, vp, vp]),
    "gl_softmax_rows": (i32, [vp, i32, i32, i32, f32, vp]),
    "gl_abi_version": (i32, []),
    "gl_sizeof_gemm_args": (i32, []),
    "gl_sizeof_conv_args": (i32, []),
    "gl_sizeof_attn_args": (i32, []),
    "gl_init": (i32, []),
    "gl_create": (i32, [C.POINTER(UNetConfigC), C.POINTER(vp)]),
    "gl_destroy": (i32, [vp]),
    "gl_num_weights": (i32, [vp]),
    "gl_weight_at": (i32, [vp, i32, C.POINTER(WeightInfo)]),
    "gl_weights_bytes": (i64, [vp]),
    "gl_load_weights": (i32, [vp, vp, i64, i32, vp]),
    "gl_set_conditioning": (i32, [vp, fp, fp, fp, fp, fp, i32, i32, i32, i32, vp]),
    "gl_unet_forward": (i32, [vp, fp, fp, f32, i32, f32, i32, fp, i32, vp]),
    "gl_plms_step": (i32, [vp, C.POINTER(PlmsStepArgs), vp]),
    "gl_pool_bytes": (i64, [vp]),
    "gl_num_launches": (i32, [vp]),
    "gl_sizeof_unet_config": (i32, []),
    "gl_sizeof_weight_info": (i32, []),
    "gl_sizeof_plms_step_args": (i32, []),
    "gl_reward_score": (i32, [C.POINTER(RewardArgs), vp]),
    "gl_sizeof_reward_args": (i32, []),
    "gl_ff_fused": (i32, [C.POINTER(FFArgs), vp]),
    "gl_ff_fused_supported": (i32, [i32]),
    "gl_ff_fused_applicable": (i32, [i32, i32]),
    "gl_sizeof_ff_args": (i32, []),
    "gl_image_to_u8": (i32, [vp, i32, i32, i32, vp, vp]),
    "gl_resample_h_u8": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, vp, vp]),
    "gl_resample_v_norm": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "gl_set_option": (i32, [i32, i32]),
    "gl_set_handle_option": (i32, [vp, i32, i32]),
    "gl_clear_handle_options": (i32, [vp]),
    "gl_vae_set_option": (i32, [vp, i32, i32]),
    "gl_debug_read": (i32, [i32, vp, i64]),
    "gl_vae_create": (i32, [C.POINTER(VaeConfigC), C.POINTER(vp)]),
    "gl_vae_destroy": (i32, [vp]),
    "gl_vae_num_weights": (i32, [vp]),
    "gl_vae_weight_at": (i32, [vp, i32, C.POINTER(WeightInfo)]),
    "gl_vae_weights_bytes": (i64, [vp]),
    "gl_vae_load_weights": (i32, [vp, vp, i64, vp]),
    "gl_vae_decode": (i32, [vp, fp, i32, i32, fp, i32, vp]),
    "gl_vae_num_launches": (i32, [vp]),
    "gl_vae_pool_bytes": (i64, [vp]),
    "gl_sizeof_vae_config": (i32, []),
    "gl_clip_patchify": (i32, [fp, i32, i32, i32, i32, vp, vp]),
    "gl_clip_assemble": (i32, [vp, i32, fp, fp, i32, i32, i32, fp, fp, f32, fp, vp]),
    "gl_clip_embed_tokens": (i32, [vp, fp, fp, i32, i32, i32, i32, fp, vp]),
    "gl_clip_gather_rows": (i32, [fp, i32, vp, i32, i32, fp, vp]),
    "gl_attention_small": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp, i32, vp]),
}

_lib = None
_inited = set()


class HipLibraryError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Loads (once) and returns the HIP library; raises HipLibraryError if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -m layoutllm_t2i_amd.csrc.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the denoising path.")
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if l.gl_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: lib {l.gl_abi_version()} vs host {ABI_VERSION}")
    for cls, fn in ((GemmArgs, l.gl_sizeof_gemm_args), (ConvArgs, l.gl_sizeof_conv_args), (AttnArgs, l.gl_sizeof_attn_args),
                    (UNetConfigC, l.gl_sizeof_unet_config), (WeightInfo, l.gl_sizeof_weight_info),
                    (PlmsStepArgs, l.gl_sizeof_plms_step_args), (RewardArgs, l.gl_sizeof_reward_args), (FFArgs, l.gl_sizeof_ff_args),
                    (VaeConfigC, l.gl_sizeof_vae_config), (GnArgs, l.gl_sizeof_gn_args)):
        if C.sizeof(cls) != fn():
            raise HipLibraryError(f"struct size mismatch for {cls.__name__}: host {C.sizeof(cls)} vs lib {fn()}")
    _lib = l
    return l


def init_device(index=None) -> None:
    """One-time per-device setup (needs a GPU): raises the dynamic-LDS limits of the tiled kernels on the CURRENT
    device, so it is tracked per device index."""
    import torch
    if index is None:
        index = torch.cuda.current_device()
    if index not in _inited:
        with torch.cuda.device(index):
            check(lib().gl_init(), "gl_init")
        _inited.add(index)


def unet_config_c(cfg) -> UNetConfigC:
    """arch.UNetConfig -> gl_unet_config"""
    c = UNetConfigC()
    c.in_channels, c.model_channels, c.out_channels, c.num_res_blocks = cfg.in_channels, cfg.model_channels, cfg.out_channels, cfg.num_res_blocks
    if len(cfg.channel_mult) > 8 or len(cfg.attention_resolutions) > 8:
        raise ValueError("at most 8 levels / attention resolutions")
    c.n_levels = len(cfg.channel_mult)
    for i, m in enumerate(cfg.channel_mult):
        c.channel_mult[i] = int(m)
    c.n_attn_res = len(cfg.attention_resolutions)
    for i, m in enumerate(cfg.attention_resolutions):
        c.attention_resolutions[i] = int(m)
    c.num_heads, c.context_dim = cfg.num_heads, cfg.context_dim
    c.pos_in_dim, c.pos_out_dim, c.fourier_freqs, c.max_objs = cfg.pos_in_dim, cfg.pos_out_dim, cfg.fourier_freqs, cfg.max_objs
    c.split_weights = int(bool(getattr(cfg, "split_weights", False)))
    return c


def create_engine(cfg) -> int:
    """gl_create: returns the opaque handle (an int address).  Needs no GPU."""
    h = vp()
    cc = unet_config_c(cfg)
    check(lib().gl_create(C.byref(cc), C.byref(h)), "gl_create")
    return h.value


def create_vae(cfg) -> int:
    """gl_vae_create from an arch.VAEConfig: returns the opaque handle.  Needs no GPU."""
    cc = VaeConfigC()
    cc.ch, cc.n_mult, cc.num_res_blocks, cc.z_channels, cc.out_ch = cfg.ch, len(cfg.ch_mult), cfg.num_res_blocks, cfg.z_channels, cfg.out_ch
    for i, m in enumerate(cfg.ch_mult):
        cc.ch_mult[i] = int(m)
    cc.embed_dim, cc.scale_factor = cfg.embed_dim, float(cfg.scale_factor)
    h = vp()
    check(lib().gl_vae_create(C.byref(cc), C.byref(h)), "gl_vae_create")
    return h.value


def vae_weight_table(handle: int):
    l = lib()
    out = []
    info = WeightInfo()
    for i in range(l.gl_vae_num_weights(handle)):
        check(l.gl_vae_weight_at(handle, i, C.byref(info)), "gl_vae_weight_at")
        out.append((info.name.decode(), int(info.offset), int(info.nbytes), int(info.dtype), tuple(int(info.shape[k]) for k in range(info.ndim))))
    return out, int(l.gl_vae_weights_bytes(handle))


def weight_table(handle: int):
    """[(name, offset, nbytes, dtype 0 = fp16 / 1 = fp32, shape)] of the flat packed-weight buffer, + total bytes"""
    l = lib()
    out = []
    info = WeightInfo()
    for i in range(l.gl_num_weights(handle)):
        check(l.gl_weight_at(handle, i, C.byref(info)), "gl_weight_at")
        out.append((info.name.decode(), int(info.offset), int(info.nbytes), int(info.dtype), tuple(int(info.shape[k]) for k in range(info.ndim))))
    return out, int(l.gl_weights_bytes(handle))


def check(code: int, what: str) -> None:
    if code != 0:
        raise HipLibraryError(f"{what} failed with code {code}" + (" (bad argument)" if code == -1 else ""))
