// Small kernels of the denoising path: grounding-token input (Fourier box embedding + null blend),
// sinusoidal timestep embedding, SiLU, the PLMS/CFG latent update, latent packing, ABI introspection.
#include "common.h"
#include "gligen_hip.h"
#include "opts.h"

namespace {

// text_grounding_net.py:30-41 + util.py:12-26.  One block per (b, i) row; out row = [in_dim | 8*num_freqs].
// Fourier order: for each frequency f_j = 100^(j/num_freqs): sin(f_j * xyxy) (4) then cos(f_j * xyxy) (4).
// OT = half_t (default) or float (strict mode: the row is split into [hi | lo] by gl_split_f32 afterwards)
template <typename OT>
__global__ __launch_bounds__(256) void posnet_input_kernel(const float* __restrict__ boxes, const float* __restrict__ masks,
                                                           const float* __restrict__ emb, const float* __restrict__ null_pos,
                                                           const float* __restrict__ null_xyxy, int in_dim, int num_freqs,
                                                           OT* __restrict__ out) {
    const int row = blockIdx.x;
    const float m = masks[row];
    const int pos_dim = num_freqs * 8;
    OT* o = out + (size_t)row * (in_dim + pos_dim);
    for (int c = threadIdx.x; c < in_dim; c += 256) {
        const float v = emb[(size_t)row * in_dim + c] * m + (1.0f - m) * null_pos[c];
        o[c] = (OT)v;
    }
    for (int c = threadIdx.x; c < pos_dim; c += 256) {
        const int j = c / 8;
        const int r = c - j * 8;
        const int coord = r & 3;
        const float freq = powf(100.0f, (float)j / (float)num_freqs);
        const float arg = freq * boxes[(size_t)row * 4 + coord];
        const float e = (r < 4) ? sinf(arg) : cosf(arg);
        o[in_dim + c] = (OT)(e * m + (1.0f - m) * null_xyxy[c]);
    }
}

// util.py:161-181: [cos(t*w) | sin(t*w)], w_k = exp(-ln(10000) * k / half)
template <typename OT>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int dim, OT* __restrict__ out) {
    const int b = blockIdx.x;
    const int half_dim = dim / 2;
    for (int k = threadIdx.x; k < half_dim; k += blockDim.x) {
        const float w = expf(-9.210340371976184f * (float)k / (float)half_dim);
        const float a = t[b] * w;
        out[(size_t)b * dim + k] = (OT)cosf(a);
        out[(size_t)b * dim + half_dim + k] = (OT)sinf(a);
    }
    if ((dim & 1) && threadIdx.x == 0) out[(size_t)b * dim + dim - 1] = (OT)0.0f;
}

// fp32 rows -> fp16 [hi | lo] rows (8 channels per thread)
__global__ void split_f32_kernel(const float* __restrict__ x, int ldx, size_t rows, int nvec, half_t* __restrict__ y, int ldy) {
    const size_t total = rows * (size_t)nvec;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (size_t)nvec;
        const int c = (int)(i - r * (size_t)nvec) * 8;
        const float4 a = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float4 b = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        half8_t hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (half_t)v[j]; lo[j] = (half_t)(v[j] - (float)hi[j]); }
        st16(y + r * ldy + c, *reinterpret_cast<uint4*>(&hi));
        st16(y + r * ldy + (size_t)nvec * 8 + c, *reinterpret_cast<uint4*>(&lo));
    }
}

__global__ void silu_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, size_t nvec) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        uint4 raw = ld16(x + i * 8);
        const half8_t v = *reinterpret_cast<half8_t*>(&raw);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)silu_f((float)v[j]);
        st16(y + i * 8, *reinterpret_cast<uint4*>(&o));
    }
}

// The sampler arithmetic follows the reference's operation order with contraction disabled so that,
// given identical eps, x_prev is bit-identical to torch fp32 (plms.py:123,126-161).
#pragma clang fp contract(off)
__global__ void cfg_combine_kernel(const float* __restrict__ eps2b, float guidance, size_t n, float* __restrict__ e) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float ec = eps2b[i];
        const float eu = eps2b[n + i];
        e[i] = eu + guidance * (ec - eu);
    }
}

// No __restrict__: gl_plms_step updates in place (x == x_prev) and an eps term may alias the output; every access is
// same-index and element-wise, which is well defined only without the no-alias promise.
__global__ void plms_update_kernel(const float* x, const float* e, const float* e1, const float* e2, const float* e3, float c0, float c1,
                                   float c2, float c3, float div, float sqrt_at, float s1m, float sqrt_aprev,
                                   float dir_coef, size_t n, float* x_prev) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // e' : (c0*e + c1*e1 + c2*e2 + c3*e3) / div, left to right as plms.py:146-159 writes it
        float ep = c0 * e[i];
        if (e1) ep = ep + c1 * e1[i];
        if (e2) ep = ep + c2 * e2[i];
        if (e3) ep = ep + c3 * e3[i];
        ep = ep / div;
        const float pred_x0 = (x[i] - s1m * ep) / sqrt_at;
        const float dir_xt = dir_coef * ep;
        x_prev[i] = sqrt_aprev * pred_x0 + dir_xt;
    }
}
#pragma clang fp contract(fast)

// x fp32 [B, C, hw] -> fp16 [reps*B, hw, Cpad].  split: channels [0, C) = hi = fp16(x), [C, 2C) = lo = fp16(x - hi), [2C, 3C) = hi again -- against
// first-conv weights packed [Whi | Whi | Wlo] (weights.py pack_first_conv) the one conv launch computes xhi.Whi + xlo.Whi + xhi.Wlo in the channel
// padding it carries anyway (4 of 64 channels used)
__global__ void pack_latent_kernel(const float* __restrict__ x, int B, int C, int hw, int Cpad, int reps, int split,
                                   half_t* __restrict__ out) {
    const size_t total = (size_t)reps * B * hw * Cpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const size_t t = i / Cpad;
        const int p = (int)(t % hw);
        const int rb = (int)(t / hw);
        const int b = rb % B;
        half_t o = (half_t)0.0f;
        if (c < C) {
            o = (half_t)x[((size_t)b * C + c) * hw + p];
        } else if (split && c < 3 * C) {
            const int cc = c < 2 * C ? c - C : c - 2 * C;
            const float v = x[((size_t)b * C + cc) * hw + p];
            const half_t hi = (half_t)v;
            o = c < 2 * C ? (half_t)(v - (float)hi) : hi;
        }
        out[i] = o;
    }
}

// Row softmax in place over fp16 [rows, n] (row stride ld): the single-head d = C mid-block attention of
// the VAE decoder (model.py:180-186) runs as two GEMMs around this kernel because its head dim (512)
// exceeds the flash kernel's register budget; it runs once per image, not per step.
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* __restrict__ x, int n, int ld, float scale) {
    __shared__ float red[4];
    half_t* row = x + (size_t)blockIdx.x * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int i = threadIdx.x * 8; i < n; i += 256 * 8) {
        uint4 raw = ld16(row + i);
        const half8_t v = *reinterpret_cast<half8_t*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)v[j]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
    __syncthreads();
    float sum = 0.0f;
    for (int i = threadIdx.x * 8; i < n; i += 256 * 8) {
        uint4 raw = ld16(row + i);
        const half8_t v = *reinterpret_cast<half8_t*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += __expf((float)v[j] * scale - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = threadIdx.x * 8; i < n; i += 256 * 8) {
        uint4 raw = ld16(row + i);
        const half8_t v = *reinterpret_cast<half8_t*>(&raw);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)(__expf((float)v[j] * scale - mx) * inv);
        st16(row + i, *reinterpret_cast<uint4*>(&o));
    }
}

// z fp32 [B, C, hw] -> fp16 [B, hw, Cpad]: out[:, :, co] = bias[co] + sum_ci w[co, ci] * (z[:, ci, :] * pre),
// zero for co >= C.  AutoencoderKL.decode's 1/scale_factor and 1x1 post_quant_conv (autoencoder.py:41-42).
__global__ void latent_affine_pack_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                          const float* __restrict__ bias, float pre, int B, int C, int hw, int Cpad,
                                          half_t* __restrict__ out) {
    const size_t total = (size_t)B * hw * Cpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cpad);
        const size_t t = i / Cpad;
        const int p = (int)(t % hw);
        const int b = (int)(t / hw);
        float v = 0.0f;
        if (co < C) {
            v = bias[co];
            for (int ci = 0; ci < C; ++ci) v += w[co * C + ci] * (z[((size_t)b * C + ci) * hw + p] * pre);
        }
        out[i] = (half_t)v;
    }
}

int ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b == 0 ? 1 : b));
}

}  // namespace

extern "C" int gl_posnet_input(const float* boxes, const float* masks, const float* emb, const float* null_pos,
                               const float* null_xyxy, int32_t rows, int32_t in_dim, int32_t num_freqs, void* out,
                               void* stream) {
    if (!boxes || !masks || !emb || !null_pos || !null_xyxy || !out || rows <= 0) return GL_ERR_BAD_ARG;
    posnet_input_kernel<half_t><<<dim3(rows), dim3(256), 0, (hipStream_t)stream>>>(boxes, masks, emb, null_pos, null_xyxy, in_dim,
                                                                                   num_freqs, reinterpret_cast<half_t*>(out));
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_posnet_input_f32(const float* boxes, const float* masks, const float* emb, const float* null_pos,
                                   const float* null_xyxy, int32_t rows, int32_t in_dim, int32_t num_freqs, float* out, void* stream) {
    if (!boxes || !masks || !emb || !null_pos || !null_xyxy || !out || rows <= 0) return GL_ERR_BAD_ARG;
    posnet_input_kernel<float><<<dim3(rows), dim3(256), 0, (hipStream_t)stream>>>(boxes, masks, emb, null_pos, null_xyxy, in_dim, num_freqs, out);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_timestep_embedding_f32(const float* t, int32_t B, int32_t dim, float* out, void* stream) {
    if (!t || !out || B <= 0 || dim <= 0) return GL_ERR_BAD_ARG;
    timestep_embedding_kernel<float><<<dim3(B), dim3(256), 0, (hipStream_t)stream>>>(t, dim, out);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_split_f32(const float* x, int32_t ldx, int64_t rows, int32_t C, void* y, int32_t ldy, void* stream) {
    if (!x || !y || rows <= 0 || C <= 0 || (C % 8) || (ldx % 4) || (ldy % 8) || ldy < 2 * C || ldx < C) return GL_ERR_BAD_ARG;
    const size_t total = (size_t)rows * (C / 8);
    split_f32_kernel<<<dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream>>>(x, ldx, (size_t)rows, C / 8, reinterpret_cast<half_t*>(y), ldy);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_timestep_embedding(const float* t, int32_t B, int32_t dim, void* out, void* stream) {
    if (!t || !out || B <= 0 || dim <= 0) return GL_ERR_BAD_ARG;
    timestep_embedding_kernel<half_t><<<dim3(B), dim3(256), 0, (hipStream_t)stream>>>(t, dim, reinterpret_cast<half_t*>(out));
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_silu_f16(const void* x, void* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0 || (n % 8)) return GL_ERR_BAD_ARG;
    silu_kernel<<<dim3(ew_blocks((size_t)n / 8)), dim3(256), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const half_t*>(x), reinterpret_cast<half_t*>(y), (size_t)n / 8);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_cfg_combine(const float* eps2b, float guidance, int64_t n, float* e_out, void* stream) {
    if (!eps2b || !e_out || n <= 0) return GL_ERR_BAD_ARG;
    cfg_combine_kernel<<<dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream>>>(eps2b, guidance, (size_t)n, e_out);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_plms_update(const float* x, const float* e, const float* e1, const float* e2, const float* e3, float c0,
                              float c1, float c2, float c3, float div, float sqrt_at, float s1m, float sqrt_aprev,
                              float dir_coef, int64_t n, float* x_prev, void* stream) {
    if (!x || !e || !x_prev || n <= 0) return GL_ERR_BAD_ARG;
    plms_update_kernel<<<dim3(ew_blocks((size_t)n)), dim3(256), 0, (hipStream_t)stream>>>(
        x, e, e1, e2, e3, c0, c1, c2, c3, div, sqrt_at, s1m, sqrt_aprev, dir_coef, (size_t)n, x_prev);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_pack_latent(const float* x, int32_t B, int32_t C, int32_t hw, int32_t Cpad, int32_t reps, int32_t split, void* out,
                              void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || hw <= 0 || Cpad < (split ? 3 * C : C) || reps <= 0) return GL_ERR_BAD_ARG;
    pack_latent_kernel<<<dim3(ew_blocks((size_t)reps * B * hw * Cpad)), dim3(256), 0, (hipStream_t)stream>>>(
        x, B, C, hw, Cpad, reps, split, reinterpret_cast<half_t*>(out));
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_softmax_rows(void* x, int32_t rows, int32_t n, int32_t ld, float scale, void* stream) {
    if (!x || rows <= 0 || n <= 0 || (n % 8) || (ld % 8)) return GL_ERR_BAD_ARG;
    softmax_rows_kernel<<<dim3(rows), dim3(256), 0, (hipStream_t)stream>>>(reinterpret_cast<half_t*>(x), n, ld, scale);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_latent_affine_pack(const float* z, const float* w, const float* bias, float pre, int32_t B, int32_t C,
                                     int32_t hw, int32_t Cpad, void* out, void* stream) {
    if (!z || !w || !bias || !out || B <= 0 || C <= 0 || hw <= 0 || Cpad < C) return GL_ERR_BAD_ARG;
    latent_affine_pack_kernel<<<dim3(ew_blocks((size_t)B * hw * Cpad)), dim3(256), 0, (hipStream_t)stream>>>(
        z, w, bias, pre, B, C, hw, Cpad, reinterpret_cast<half_t*>(out));
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_init_gemm(void);
extern "C" int gl_init_ff(void);
extern "C" int gl_init_attn(void);
// ---- option table (opts.h): process defaults, per-thread effective table of the running handle
static gl_opts make_default_opts() {
    gl_opts o{};
    o.v[2] = 0;    o.v[3] = 0;    o.v[4] = 400;  o.v[5] = 300;  o.v[GL_OPT_SPLITK_TILES_CONV] = 450;
    o.v[6] = 16;   o.v[7] = 300;  o.v[8] = 1;    o.v[10] = -1;  o.v[13] = 3;
    o.v[16] = 16;  o.v[17] = 1;   o.v[20] = 0;   o.v[21] = 1;   o.v[23] = 1;
    o.v[24] = 64;  o.v[25] = 1;   o.v[27] = 1;   o.v[29] = 1;   o.v[30] = 1;
    o.v[31] = 200; o.v[32] = 0;   o.v[33] = 0;   o.v[34] = 11;  o.v[35] = 5;
    o.v[37] = 1;
    o.v[41] = 1;
    o.v[42] = 1;
    o.v[43] = 1;
    o.v[38] = 1;
    o.v[44] = 1;
    o.v[45] = 1024;
    o.v[46] = 11;
    o.v[47] = 100;
    o.v[50] = 0;       // strict mode (handles created with split_weights)
    o.v[51] = 1;       // ... with the third pass x.Wlo
    o.v[52] = 1;       // three-pass products on the dedicated three-pass loop of the 8-wave kernel (gemm8.hip S3; 0 = K-walk)
    return o;
}
gl_opts g_gl_opts = make_default_opts();
thread_local const gl_opts* tl_gl_opts = nullptr;
int g_gl_option_epoch = 0;

bool gl_opts_store(gl_opts& t, int key, int value) {
    switch (key) {
        case 2: case 3: case 4: case 6: case 7: case 8: case 10: case 13: case 17: case 20: case 21: case 23: case 24: case 25:
        case 27: case 29: case 30: case 31: case 32: case 33: case 35: case 37: case 38: case 41: case 42: case 43: case 44: case 45: case 46: case 47: case 50: case 51: case 52: case 53:
            t.v[key] = value;
            return true;
        case 5:                                  // < 0: the built-in thresholds (plain GEMM 300 tiles, conv 450)
            t.v[5] = value < 0 ? 300 : value;
            t.v[GL_OPT_SPLITK_TILES_CONV] = value < 0 ? 450 : value;
            return true;
        case 16: t.v[16] = value > 0 ? value : 16; return true;
        case 34: t.v[34] = value < 1 ? 1 : value; return true;
        default: return false;
    }
}

extern "C" int gl_set_option(int key, int value) {
    if (key < 0 || key >= GL_OPT_MAX || !gl_opts_store(g_gl_opts, key, value)) return GL_ERR_BAD_ARG;
    ++g_gl_option_epoch;
    return 0;
}

extern "C" int gl_abi_version(void) { return GL_ABI_VERSION; }
extern "C" int gl_sizeof_gemm_args(void) { return (int)sizeof(gl_gemm_args); }
extern "C" int gl_sizeof_conv_args(void) { return (int)sizeof(gl_conv_args); }
extern "C" int gl_sizeof_attn_args(void) { return (int)sizeof(gl_attn_args); }
extern "C" int gl_sizeof_gn_args(void) { return (int)sizeof(gl_gn_args); }
extern "C" int gl_init(void) {
    int e = gl_init_gemm();
    if (!e) e = gl_init_ff();
    return e ? e : gl_init_attn();
}
