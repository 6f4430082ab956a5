// RelationCrossAttention (LayoutLLM-T2I's addition, attention.py:315-359) in closed form.
//
// The reference materialises three B x 30 x h x w x C tensors (157 MB fp32 per sample each at 64x64)
// and runs python loops with 4 host syncs per call.  Algebraically
//     out[p] = hid[p] + (1/max_objs) * sum_i 1[p in rect_i] * f_i ,      hid = LayerNorm3(x)
// where f_i is the (attention + GEGLU-refined) mean of hid over box i.  Two small HBM-bound kernels:
//   rela_pool  : segmented mean of hid over each valid box rectangle  (one block per (sample, box))
//   rela_merge : y = ((hid + sum/max_objs) + x) / 2, the /2 being BasicTransformerBlock's
//                (rela_fuse(x) + x) / 2 (attention.py:398)
// Rectangles, the valid-box count with the `break` rule and the NaN poison flag are computed once
// per image on the host in fp32 (int truncation hazards), see host.py:box_rects.
#include "common.h"
#include "gligen_hip.h"

namespace {

constexpr int RELA_MAX_C = 2048;
constexpr int RP_NT = 1024;          // rela_pool block size

// grid (max_objs, B), block RP_NT: thread -> (pixel plane, 8-channel vector).  Only the used slots of the conditional samples
// have work (32 blocks at B = 4 with 8 boxes), each over a rectangle of up to H*W pixels: 1024 threads per block keep 4x the
// loads in flight per box (21.6 -> ~9 us at 64x64x320)
// LN3 (round 4): the source is the fp32 stream x itself with the per-row (mean, rstd) of LayerNorm3 (gl_layernorm_stats); the pooled mean of
// hid = LN3(x) over a rectangle is then gamma * mean_rect((x - mean_r) * rstd_r) + beta, evaluated in fp32 -- hid is never rounded to fp16
// (nor stored at all).
template <bool LN3>
__global__ __launch_bounds__(RP_NT) void rela_pool_kernel(const half_t* __restrict__ hid, const float* __restrict__ x32,
                                                        const float* __restrict__ ln3_stats, const float* __restrict__ ln3_g,
                                                        const float* __restrict__ ln3_b, int H, int W, int C,
                                                        const int* __restrict__ rects, const int* __restrict__ nvalid,
                                                        const int* __restrict__ poison, int max_objs, int slots,
                                                        half_t* __restrict__ feat, const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, half_t* __restrict__ ln_out) {
    __shared__ float lacc[8 * RP_NT];    // [plane][C] partial sums, nplanes * C <= 8 * RP_NT
    __shared__ float red[RP_NT / 64];
    const int i = blockIdx.x;
    const int b = blockIdx.y;
    const int nvec = C / 8;
    half_t* frow = feat + ((size_t)b * slots + i) * C;      // rows: `slots` per sample; rects: max_objs per sample
    // mode: 0 = pooled mean, 1 = unused slot (zero row), 2 = NaN row (empty python slice: torch.mean of nothing,
    // attention.py:343, or a poisoned sample)
    int mode = 0;
    int top = 0, left = 0, rw = 1, npix = 1;
    if (i >= nvalid[b]) {
        mode = 1;
    } else {
        const int* r = rects + ((size_t)b * max_objs + i) * 4;
        top = r[0]; left = r[2];
        rw = r[3] - r[2];
        npix = (r[1] - r[0]) * rw;
        if (npix <= 0 || poison[b] != 0) mode = 2;
    }
    const int vlanes = min(nvec, RP_NT);
    const int nplanes = RP_NT / vlanes;
    if (mode == 0) {
        const int plane = threadIdx.x / vlanes;
        const int v0 = threadIdx.x - plane * vlanes;
        if (plane < nplanes) {
            for (int vec = v0; vec < nvec; vec += vlanes) {
                float s[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] = 0.0f;
                int q = plane;
                if constexpr (LN3) {
                    // 2 pixels x (two 16-byte loads + the row's statistics) in flight per thread
                    for (; q + nplanes < npix; q += 2 * nplanes) {
                        float4 xa[2], xb[2];
                        float2 ms[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int qq = q + u * nplanes;
                            const size_t row = ((size_t)b * H + top + qq / rw) * W + left + qq % rw;
                            const float* xr = x32 + row * C + vec * 8;
                            xa[u] = *reinterpret_cast<const float4*>(xr);
                            xb[u] = *reinterpret_cast<const float4*>(xr + 4);
                            ms[u] = *reinterpret_cast<const float2*>(ln3_stats + row * 2);
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const float xv[8] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w, xb[u].x, xb[u].y, xb[u].z, xb[u].w};
#pragma unroll
                            for (int j = 0; j < 8; ++j) s[j] += (xv[j] - ms[u].x) * ms[u].y;
                        }
                    }
                    for (; q < npix; q += nplanes) {
                        const size_t row = ((size_t)b * H + top + q / rw) * W + left + q % rw;
                        const float* xr = x32 + row * C + vec * 8;
                        const float4 a4 = *reinterpret_cast<const float4*>(xr), b4 = *reinterpret_cast<const float4*>(xr + 4);
                        const float2 m2 = *reinterpret_cast<const float2*>(ln3_stats + row * 2);
                        const float xv[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) s[j] += (xv[j] - m2.x) * m2.y;
                    }
                } else {
                // 4 independent 16-byte loads in flight per thread (rectangles can span 2k+ pixels)
                for (; q + 3 * nplanes < npix; q += 4 * nplanes) {
                    uint4 raw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int qq = q + u * nplanes;
                        const int py = top + qq / rw;
                        const int px = left + qq % rw;
                        raw[u] = ld16(hid + (((size_t)b * H + py) * W + px) * C + vec * 8);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const half8_t hv = *reinterpret_cast<half8_t*>(&raw[u]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) s[j] += (float)hv[j];
                    }
                }
                for (; q < npix; q += nplanes) {
                    const int py = top + q / rw;
                    const int px = left + q % rw;
                    uint4 raw = ld16(hid + (((size_t)b * H + py) * W + px) * C + vec * 8);
                    const half8_t hv = *reinterpret_cast<half8_t*>(&raw);
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[j] += (float)hv[j];
                }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) lacc[plane * C + vec * 8 + j] = s[j];
            }
        }
    }
    __syncthreads();
    const float inv = 1.0f / (float)npix;
    // the row as the next kernels see it: rounded to fp16 (kept in lacc[0 .. C) as floats for the fused LayerNorm)
    float lsum = 0.0f;
    for (int c = threadIdx.x; c < C; c += RP_NT) {
        float v = 0.0f;
        if (mode == 0) {
            float s = 0.0f;
            for (int pl = 0; pl < nplanes; ++pl) s += lacc[pl * C + c];
            v = s * inv;
            if constexpr (LN3) v = fmaf(v, ln3_g[c], ln3_b[c]);
        } else if (mode == 2) {
            v = __builtin_nanf("");
        }
        const half_t hvv = (half_t)v;
        frow[c] = hvv;
        v = (float)hvv;
        lsum += v;
        if (ln_out != nullptr) {
            // plane 0 of lacc is also a SOURCE above: every thread re-reads only its own channels c, written by itself
            lacc[c] = v;
        }
    }
    if (ln_out == nullptr) return;
    // fused LayerNorm1 of the pooled row (attention.py:348: attn(norm1(obj_features), ...)), two-pass, fp32
    auto bsum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < RP_NT / 64; ++w) t += red[w];
        return t;
    };
    const float mean = bsum(lsum) / (float)C;
    float lss = 0.0f;
    for (int c = threadIdx.x; c < C; c += RP_NT) {
        const float d = lacc[c] - mean;
        lss = fmaf(d, d, lss);
    }
    const float rstd = rsqrtf(bsum(lss) / (float)C + 1e-5f);
    half_t* lrow = ln_out + ((size_t)b * slots + i) * C;
    for (int c = threadIdx.x; c < C; c += RP_NT) lrow[c] = (half_t)((lacc[c] - mean) * rstd * ln_g[c] + ln_b[c]);
}

// elementwise over [B, H*W, C]: 8 channels per thread.  XF32: x and y are the fp32 residual stream and hid = LN3(x) is
// RE-EVALUATED in fp32 from the per-row (mean, rstd) the LayerNorm kernel stored, instead of being read back rounded to
// fp16 -- hid enters the residual stream directly here (attention.py:354-358,398), not through a matrix product.
template <bool XF32>
__global__ __launch_bounds__(256) void rela_merge_kernel(const void* __restrict__ xv, const half_t* __restrict__ hid,
                                                         const float* __restrict__ ln_stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const half_t* __restrict__ f,
                                                         int H, int W, int C, const int* __restrict__ rects,
                                                         const int* __restrict__ nvalid, const int* __restrict__ poison,
                                                         int max_objs, int slots, void* __restrict__ yv, size_t total) {
    const int nvec = C / 8;
    const int HW = H * W;
    const float inv_mo = 1.0f / (float)max_objs;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t tok = idx / nvec;
        const int vec = (int)(idx - tok * nvec);
        const int b = (int)(tok / HW);
        const int p = (int)(tok - (size_t)b * HW);
        const int py = p / W, px = p - py * W;
        float xf[8], hf[8];
        if constexpr (XF32) {
            const float* xr = reinterpret_cast<const float*>(xv) + tok * C + vec * 8;
            const float4 a = *reinterpret_cast<const float4*>(xr);
            const float4 c = *reinterpret_cast<const float4*>(xr + 4);
            xf[0] = a.x; xf[1] = a.y; xf[2] = a.z; xf[3] = a.w; xf[4] = c.x; xf[5] = c.y; xf[6] = c.z; xf[7] = c.w;
        } else {
            uint4 rx = ld16(reinterpret_cast<const half_t*>(xv) + tok * C + vec * 8);
            const half8_t x8 = *reinterpret_cast<half8_t*>(&rx);
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[j] = (float)x8[j];
        }
        if (ln_stats != nullptr) {
            const float mean = ln_stats[tok * 2], rstd = ln_stats[tok * 2 + 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) hf[j] = fmaf((xf[j] - mean) * rstd, gamma[vec * 8 + j], beta[vec * 8 + j]);   // explicit FMAs: the row form below must round identically
        } else {
            uint4 rh = ld16(hid + tok * C + vec * 8);
            const half8_t hv = *reinterpret_cast<half8_t*>(&rh);
#pragma unroll
            for (int j = 0; j < 8; ++j) hf[j] = (float)hv[j];
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        const int nv = min(nvalid[b], slots);
        const int* rb = rects + (size_t)b * max_objs * 4;
        for (int i = 0; i < nv; ++i) {
            const int top = rb[4 * i], bottom = rb[4 * i + 1], left = rb[4 * i + 2], right = rb[4 * i + 3];
            if (py >= top && py < bottom && px >= left && px < right) {
                uint4 rf = ld16(f + ((size_t)b * slots + i) * C + vec * 8);
                const half8_t fv = *reinterpret_cast<half8_t*>(&rf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)fv[j];
            }
        }
        const bool bad = poison[b] != 0;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o[j] = 0.5f * (fmaf(acc[j], inv_mo, hf[j]) + xf[j]);
            if (bad) o[j] = __builtin_nanf("");
        }
        if constexpr (XF32) {
            float* yr = reinterpret_cast<float*>(yv) + tok * C + vec * 8;
            *reinterpret_cast<float4*>(yr) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(yr + 4) = make_float4(o[4], o[5], o[6], o[7]);
        } else {
            half8_t ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (half_t)o[j];
            st16(reinterpret_cast<half_t*>(yv) + tok * C + vec * 8, *reinterpret_cast<uint4*>(&ov));
        }
    }
}

// Row form of rela_merge for the fp32 stream, one wave per token: besides y it writes LayerNorm(y) (fp16) -- the norm2
// that feeds attn2's q projection (attention.py:400) -- so the freshly written rows are not read back by a LayerNorm launch.
template <int NV>
__global__ __launch_bounds__(256) void rela_merge_ln_kernel(const float* __restrict__ x, const float* __restrict__ ln_stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const half_t* __restrict__ f, int H, int W, int C,
                                                            const int* __restrict__ rects, const int* __restrict__ nvalid,
                                                            const int* __restrict__ poison, int max_objs, int slots, float* __restrict__ y,
                                                            const float* __restrict__ g2, const float* __restrict__ b2,
                                                            half_t* __restrict__ ln_out, int ntok) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= ntok) return;
    const int nvec = C / 8;
    const int HW = H * W;
    const int b = tok / HW;
    const int p = tok - b * HW;
    const int py = p / W, px = p - py * W;
    const float inv_mo = 1.0f / (float)max_objs;
    const float mean3 = ln_stats[(size_t)tok * 2], rstd3 = ln_stats[(size_t)tok * 2 + 1];
    const int nv = min(nvalid[b], slots);
    const int* rb = rects + (size_t)b * max_objs * 4;
    const bool bad = poison[b] != 0;
    // which boxes cover this pixel: wave-uniform bit mask (max_objs <= 32 checked by the launcher)
    unsigned cover = 0;
    for (int i = 0; i < nv; ++i) {
        const int top = rb[4 * i], bottom = rb[4 * i + 1], left = rb[4 * i + 2], right = rb[4 * i + 3];
        if (py >= top && py < bottom && px >= left && px < right) cover |= 1u << i;
    }
    float o[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vec = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[i][j] = 0.0f;
        if (vec < nvec) {
            const float* xr = x + (size_t)tok * C + vec * 8;
            const float4 a = *reinterpret_cast<const float4*>(xr);
            const float4 c = *reinterpret_cast<const float4*>(xr + 4);
            const float xf[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
            for (unsigned m = cover; m != 0; m &= m - 1) {
                const int bi = __builtin_ctz(m);
                uint4 rf = ld16(f + ((size_t)b * slots + bi) * C + vec * 8);
                const half8_t fv = *reinterpret_cast<half8_t*>(&rf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)fv[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float hf = fmaf((xf[j] - mean3) * rstd3, gamma[vec * 8 + j], beta[vec * 8 + j]);
                o[i][j] = 0.5f * (fmaf(acc[j], inv_mo, hf) + xf[j]);
                if (bad) o[i][j] = __builtin_nanf("");
            }
            float* yr = y + (size_t)tok * C + vec * 8;
            *reinterpret_cast<float4*>(yr) = make_float4(o[i][0], o[i][1], o[i][2], o[i][3]);
            *reinterpret_cast<float4*>(yr + 4) = make_float4(o[i][4], o[i][5], o[i][6], o[i][7]);
        }
    }
    // LayerNorm of the row just formed: same two-pass arithmetic as layernorm_kernel (norms.hip)
    float a = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a += o[i][j];
    const float mean = wave_sum(a) / (float)C;
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dlt = o[i][j] - mean;
                ss = fmaf(dlt, dlt, ss);
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)C + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vec = lane + 64 * i;
        if (vec < nvec) {
            half8_t ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (half_t)fmaf((o[i][j] - mean) * rstd, g2[vec * 8 + j], b2[vec * 8 + j]);
            st16(ln_out + (size_t)tok * C + vec * 8, *reinterpret_cast<uint4*>(&ov));
        }
    }
}

}  // namespace

extern "C" int gl_rela_pool(const void* hid, int32_t B, int32_t H, int32_t W, int32_t C, const int32_t* rects,
                            const int32_t* nvalid, const int32_t* poison, int32_t max_objs, int32_t slots, void* feat, const float* ln_gamma,
                            const float* ln_beta, void* ln_out, void* stream) {
    if (!hid || !rects || !nvalid || !poison || !feat || C <= 0 || (C % 8) || C > RELA_MAX_C) return GL_ERR_BAD_ARG;
    if (ln_out != nullptr && (!ln_gamma || !ln_beta)) return GL_ERR_BAD_ARG;
    if (slots <= 0) slots = max_objs;
    if (slots > max_objs) return GL_ERR_BAD_ARG;
    rela_pool_kernel<false><<<dim3(slots, B), dim3(RP_NT), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const half_t*>(hid), nullptr, nullptr, nullptr, nullptr, H, W, C, rects, nvalid, poison, max_objs, slots,
        reinterpret_cast<half_t*>(feat), ln_gamma, ln_beta, reinterpret_cast<half_t*>(ln_out));
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_rela_pool_ln3(const float* x, const float* ln3_stats, const float* ln3_gamma, const float* ln3_beta, int32_t B, int32_t H,
                                int32_t W, int32_t C, const int32_t* rects, const int32_t* nvalid, const int32_t* poison, int32_t max_objs,
                                int32_t slots, void* feat, const float* ln_gamma, const float* ln_beta, void* ln_out, void* stream) {
    if (!x || !ln3_stats || !ln3_gamma || !ln3_beta || !rects || !nvalid || !poison || !feat || C <= 0 || (C % 8) || C > RELA_MAX_C)
        return GL_ERR_BAD_ARG;
    if (ln_out != nullptr && (!ln_gamma || !ln_beta)) return GL_ERR_BAD_ARG;
    if (slots <= 0) slots = max_objs;
    if (slots > max_objs) return GL_ERR_BAD_ARG;
    rela_pool_kernel<true><<<dim3(slots, B), dim3(RP_NT), 0, (hipStream_t)stream>>>(
        nullptr, x, ln3_stats, ln3_gamma, ln3_beta, H, W, C, rects, nvalid, poison, max_objs, slots, reinterpret_cast<half_t*>(feat), ln_gamma, ln_beta,
        reinterpret_cast<half_t*>(ln_out));
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_rela_merge(const void* x, int32_t x_f32, const void* hid, const float* ln_stats, const float* gamma,
                             const float* beta, const void* f, int32_t B, int32_t H, int32_t W, int32_t C,
                             const int32_t* rects, const int32_t* nvalid, const int32_t* poison, int32_t max_objs, int32_t slots, void* y,
                             const float* ln2_gamma, const float* ln2_beta, void* ln2_out, void* stream) {
    if (!x || !f || !rects || !nvalid || !poison || !y || C <= 0 || (C % 8)) return GL_ERR_BAD_ARG;
    if (slots <= 0) slots = max_objs;
    if (slots > max_objs) return GL_ERR_BAD_ARG;
    if (ln_stats ? (!gamma || !beta) : !hid) return GL_ERR_BAD_ARG;
    if (ln2_out != nullptr) {
        // fused form: fp32 stream + recomputed LN3 only, one wave per token
        if (!x_f32 || !ln_stats || !ln2_gamma || !ln2_beta || slots > 32 || C > 2048) return GL_ERR_UNSUPPORTED;
        const int ntok = B * H * W;
        const int nv = gl_cdiv(C / 8, 64);
        const dim3 grid(gl_cdiv(ntok, 4)), blk(256);
#define GL_RM(V)                                                                                                                         \
    rela_merge_ln_kernel<V><<<grid, blk, 0, (hipStream_t)stream>>>(reinterpret_cast<const float*>(x), ln_stats, gamma, beta,              \
                                                                   reinterpret_cast<const half_t*>(f), H, W, C, rects, nvalid, poison,    \
                                                                   max_objs, slots, reinterpret_cast<float*>(y), ln2_gamma, ln2_beta,      \
                                                                   reinterpret_cast<half_t*>(ln2_out), ntok)
        if (nv == 1) GL_RM(1); else if (nv == 2) GL_RM(2); else if (nv == 3) GL_RM(3); else GL_RM(4);
#undef GL_RM
        GL_CHECK_LAUNCH();
        return 0;
    }
    const size_t total = (size_t)B * H * W * (C / 8);
    int nblk = (int)((total + 255) / 256);
    if (nblk > 2048) nblk = 2048;
    const half_t* hp = reinterpret_cast<const half_t*>(hid);
    const half_t* fp = reinterpret_cast<const half_t*>(f);
    if (x_f32)
        rela_merge_kernel<true><<<dim3(nblk), dim3(256), 0, (hipStream_t)stream>>>(x, hp, ln_stats, gamma, beta, fp, H, W, C, rects,
                                                                                  nvalid, poison, max_objs, slots, y, total);
    else
        rela_merge_kernel<false><<<dim3(nblk), dim3(256), 0, (hipStream_t)stream>>>(x, hp, ln_stats, gamma, beta, fp, H, W, C, rects,
                                                                                   nvalid, poison, max_objs, slots, y, total);
    GL_CHECK_LAUNCH();
    return 0;
}
