// GroupNorm(32) (+SiLU, + two-source channel concat) and LayerNorm over token-major fp16 activations,
// fp32 statistics.  Both are HBM-bound: 16-byte coalesced loads, per-thread partial sums, LDS / wave
// shuffle reductions.
//
// GroupNorm32 (util.py:226-228, eps 1e-5) + SiLU feeds every ResBlock conv (openaimodel.py:155-157,
// 180-181) and the out conv (:385-388); Normalize (attention.py:78-79, eps 1e-6) feeds proj_in.
// In NHWC a group is C/32 contiguous channels of every pixel, so statistics are gathered in two
// stages: gn_stats writes per-(sample, pixel-chunk, group) partial (sum, sumsq); gn_apply re-reduces
// the partials of its sample (<= 64 chunks), then normalises.  The reduction order is fixed
// (bitwise reproducible, no atomics).
#include "common.h"
#include "gligen_hip.h"

namespace {

__device__ __forceinline__ const half_t* src_ptr(const half_t* x1, int C1, const half_t* x2, int C2, size_t pix,
                                                 int c, int* cl) {
    // channel c of the virtual concat [x1 | x2] at flat pixel index `pix`
    if (c < C1) { *cl = c; return x1 + pix * C1; }
    *cl = c - C1;
    return x2 + pix * C2;
}

// grid (nchunk, B); block 256.  thread -> fixed 8-channel vector(s), strided over the chunk's pixels.
// Per-channel partial sums go to LDS as [pixel-lane][channel]; 32 threads then fold them per group in
// a fixed order (bitwise reproducible).
constexpr int GN_MAX_C = 2560;
__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* __restrict__ x1, int C1,
                                                       const half_t* __restrict__ x2, int C2, int HW, int nchunk,
                                                       float* __restrict__ partial) {
    __shared__ float lsum[GN_MAX_C];
    __shared__ float lsq[GN_MAX_C];
    const int C = C1 + C2;
    const int cpg = C / 32;
    const int nvec = C / 8;
    const int b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int ppc = (HW + nchunk - 1) / nchunk;
    const int p0 = chunk * ppc;
    const int p1 = min(HW, p0 + ppc);

    const int vlanes = min(nvec, 256);       // vectors handled side by side
    const int nplanes = 256 / vlanes;        // pixel lanes (nplanes * C <= 2048 when nvec <= 256)
    const int plane = threadIdx.x / vlanes;
    const int v0 = threadIdx.x - plane * vlanes;
    if (plane < nplanes) {
        for (int vec = v0; vec < nvec; vec += vlanes) {
            float s[8], ss[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = 0.0f; ss[j] = 0.0f; }
            const int c = vec * 8;
            int cl;
            const half_t* base = src_ptr(x1, C1, x2, C2, (size_t)b * HW + p0 + plane, c, &cl) + cl;
            const size_t step = (size_t)nplanes * (c < C1 ? C1 : C2);
            int pix = p0 + plane;
            // 4 independent 16-byte loads in flight per thread
            for (; pix + 3 * nplanes < p1; pix += 4 * nplanes, base += 4 * step) {
                uint4 raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) raw[u] = ld16(base + u * step);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const half8_t hv = *reinterpret_cast<half8_t*>(&raw[u]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = (float)hv[j];
                        s[j] += f;
                        ss[j] += f * f;
                    }
                }
            }
            for (; pix < p1; pix += nplanes, base += step) {
                uint4 raw = ld16(base);
                const half8_t hv = *reinterpret_cast<half8_t*>(&raw);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = (float)hv[j];
                    s[j] += f;
                    ss[j] += f * f;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lsum[plane * C + c + j] = s[j];
                lsq[plane * C + c + j] = ss[j];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int g = threadIdx.x;
        float s = 0.0f, ss = 0.0f;
        for (int pl = 0; pl < nplanes; ++pl)
            for (int cc = 0; cc < cpg; ++cc) {
                s += lsum[pl * C + g * cpg + cc];
                ss += lsq[pl * C + g * cpg + cc];
            }
        float* pp = partial + (((size_t)b * nchunk + chunk) * 32 + g) * 2;
        pp[0] = s;
        pp[1] = ss;
    }
}

// grid (nblk, B); block 256: normalise (+SiLU) the virtual concat into out [B, HW, C].
// A thread owns fixed 8-channel vectors (like gn_stats), so the per-channel affine
//   y = x * (rstd*gamma) + (beta - mean*rstd*gamma)
// is folded into 8 (scale, shift) register pairs once and the pixel loop is one 16-byte load, 8 FMAs
// (+SiLU) and one 16-byte store with no integer division.
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ x1, int C1,
                                                       const half_t* __restrict__ x2, int C2, int HW, int nchunk,
                                                       const float* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int silu, half_t* __restrict__ out, int ppb) {
    __shared__ float mean_s[32], rstd_s[32];
    __shared__ float red_s[8][32], red_q[8][32];
    const int C = C1 + C2;
    const int cpg = C / 32;
    const int nvec = C / 8;
    const int b = blockIdx.y;
    {
        // fold the <= 64 per-chunk partials: 8 slices x 32 groups in parallel (independent loads), then a
        // fixed-order sum over the slices -- the serial 64-step version cost ~25 us of dependent L2 latency
        const int g = threadIdx.x & 31, sl = threadIdx.x >> 5;
        float s = 0.0f, ss = 0.0f;
        for (int ch = sl; ch < nchunk; ch += 8) {
            const float* pp = partial + (((size_t)b * nchunk + ch) * 32 + g) * 2;
            s += pp[0];
            ss += pp[1];
        }
        red_s[sl][g] = s;
        red_q[sl][g] = ss;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.0f, ss = 0.0f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) { s += red_s[sl][threadIdx.x]; ss += red_q[sl][threadIdx.x]; }
        const float n = (float)cpg * (float)HW;
        const float mean = s / n;
        float var = ss / n - mean * mean;
        var = fmaxf(var, 0.0f);
        mean_s[threadIdx.x] = mean;
        rstd_s[threadIdx.x] = rsqrtf(var + eps);
    }
    __syncthreads();
    const int p0 = blockIdx.x * ppb;
    const int p1 = min(HW, p0 + ppb);
    const int vlanes = min(nvec, 256);
    const int nplanes = 256 / vlanes;
    const int plane = threadIdx.x / vlanes;
    const int v0 = threadIdx.x - plane * vlanes;
    if (plane >= nplanes) return;
    for (int vec = v0; vec < nvec; vec += vlanes) {
        const int c = vec * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (c + j) / cpg;
            const float a = rstd_s[g] * gamma[c + j];
            sc[j] = a;
            sh[j] = beta[c + j] - mean_s[g] * a;
        }
        const half_t* src;
        int cs, cl;
        if (c < C1) { src = x1; cs = C1; cl = c; } else { src = x2; cs = C2; cl = c - C1; }
        const half_t* sp = src + ((size_t)b * HW + p0 + plane) * cs + cl;
        half_t* dp = out + ((size_t)b * HW + p0 + plane) * C + c;
        const size_t sstep = (size_t)nplanes * cs, dstep = (size_t)nplanes * C;
        int pix = p0 + plane;
        for (; pix + 3 * nplanes < p1; pix += 4 * nplanes, sp += 4 * sstep, dp += 4 * dstep) {
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = ld16(sp + u * sstep);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const half8_t hv = *reinterpret_cast<half8_t*>(&raw[u]);
                half8_t ov;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = fmaf((float)hv[j], sc[j], sh[j]);
                    if (silu) v = silu_f(v);
                    ov[j] = (half_t)v;
                }
                st16(dp + u * dstep, *reinterpret_cast<uint4*>(&ov));
            }
        }
        for (; pix < p1; pix += nplanes, sp += sstep, dp += dstep) {
            uint4 raw = ld16(sp);
            const half8_t hv = *reinterpret_cast<half8_t*>(&raw);
            half8_t ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = fmaf((float)hv[j], sc[j], sh[j]);
                if (silu) v = silu_f(v);
                ov[j] = (half_t)v;
            }
            st16(dp, *reinterpret_cast<uint4*>(&ov));
        }
    }
}

// One wave per row, RPW rows per wave with all their loads issued before the first reduction (the kernel is
// latency-bound: 8k resident waves x one 640-byte row each left HBM at ~30 %); up to NV x 64 8-channel
// vectors per row.
template <int RPW, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, int ldx, half_t* __restrict__ y,
                                                        int ldy, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int nrows, int rows_in,
                                                        int rows_out, int row_off, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= nrows) return;
    const int nvec = C / 8;
    half8_t v[RPW][NV];
    float s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, nrows - 1);
        const half_t* xr = x + (size_t)row * ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vec = lane + 64 * i;
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (vec < nvec) raw = ld16(xr + vec * 8);
            v[r][i] = *reinterpret_cast<half8_t*>(&raw);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) a += (float)v[r][i][j];      // lanes past the row hold zeros
        s[r] = a;
    }
    float mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) mean[r] = wave_sum(s[r]) / (float)C;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vec = lane + 64 * i;
            if (vec < nvec) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dlt = (float)v[r][i][j] - mean[r];
                    ss += dlt * dlt;
                }
            }
        }
        s[r] = ss;
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) rstd[r] = rsqrtf(wave_sum(s[r]) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vec = lane + 64 * i;
        if (vec < nvec) {
            float g[8], bt[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                g[j] = gamma[vec * 8 + j];
                bt[j] = beta[vec * 8 + j];
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = row0 + r;
                if (row < nrows) {
                    const int bidx = row / rows_in;
                    const int i_in = row - bidx * rows_in;
                    half_t* yr = y + ((size_t)bidx * rows_out + row_off + i_in) * ldy;
                    half8_t ov;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (half_t)(((float)v[r][i][j] - mean[r]) * rstd[r] * g[j] + bt[j]);
                    st16(yr + vec * 8, *reinterpret_cast<uint4*>(&ov));
                }
            }
        }
    }
}

template <int RPW, int NV>
void launch_ln(const half_t* x, int ldx, half_t* y, int ldy, const float* gamma, const float* beta, int nrows,
               int rows_in, int rows_out, int row_off, int C, float eps, hipStream_t st) {
    layernorm_kernel<RPW, NV><<<dim3(gl_cdiv(nrows, 4 * RPW)), dim3(256), 0, st>>>(x, ldx, y, ldy, gamma, beta, nrows, rows_in,
                                                                             rows_out, row_off, C, eps);
}

int g_ln_rpw = 0;   // rows per wave override for A/B (0 = auto)
int g_gn_ppb = 16;  // GroupNorm apply: pixels per pixel-lane per block (A/B knob 16)

}  // namespace

extern "C" int gl_set_option_norm(int key, int value) {
    if (key == 11) { g_ln_rpw = value; return 0; }
    if (key == 16) { g_gn_ppb = value > 0 ? value : 16; return 0; }
    return GL_ERR_BAD_ARG;
}

extern "C" int gl_groupnorm_stats(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t B, int32_t HW,
                                  float* partial, int32_t nchunk, void* stream) {
    const int C = C1 + (x2 ? C2 : 0);
    if (!x1 || !partial || C <= 0 || C > GN_MAX_C || (C % 32) || (C1 % 8) || (x2 && (C2 % 8)) || nchunk <= 0 || nchunk > HW)
        return GL_ERR_BAD_ARG;
    gn_stats_kernel<<<dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const half_t*>(x1), C1, reinterpret_cast<const half_t*>(x2), x2 ? C2 : 0, HW, nchunk, partial);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_groupnorm_apply(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t B, int32_t HW,
                                  const float* partial, int32_t nchunk, const float* gamma, const float* beta,
                                  float eps, int32_t silu, void* out, void* stream) {
    const int C = C1 + (x2 ? C2 : 0);
    if (!x1 || !partial || !gamma || !beta || !out || C <= 0 || (C % 32) || (C1 % 8) || (x2 && (C2 % 8)))
        return GL_ERR_BAD_ARG;
    // pixels per block: >= 2 pixels per pixel-lane, ~1-2k blocks at the 64x64 level
    const int nvec = C / 8;
    const int nplanes = 256 / (nvec < 256 ? nvec : 256);
    int ppb = g_gn_ppb * nplanes;
    if (ppb < 16) ppb = 16;
    if (ppb > HW) ppb = HW;
    const int nblk = gl_cdiv(HW, ppb);
    gn_apply_kernel<<<dim3(nblk, B), dim3(256), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const half_t*>(x1), C1, reinterpret_cast<const half_t*>(x2), x2 ? C2 : 0, HW, nchunk, partial,
        gamma, beta, eps, silu, reinterpret_cast<half_t*>(out), ppb);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_layernorm(const void* x, int32_t ldx, void* y, int32_t ldy, const float* gamma, const float* beta,
                            int32_t B, int32_t rows_in, int32_t rows_out, int32_t row_off, int32_t C, float eps,
                            void* stream) {
    if (!x || !y || !gamma || !beta || C <= 0 || (C % 8) || C > 2048 || (ldx % 8) || (ldy % 8)) return GL_ERR_BAD_ARG;
    const int nrows = B * rows_in;
    if (nrows <= 0) return GL_ERR_BAD_ARG;
    const half_t* xp = reinterpret_cast<const half_t*>(x);
    half_t* yp = reinterpret_cast<half_t*>(y);
    hipStream_t st = (hipStream_t)stream;
    const int nv = gl_cdiv(C / 8, 64);
    // rows per wave: 2 / 4 measured equal to 1 on MI355X (12.1-12.9 us at 32768 x 320) -- the kernel sits on its
    // launch + dependent-load latency floor, not on bytes in flight -- so 1 stays the default (option 11 = A/B)
    int rpw = g_ln_rpw ? g_ln_rpw : 1;
    if (nv > 2 && rpw > 2) rpw = 2;
#define GL_LN(R, V) launch_ln<R, V>(xp, ldx, yp, ldy, gamma, beta, nrows, rows_in, rows_out, row_off, C, eps, st)
    if (nv == 1) { if (rpw == 4) GL_LN(4, 1); else if (rpw == 2) GL_LN(2, 1); else GL_LN(1, 1); }
    else if (nv == 2) { if (rpw == 4) GL_LN(4, 2); else if (rpw == 2) GL_LN(2, 2); else GL_LN(1, 2); }
    else if (nv == 3) { if (rpw == 2) GL_LN(2, 3); else GL_LN(1, 3); }
    else { if (rpw == 2) GL_LN(2, 4); else GL_LN(1, 4); }
#undef GL_LN
    GL_CHECK_LAUNCH();
    return 0;
}
