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
#include "opts.h"

namespace {

// An 8-channel piece of an input row, fp16 (one 16-byte load) or fp32 (the residual stream: two 16-byte loads).  The
// register-resident kernels keep the piece in its storage form (4 or 8 VGPRs).
template <bool XF32> struct GnVec;
template <> struct GnVec<false> {
    uint4 r;
    __device__ __forceinline__ void zero() { r = make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ void load(const void* base, size_t elem) { r = ld16(reinterpret_cast<const half_t*>(base) + elem); }
    __device__ __forceinline__ void get(float (&v)[8]) {
        // opaque to the optimiser: the register-resident kernels call get() once per pass, and without this the converted
        // floats of ALL vectors are kept live across the passes (8 instead of 4 VGPRs per vector: hundreds of spills at NV = 22)
        asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
        const half8_t hv = *reinterpret_cast<const half8_t*>(&r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)hv[j];
    }
};
template <> struct GnVec<true> {
    float4 a, b;
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
    __device__ __forceinline__ void load(const void* base, size_t elem) {
        const float* p = reinterpret_cast<const float*>(base) + elem;
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void get(float (&v)[8]) const {
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};

// fp16 store of 8 values; with `lo` also the fp16 residual v - float(fp16(v)) (split-fp16 operand, DESIGN.md 4: the
// consumer multiplies [hi | lo] against [W | W], which restores ~22 mantissa bits of the activation operand)
__device__ __forceinline__ void store_hl(half_t* hi, half_t* lo, const float (&v)[8]) {
    if (lo != nullptr) {
        // hi and lo from ONE pinned value each (common.h pin_value)
        half8_t h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = pin_value(v[j]);
            h[j] = (half_t)x;
            l[j] = (half_t)(x - (float)h[j]);
        }
        st16(hi, *reinterpret_cast<uint4*>(&h));
        st16(lo, *reinterpret_cast<uint4*>(&l));
        return;
    }
    half8_t h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
    st16(hi, *reinterpret_cast<uint4*>(&h));
}

// outputs of one GroupNorm launch (all row-major over the B * HW pixels):
//   out [*, ldo] fp16 normalised rows; out_lo (optional, same stride) their fp16 residuals;
//   raw (optional) [*, ldraw]: columns [0, C) = fp16(x), [C, 2C) = fp16(x - fp16(x)) of the INPUT concat
struct GnOut {
    half_t* out; half_t* out_lo; int ldo;
    half_t* raw; int ldraw;
};

// grid (nchunk, B); block 256.  thread -> fixed 8-channel vector(s), strided over the chunk's pixels.
//
// Statistics are gathered Welford-style (torch's GroupNorm uses a Welford update; a one-pass E[x^2] - mean^2 in
// fp32 loses the variance of a channel whose |mean| >> std, which real checkpoints have): every thread sums
// (x - s) and (x - s)^2 about a per-channel shift s = the first value it sees, converts to (mean, M2 = sum of squared
// deviations) and the partials are merged with the exact pairwise formula
//     mean = sum_i n_i mean_i / N ,   M2 = sum_i [ M2_i + n_i (mean_i - mean)^2 ]
// in a fixed order (bitwise reproducible, no atomics).  partial[b][chunk][group] = (mean, M2); the element count of
// a chunk follows from the geometry.
constexpr int GN_MAX_C = 2560;
template <bool XF32>
__global__ __launch_bounds__(256) void gn_stats_kernel(const void* __restrict__ x1, int C1,
                                                       const void* __restrict__ x2, int C2, int HW, int nchunk,
                                                       float* __restrict__ partial) {
    __shared__ float lmean[GN_MAX_C];
    __shared__ float lm2[GN_MAX_C];
    __shared__ float lcnt[256];
    const int C = C1 + C2;
    const int cpg = C / 32;
    const int nvec = C / 8;
    const int b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int ppc = (HW + nchunk - 1) / nchunk;
    const int p0 = min(HW, chunk * ppc);
    const int p1 = min(HW, p0 + ppc);

    const int vlanes = min(nvec, 256);       // vectors handled side by side
    const int nplanes = 256 / vlanes;        // pixel lanes (nplanes * C <= 2048 when nvec <= 256)
    const int plane = threadIdx.x / vlanes;
    const int v0 = threadIdx.x - plane * vlanes;
    if (plane < nplanes) {
        const int first = p0 + plane;
        const int npix = first < p1 ? (p1 - first + nplanes - 1) / nplanes : 0;
        if (v0 == 0) lcnt[plane] = (float)npix;
        for (int vec = v0; vec < nvec; vec += vlanes) {
            float s[8], ss[8], sh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = 0.0f; ss[j] = 0.0f; sh[j] = 0.0f; }
            const int c = vec * 8;
            if (npix > 0) {
                const void* src = c < C1 ? x1 : x2;
                const int cs = c < C1 ? C1 : C2;
                size_t e = ((size_t)b * HW + first) * cs + (c < C1 ? c : c - C1);
                const size_t step = (size_t)nplanes * cs;
                {
                    GnVec<XF32> r0;
                    r0.load(src, e);
                    r0.get(sh);
                }
                int pix = first;
                // 4 independent loads in flight per thread
                for (; pix + 3 * nplanes < p1; pix += 4 * nplanes, e += 4 * step) {
                    GnVec<XF32> raw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) raw[u].load(src, e + u * step);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float xv[8];
                        raw[u].get(xv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float f = xv[j] - sh[j];
                            s[j] += f;
                            ss[j] = fmaf(f, f, ss[j]);
                        }
                    }
                }
                for (; pix < p1; pix += nplanes, e += step) {
                    GnVec<XF32> raw;
                    raw.load(src, e);
                    float xv[8];
                    raw.get(xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = xv[j] - sh[j];
                        s[j] += f;
                        ss[j] = fmaf(f, f, ss[j]);
                    }
                }
            }
            const float inv = npix > 0 ? 1.0f / (float)npix : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dm = s[j] * inv;                      // mean - shift
                lmean[plane * C + c + j] = sh[j] + dm;
                lm2[plane * C + c + j] = fmaxf(ss[j] - s[j] * dm, 0.0f);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int g = threadIdx.x;
        float sw = 0.0f, wsum = 0.0f;
        for (int pl = 0; pl < nplanes; ++pl) {
            const float n = lcnt[pl];
            for (int cc = 0; cc < cpg; ++cc) sw = fmaf(n, lmean[pl * C + g * cpg + cc], sw);
            wsum += n * (float)cpg;
        }
        const float mean = wsum > 0.0f ? sw / wsum : 0.0f;
        float m2 = 0.0f;
        for (int pl = 0; pl < nplanes; ++pl) {
            const float n = lcnt[pl];
            for (int cc = 0; cc < cpg; ++cc) {
                const float d = lmean[pl * C + g * cpg + cc] - mean;
                m2 += lm2[pl * C + g * cpg + cc] + n * d * d;
            }
        }
        float* pp = partial + (((size_t)b * nchunk + chunk) * 32 + g) * 2;
        pp[0] = mean;
        pp[1] = m2;
    }
}

// grid (nblk, B); block 256: normalise (+SiLU) the virtual concat into out [B, HW, ldo].
// A thread owns fixed 8-channel vectors (like gn_stats), so the per-channel affine
//   y = x * (rstd*gamma) + (beta - mean*rstd*gamma)
// is folded into 8 (scale, shift) register pairs once and the pixel loop is one 16-byte load, 8 FMAs
// (+SiLU) and one 16-byte store with no integer division.
template <bool XF32>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* __restrict__ x1, int C1,
                                                       const void* __restrict__ x2, int C2, int HW, int nchunk,
                                                       const float* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int silu, GnOut o, int ppb) {
    __shared__ float mean_s[32], rstd_s[32];
    __shared__ float red_s[8][32], red_q[8][32];
    const int C = C1 + C2;
    const int cpg = C / 32;
    const int nvec = C / 8;
    const int b = blockIdx.y;
    {
        // merge the <= 64 per-chunk (mean, M2) partials (weights n_c = chunk pixels x channels per group): 8 slices x 32
        // groups in parallel (independent loads), then fixed-order sums over the slices; two passes (mean, then M2)
        const int g = threadIdx.x & 31, sl = threadIdx.x >> 5;
        const int ppc = (HW + nchunk - 1) / nchunk;
        float sw = 0.0f;
        for (int ch = sl; ch < nchunk; ch += 8) {
            const float n = (float)(max(0, min(HW, (ch + 1) * ppc) - min(HW, ch * ppc)) * cpg);
            sw = fmaf(n, partial[(((size_t)b * nchunk + ch) * 32 + g) * 2], sw);
        }
        red_s[sl][g] = sw;
        __syncthreads();
        if (threadIdx.x < 32) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red_s[k][threadIdx.x];
            mean_s[threadIdx.x] = s / ((float)cpg * (float)HW);
        }
        __syncthreads();
        const float mean = mean_s[g];
        float m2 = 0.0f;
        for (int ch = sl; ch < nchunk; ch += 8) {
            const float n = (float)(max(0, min(HW, (ch + 1) * ppc) - min(HW, ch * ppc)) * cpg);
            const float* pp = partial + (((size_t)b * nchunk + ch) * 32 + g) * 2;
            const float d = pp[0] - mean;
            m2 += pp[1] + n * d * d;
        }
        red_q[sl][g] = m2;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float ss = 0.0f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) ss += red_q[sl][threadIdx.x];
        const float var = ss / ((float)cpg * (float)HW);
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
        const void* src = c < C1 ? x1 : x2;
        const int cs = c < C1 ? C1 : C2;
        size_t e = ((size_t)b * HW + p0 + plane) * cs + (c < C1 ? c : c - C1);
        const size_t row0 = (size_t)b * HW + p0 + plane;
        half_t* dp = o.out + row0 * o.ldo + c;
        half_t* dl = o.out_lo ? o.out_lo + row0 * o.ldo + c : nullptr;
        half_t* rp = o.raw ? o.raw + row0 * o.ldraw + c : nullptr;
        const size_t sstep = (size_t)nplanes * cs, dstep = (size_t)nplanes * o.ldo, rstep = (size_t)nplanes * o.ldraw;
        auto one = [&](GnVec<XF32>& raw, size_t u) {
            float xv[8], yv[8];
            raw.get(xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = fmaf(xv[j], sc[j], sh[j]);
                if (silu) v = silu_f(v);
                yv[j] = v;
            }
            store_hl(dp + u * dstep, dl ? dl + u * dstep : nullptr, yv);
            if (rp) store_hl(rp + u * rstep, rp + u * rstep + C, xv);
        };
        int pix = p0 + plane;
        for (; pix + 3 * nplanes < p1; pix += 4 * nplanes, e += 4 * sstep, dp += 4 * dstep) {
            GnVec<XF32> raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u].load(src, e + u * sstep);
#pragma unroll
            for (int u = 0; u < 4; ++u) one(raw[u], u);
            if (dl) dl += 4 * dstep;
            if (rp) rp += 4 * rstep;
        }
        for (; pix < p1; pix += nplanes, e += sstep, dp += dstep) {
            GnVec<XF32> raw;
            raw.load(src, e);
            one(raw, 0);
            if (dl) dl += dstep;
            if (rp) rp += rstep;
        }
    }
}

// Small maps (the 16x16 / 8x8 levels: 1280- / 2560-channel tensors of 64..256 pixels): ONE launch, one block per (sample,
// group).  The group's HW x (C/32) slab (<= 20 K elements) is loaded once into registers, reduced two-pass (mean, then
// squared deviations: exact, no E[x^2] - mean^2 cancellation) and normalised from the registers -- instead of
// gn_stats + gn_apply, which at these sizes are two latency-bound launches that each re-read the tensor.
// Needs whole 8-channel vectors per group (C % 256 == 0) and HW * C / 256 <= 256 * NV vectors.
template <bool XF32, int NV>
__global__ __launch_bounds__(256) void gn_fused_kernel(const void* __restrict__ x1, int C1, const void* __restrict__ x2, int C2,
                                                       int HW, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int silu, GnOut o) {
    __shared__ float red[4];
    const int C = C1 + C2;
    const int cpg = C / 32;
    const int nv = cpg / 8;
    const int total = HW * nv;
    const int g = blockIdx.x, b = blockIdx.y;
    auto bsum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    GnVec<XF32> raw[NV];
    int chan[NV], pix[NV];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = threadIdx.x + 256 * k;
        raw[k].zero();
        chan[k] = -1;
        pix[k] = 0;
        if (idx < total) {
            const int p = idx / nv;
            const int c = g * cpg + (idx - p * nv) * 8;
            chan[k] = c;
            pix[k] = p;
            if (c < C1) raw[k].load(x1, ((size_t)b * HW + p) * C1 + c); else raw[k].load(x2, ((size_t)b * HW + p) * C2 + (c - C1));
            float xv[8];
            raw[k].get(xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += xv[j];
        }
    }
    const float n = (float)cpg * (float)HW;
    const float mean = bsum(s) / n;
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (chan[k] >= 0) {
            float xv[8];
            raw[k].get(xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = xv[j] - mean;
                ss = fmaf(d, d, ss);
            }
        }
    }
    const float rstd = rsqrtf(bsum(ss) / n + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (chan[k] >= 0) {
            const int c = chan[k];
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + c), g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + c), b1 = *reinterpret_cast<const float4*>(beta + c + 4);
            const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float xv[8], yv[8];
            raw[k].get(xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = (xv[j] - mean) * rstd * gm[j] + bt[j];
                if (silu) v = silu_f(v);
                yv[j] = v;
            }
            const size_t row = (size_t)b * HW + pix[k];
            store_hl(o.out + row * o.ldo + c, o.out_lo ? o.out_lo + row * o.ldo + c : nullptr, yv);
            if (o.raw) store_hl(o.raw + row * o.ldraw + c, o.raw + row * o.ldraw + C + c, xv);
        }
    }
}

// Single-launch GroupNorm for channel widths whose groups are NOT whole 8-channel vectors (C = 320 / 640 / 960 / 1920:
// 10 / 20 / 30 / 60 channels per group): one block per (sample, BUNDLE of groups) where a bundle is the smallest run of
// groups that is a whole number of vectors (lcm(cpg, 8) channels: 4 / 2 / 4 / 2 groups).  The bundle's HW x BC slab is
// loaded once into registers (NV vectors per thread, up to 1024 threads), per-group statistics are taken two-pass (a
// vector spans at most two groups: it contributes a low and a high part), and the rows are normalised from the
// registers -- one read + one write of the tensor instead of gn_stats (read) + gn_apply (read + write) in two
// latency-bound launches.  80..240-byte pieces per pixel: ~80 % sector efficiency, still far cheaper than a second pass.
template <bool XF32, int NV, int NT>
__global__ __launch_bounds__(NT) void gn_bundle_kernel(const void* __restrict__ x1, int C1, const void* __restrict__ x2, int C2,
                                                       int HW, int cpg, int gb, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int silu, GnOut o) {
    constexpr int NWV = (NT + 63) / 64;
    __shared__ float red[NWV][4];
    __shared__ float stat[2][4];
    const int C = C1 + C2;
    const int bc = cpg * gb;                 // channels per bundle (multiple of 8)
    const int nvp = bc / 8;                  // vectors per pixel; NT % nvp == 0 (launcher), so a thread keeps ONE channel vector
    const int ps = NT / nvp;                 // pixels covered per pass of the block
    const int v = threadIdx.x % nvp;
    const int pl = threadIdx.x / nvp;
    const int c = blockIdx.x * bc + v * 8;   // this thread's 8 channels
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6;
    const int g0 = (v * 8) / cpg;            // group (within the bundle) of the vector's first channel
    const int js = (g0 + 1) * cpg - v * 8;   // elements [0, js) belong to g0, the rest to g0 + 1 (a vector spans <= 2 groups)
    const void* src = (c < C1) ? x1 : x2;
    const int sstride = (c < C1) ? C1 : C2;
    const size_t e0 = (size_t)b * HW * sstride + ((c < C1) ? c : c - C1);
    // block-wide sums of the (up to 4) per-group values
    auto bsum4 = [&](float lo, float hi, int slot) {
        float vq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) vq[q] = wave_sum((q == g0 ? lo : 0.0f) + (q == g0 + 1 ? hi : 0.0f));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave][q] = vq[q];
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            float t = 0.0f;
            for (int w = 0; w < NWV; ++w) t += red[w][threadIdx.x];
            stat[slot][threadIdx.x] = t;
        }
        __syncthreads();
    };
    GnVec<XF32> raw[NV];
    float lo = 0.0f, hi = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int p = pl + ps * k;
        raw[k].zero();
        if (p < HW) {
            raw[k].load(src, e0 + (size_t)p * sstride);
            float xv[8];
            raw[k].get(xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < js) lo += xv[j]; else hi += xv[j];
            }
        }
    }
    const float n = (float)cpg * (float)HW;
    bsum4(lo, hi, 0);
    const int g1 = g0 + 1 < 4 ? g0 + 1 : 3;
    const float m0 = stat[0][g0] / n, m1 = stat[0][g1] / n;
    lo = 0.0f; hi = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (pl + ps * k < HW) {
            float xv[8];
            raw[k].get(xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = xv[j] - (j < js ? m0 : m1);
                if (j < js) lo = fmaf(d, d, lo); else hi = fmaf(d, d, hi);
            }
        }
    }
    bsum4(lo, hi, 1);
    const float r0 = rsqrtf(stat[1][g0] / n + eps), r1 = rsqrtf(stat[1][g1] / n + eps);
    // per-channel scale / shift: y = x * sc + sh
    float sc[8], sh[8];
    {
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c), gb4 = *reinterpret_cast<const float4*>(gamma + c + 4);
        const float4 ba = *reinterpret_cast<const float4*>(beta + c), bb4 = *reinterpret_cast<const float4*>(beta + c + 4);
        const float gm[8] = {ga.x, ga.y, ga.z, ga.w, gb4.x, gb4.y, gb4.z, gb4.w};
        const float bt[8] = {ba.x, ba.y, ba.z, ba.w, bb4.x, bb4.y, bb4.z, bb4.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float m = j < js ? m0 : m1, r = j < js ? r0 : r1;
            sc[j] = r * gm[j];
            sh[j] = bt[j] - m * sc[j];
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int p = pl + ps * k;
        if (p < HW) {
            float xv[8], yv[8];
            raw[k].get(xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float val = fmaf(xv[j], sc[j], sh[j]);
                if (silu) val = silu_f(val);
                yv[j] = val;
            }
            const size_t row = (size_t)b * HW + p;
            store_hl(o.out + row * o.ldo + c, o.out_lo ? o.out_lo + row * o.ldo + c : nullptr, yv);
            if (o.raw) store_hl(o.raw + row * o.ldraw + c, o.raw + row * o.ldraw + C + c, xv);
        }
    }
}

// One wave per row; up to NV x 64 8-channel vectors per row.  XF32: the input row is fp32 (the residual stream), else
// fp16.  Two-pass statistics in registers (mean, then sum of squared deviations).  Optionally stores (mean, rstd) per
// input row so that a consumer can re-evaluate the normalisation in fp32 (rela_merge).
// Optional second source (fp16 rows x2, rows2 per sample): sample b's output rows are [rows_in rows of x | rows2 rows of x2],
// i.e. the [x ; objs] concatenation of GatedSelfAttentionDense (attention.py:230) is normalised in ONE launch.
// YLO: the fp16 rows are followed, C columns to the right, by their fp16 residuals lo = fp16(y - fp16(y)) -- the [hi | lo] operand of a
// split-fp16 projection (strict mode); its own instantiation, the plain form compiles unchanged.
template <bool XF32, int NV, bool YF32 = false, bool X2F32 = false, bool YLO = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ xv, int ldx, half_t* __restrict__ y,
                                                        int ldy, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int nrows, int rows_in,
                                                        int rows_out, int row_off, int C, float eps,
                                                        float* __restrict__ stats, const half_t* __restrict__ x2, int ldx2, int rows2) {
    const int lane = threadIdx.x & 63;
    const int orow = blockIdx.x * 4 + (threadIdx.x >> 6);        // index over B * (rows_in + rows2)
    if (orow >= nrows) return;
    const int rtot = rows_in + rows2;
    const int bidx = orow / rtot;
    const int i_in = orow - bidx * rtot;
    const bool second = i_in >= rows_in;                           // wave-uniform
    const int row = second ? bidx * rows2 + (i_in - rows_in) : bidx * rows_in + i_in;   // row within its source
    const int nvec = C / 8;
    float v[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vec = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = 0.0f;
        if (vec < nvec) {
            if (second) {
                if constexpr (X2F32) {     // fp32 second source (its own instantiation: the fp16 form compiles unchanged)
                    const float* xr = reinterpret_cast<const float*>(x2) + (size_t)row * ldx2 + vec * 8;
                    const float4 a = *reinterpret_cast<const float4*>(xr);
                    const float4 c = *reinterpret_cast<const float4*>(xr + 4);
                    v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w;
                    v[i][4] = c.x; v[i][5] = c.y; v[i][6] = c.z; v[i][7] = c.w;
                } else {
                    uint4 raw = ld16(x2 + (size_t)row * ldx2 + vec * 8);
                    const half8_t hv = *reinterpret_cast<half8_t*>(&raw);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[i][j] = (float)hv[j];
                }
            } else if constexpr (XF32) {
                const float* xr = reinterpret_cast<const float*>(xv) + (size_t)row * ldx + vec * 8;
                const float4 a = *reinterpret_cast<const float4*>(xr);
                const float4 c = *reinterpret_cast<const float4*>(xr + 4);
                v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w;
                v[i][4] = c.x; v[i][5] = c.y; v[i][6] = c.z; v[i][7] = c.w;
            } else {
                uint4 raw = ld16(reinterpret_cast<const half_t*>(xv) + (size_t)row * ldx + vec * 8);
                const half8_t hv = *reinterpret_cast<half8_t*>(&raw);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = (float)hv[j];
            }
        }
    }
    float a = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a += v[i][j];              // lanes past the row hold zeros
    const float mean = wave_sum(a) / (float)C;
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dlt = v[i][j] - mean;
                ss = fmaf(dlt, dlt, ss);
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
    if (stats != nullptr && lane == 0 && !second) {
        stats[(size_t)row * 2] = mean;
        stats[(size_t)row * 2 + 1] = rstd;
    }
    half_t* yr = y + ((size_t)bidx * rows_out + row_off + i_in) * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vec = lane + 64 * i;
        if (vec < nvec) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + vec * 8);
            const float4 g1 = *reinterpret_cast<const float4*>(gamma + vec * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + vec * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(beta + vec * 8 + 4);
            const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            if constexpr (YF32) {     // fp32 rows (an encoder's last_hidden_state leaves the library unrounded); a separate instantiation, so
                                      // that the fp16 form below compiles exactly as before (rela_merge_ln_kernel must round identically)
                float* yf = reinterpret_cast<float*>(y) + ((size_t)bidx * rows_out + row_off + i_in) * ldy + vec * 8;
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaf((v[i][j] - mean) * rstd, g[j], bt[j]);
                *reinterpret_cast<float4*>(yf) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(yf + 4) = make_float4(o[4], o[5], o[6], o[7]);
            } else {
                if constexpr (YLO) {
                    half8_t ov, lv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float yv = pin_value(fmaf((v[i][j] - mean) * rstd, g[j], bt[j]));      // hi and lo from the same value
                        ov[j] = (half_t)yv;
                        lv[j] = (half_t)(yv - (float)ov[j]);
                    }
                    st16(yr + vec * 8, *reinterpret_cast<uint4*>(&ov));
                    st16(yr + C + vec * 8, *reinterpret_cast<uint4*>(&lv));
                } else {
                    half8_t ov;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (half_t)fmaf((v[i][j] - mean) * rstd, g[j], bt[j]);   // explicit: rela_merge_ln_kernel rounds identically
                    st16(yr + vec * 8, *reinterpret_cast<uint4*>(&ov));
                }
            }
        }
    }
}

// per-row (mean, rstd) only, of fp32 rows: the statistics half of layernorm_kernel<true, NV> (same loads, same two-pass sums, same
// wave reductions -> the same bits), for consumers that re-evaluate the normalisation in fp32 themselves (rela_pool_ln3, rela_merge)
template <int NV>
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, int ldx, int nrows, int C, float eps, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int nvec = C / 8;
    float v[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vec = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = 0.0f;
        if (vec < nvec) {
            const float* xr = x + (size_t)row * ldx + vec * 8;
            const float4 a = *reinterpret_cast<const float4*>(xr);
            const float4 c = *reinterpret_cast<const float4*>(xr + 4);
            v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w;
            v[i][4] = c.x; v[i][5] = c.y; v[i][6] = c.z; v[i][7] = c.w;
        }
    }
    float a = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a += v[i][j];
    const float mean = wave_sum(a) / (float)C;
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dlt = v[i][j] - mean;
                ss = fmaf(dlt, dlt, ss);
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
    if (lane == 0) {
        stats[(size_t)row * 2] = mean;
        stats[(size_t)row * 2 + 1] = rstd;
    }
}

#define g_gn_ppb gl_opt(16)  // default 16;  // GroupNorm apply: pixels per pixel-lane per block (A/B knob 16)
#define g_gn_fused gl_opt(17)  // default 1; // A/B knob 17: single-launch GroupNorm for small maps

}  // namespace

static int gn_stats_launch(const gl_gn_args& a, hipStream_t st) {
    const int C2 = a.x2 ? a.C2 : 0;
    const int C = a.C1 + C2;
    if (!a.x1 || !a.partial || C <= 0 || C > GN_MAX_C || (C % 32) || (a.C1 % 8) || (C2 % 8) || a.nchunk <= 0 || a.nchunk > a.HW)
        return GL_ERR_BAD_ARG;
    if (a.x_f32) gn_stats_kernel<true><<<dim3(a.nchunk, a.B), dim3(256), 0, st>>>(a.x1, a.C1, a.x2, C2, a.HW, a.nchunk, a.partial);
    else gn_stats_kernel<false><<<dim3(a.nchunk, a.B), dim3(256), 0, st>>>(a.x1, a.C1, a.x2, C2, a.HW, a.nchunk, a.partial);
    GL_CHECK_LAUNCH();
    return 0;
}

static inline GnOut gn_out(const gl_gn_args& a, int C) {
    GnOut o;
    o.out = reinterpret_cast<half_t*>(a.out);
    o.out_lo = reinterpret_cast<half_t*>(a.out_lo);
    o.ldo = a.ldo > 0 ? a.ldo : C;
    o.raw = reinterpret_cast<half_t*>(a.raw);
    o.ldraw = a.ldraw;
    return o;
}

static int gn_apply_launch(const gl_gn_args& a, hipStream_t st) {
    const int C2 = a.x2 ? a.C2 : 0;
    const int C = a.C1 + C2;
    if (!a.x1 || !a.partial || !a.gamma || !a.beta || !a.out || C <= 0 || (C % 32) || (a.C1 % 8) || (C2 % 8)) return GL_ERR_BAD_ARG;
    // pixels per block: >= 2 pixels per pixel-lane, ~1-2k blocks at the 64x64 level
    const int nvec = C / 8;
    const int nplanes = 256 / (nvec < 256 ? nvec : 256);
    int ppb = g_gn_ppb * nplanes;
    if (ppb < 16) ppb = 16;
    if (ppb > a.HW) ppb = a.HW;
    const int nblk = gl_cdiv(a.HW, ppb);
    const GnOut o = gn_out(a, C);
    if (a.x_f32)
        gn_apply_kernel<true><<<dim3(nblk, a.B), dim3(256), 0, st>>>(a.x1, a.C1, a.x2, C2, a.HW, a.nchunk, a.partial, a.gamma, a.beta, a.eps, a.silu, o, ppb);
    else
        gn_apply_kernel<false><<<dim3(nblk, a.B), dim3(256), 0, st>>>(a.x1, a.C1, a.x2, C2, a.HW, a.nchunk, a.partial, a.gamma, a.beta, a.eps, a.silu, o, ppb);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_groupnorm_stats(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t B, int32_t HW,
                                  float* partial, int32_t nchunk, void* stream) {
    gl_gn_args a{};
    a.x1 = x1; a.C1 = C1; a.x2 = x2; a.C2 = C2; a.B = B; a.HW = HW; a.partial = partial; a.nchunk = nchunk;
    return gn_stats_launch(a, (hipStream_t)stream);
}

extern "C" int gl_groupnorm_apply(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t B, int32_t HW,
                                  const float* partial, int32_t nchunk, const float* gamma, const float* beta,
                                  float eps, int32_t silu, void* out, void* stream) {
    gl_gn_args a{};
    a.x1 = x1; a.C1 = C1; a.x2 = x2; a.C2 = C2; a.B = B; a.HW = HW; a.partial = const_cast<float*>(partial); a.nchunk = nchunk;
    a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu; a.out = out;
    return gn_apply_launch(a, (hipStream_t)stream);
}

// bundle geometry of gn_bundle_kernel: groups per bundle (1, 2 or 4) such that the bundle is whole 8-channel vectors
static inline int gn_bundle_groups(int C) {
    const int cpg = C / 32;
    if (cpg % 8 == 0) return 1;
    if ((2 * cpg) % 8 == 0) return 2;
    if ((4 * cpg) % 8 == 0) return 4;
    return 0;
}

// how many launches gl_groupnorm issues for this shape: 1 (a register-resident single-launch kernel) or 2 (statistics +
// apply).  fp32 inputs hold twice the registers per vector, so the bundle form covers half the slab.
extern "C" int gl_groupnorm_launches_ex(int32_t C, int32_t HW, int32_t x_f32) {
    if (!g_gn_fused) return 2;
    if ((C % 256) == 0) {
        const int64_t vecs = (int64_t)HW * (C / 256);
        if (vecs <= 256 * 10) return 1;
    }
    if (g_gn_fused >= 2) return 2;                      // A/B: 2 = the small-map kernel only
    const int gb = gn_bundle_groups(C);
    if (gb == 0 || C / 32 < 8) return 2;                // a vector must span at most two groups
    const int nvp = (C / 32) * gb / 8;                   // 5, 10 or 15 vectors per pixel
    if (nvp != 5 && nvp != 10 && nvp != 15) return 2;
    // one block moves the whole HW x bundle slab: measured faster than the two launches up to 80 KB per block (640 ch @ 32x32:
    // 17.4 -> 15.0 us, 640 ch @ 16x16: 15.5 -> 8.1 us), slower beyond (320 ch @ 64x64 = 320 KB per block on 64 blocks: 22.8 -> 43 us)
    if ((int64_t)HW * nvp * 16 > 80 * 1024) return 2;
    return gl_cdiv(HW, 960 / nvp) <= (x_f32 ? 8 : 22) ? 1 : 2;        // and the slab fits the registers of one 960-thread block
}
extern "C" int gl_groupnorm_launches(int32_t C, int32_t HW) { return gl_groupnorm_launches_ex(C, HW, 0); }

template <bool XF32>
static int gn_single_launch(const gl_gn_args& a, int C, bool small_map, hipStream_t st) {
    const int C2 = a.x2 ? a.C2 : 0;
    const GnOut o = gn_out(a, C);
    if (small_map) {
        const int64_t vecs = (int64_t)a.HW * (C / 256);
        const dim3 grid(32, a.B), blk(256);
        if (vecs <= 256 * 3) gn_fused_kernel<XF32, 3><<<grid, blk, 0, st>>>(a.x1, a.C1, a.x2, C2, a.HW, a.gamma, a.beta, a.eps, a.silu, o);
        else if (vecs <= 256 * 5) gn_fused_kernel<XF32, 5><<<grid, blk, 0, st>>>(a.x1, a.C1, a.x2, C2, a.HW, a.gamma, a.beta, a.eps, a.silu, o);
        else gn_fused_kernel<XF32, 10><<<grid, blk, 0, st>>>(a.x1, a.C1, a.x2, C2, a.HW, a.gamma, a.beta, a.eps, a.silu, o);
        GL_CHECK_LAUNCH();
        return 0;
    }
    const int cpg = C / 32, gb = gn_bundle_groups(C);
    const dim3 grid(32 / gb, a.B);
#define GL_GNB(V, T) gn_bundle_kernel<XF32, V, T><<<grid, dim3(T), 0, st>>>(a.x1, a.C1, a.x2, C2, a.HW, cpg, gb, a.gamma, a.beta, a.eps, a.silu, o)
    // block sizes that are whole multiples of the vectors per pixel (5, 10 or 15): 320 = 5 waves, 960 = 15 waves
    const int nvp = cpg * gb / 8;
    if (nvp == 15) {
        const int passes = gl_cdiv(a.HW, 960 / 15);
        if (passes <= 4) GL_GNB(4, 960); else if (passes <= 8) GL_GNB(8, 960);
        else if constexpr (XF32) return GL_ERR_UNSUPPORTED;
        else if (passes <= 16) GL_GNB(16, 960); else GL_GNB(22, 960);
    } else {
        const int p320 = gl_cdiv(a.HW, 320 / nvp);
        const int p960 = gl_cdiv(a.HW, 960 / nvp);
        if (p320 <= 4) GL_GNB(4, 320); else if (p320 <= 8) GL_GNB(8, 320);
        else if (p960 <= 8) GL_GNB(8, 960);
        else if constexpr (XF32) return GL_ERR_UNSUPPORTED;
        else if (p960 <= 16) GL_GNB(16, 960); else GL_GNB(22, 960);
    }
#undef GL_GNB
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_groupnorm_ex(const gl_gn_args* ap, void* stream) {
    if (!ap) return GL_ERR_BAD_ARG;
    const gl_gn_args& a = *ap;
    const int C2 = a.x2 ? a.C2 : 0;
    const int C = a.C1 + C2;
    if (!a.x1 || !a.gamma || !a.beta || !a.out || C <= 0 || (C % 32) || (a.C1 % 8) || (C2 % 8) || a.B <= 0 || a.HW <= 0) return GL_ERR_BAD_ARG;
    const int ldo = a.ldo > 0 ? a.ldo : C;
    if (ldo < C || (ldo % 8) != 0) return GL_ERR_BAD_ARG;
    if (a.raw != nullptr && (a.ldraw < 2 * C || (a.ldraw % 8) != 0)) return GL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool small_map = g_gn_fused && (C % 256) == 0 && (int64_t)a.HW * (C / 256) <= 256 * 10;
    if (small_map || gl_groupnorm_launches_ex(C, a.HW, a.x_f32) == 1)
        return a.x_f32 ? gn_single_launch<true>(a, C, small_map, st) : gn_single_launch<false>(a, C, small_map, st);
    if (!a.partial || a.nchunk <= 0) return GL_ERR_BAD_ARG;
    const int rc = gn_stats_launch(a, st);
    if (rc != 0) return rc;
    return gn_apply_launch(a, st);
}

extern "C" int gl_groupnorm(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t B, int32_t HW, const float* gamma,
                            const float* beta, float eps, int32_t silu, void* out, float* partial, int32_t nchunk, void* stream) {
    gl_gn_args a{};
    a.x1 = x1; a.C1 = C1; a.x2 = x2; a.C2 = C2; a.B = B; a.HW = HW; a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu;
    a.out = out; a.partial = partial; a.nchunk = nchunk;
    return gl_groupnorm_ex(&a, stream);
}

extern "C" int gl_layernorm(const void* x, int32_t ldx, int32_t x_f32, void* y, int32_t ldy, const float* gamma,
                            const float* beta, int32_t B, int32_t rows_in, int32_t rows_out, int32_t row_off, int32_t C,
                            float eps, float* stats, const void* x2, int32_t ldx2, int32_t rows2, void* stream) {
    if (!x || !y || !gamma || !beta || C <= 0 || (C % 8) || C > 2048 || (ldx % 8) || (ldy % 8)) return GL_ERR_BAD_ARG;
    if (x2 == nullptr) rows2 = 0;
    if (rows2 < 0 || (x2 != nullptr && (ldx2 % 8))) return GL_ERR_BAD_ARG;
    const half_t* x2p = reinterpret_cast<const half_t*>(x2);
    const int nrows = B * (rows_in + rows2);
    if (nrows <= 0) return GL_ERR_BAD_ARG;
    half_t* yp = reinterpret_cast<half_t*>(y);
    hipStream_t st = (hipStream_t)stream;
    const int nv = gl_cdiv(C / 8, 64);
    const dim3 grid(gl_cdiv(nrows, 4)), blk(256);
    const int y_f32 = (x_f32 >> 1) & 1, x2_f32 = (x_f32 >> 2) & 1, y_lo = (x_f32 >> 3) & 1;
    x_f32 &= 1;
    if (y_lo) {
        // [hi | lo] fp16 rows out of the fp32 stream (optionally with the fp32 second source): ldy covers both halves
        if (!x_f32 || y_f32 || ldy < 2 * C || (x2 != nullptr && !x2_f32)) return GL_ERR_UNSUPPORTED;
#define GL_LNL(V) layernorm_kernel<true, V, false, false, true><<<grid, blk, 0, st>>>(x, ldx, yp, ldy, gamma, beta, nrows, rows_in, rows_out, row_off, C, eps, stats, x2p, ldx2, rows2)
#define GL_LNL2(V) layernorm_kernel<true, V, false, true, true><<<grid, blk, 0, st>>>(x, ldx, yp, ldy, gamma, beta, nrows, rows_in, rows_out, row_off, C, eps, stats, x2p, ldx2, rows2)
        if (x2 != nullptr) { if (nv == 1) GL_LNL2(1); else if (nv == 2) GL_LNL2(2); else if (nv == 3) GL_LNL2(3); else GL_LNL2(4); }
        else { if (nv == 1) GL_LNL(1); else if (nv == 2) GL_LNL(2); else if (nv == 3) GL_LNL(3); else GL_LNL(4); }
#undef GL_LNL
#undef GL_LNL2
        GL_CHECK_LAUNCH();
        return 0;
    }
#define GL_LN(F, V) layernorm_kernel<F, V><<<grid, blk, 0, st>>>(x, ldx, yp, ldy, gamma, beta, nrows, rows_in, rows_out, row_off, C, eps, stats, x2p, ldx2, rows2)
#define GL_LNF(V) layernorm_kernel<true, V, true><<<grid, blk, 0, st>>>(x, ldx, yp, ldy, gamma, beta, nrows, rows_in, rows_out, row_off, C, eps, stats, x2p, ldx2, rows2)
#define GL_LN2(V) layernorm_kernel<true, V, false, true><<<grid, blk, 0, st>>>(x, ldx, yp, ldy, gamma, beta, nrows, rows_in, rows_out, row_off, C, eps, stats, x2p, ldx2, rows2)
    if (x2_f32 && x2 != nullptr) {
        if (!x_f32 || y_f32 || (ldx2 % 4) != 0) return GL_ERR_UNSUPPORTED;   // fp32 [x ; x2] rows -> fp16 (the fuser's LayerNorm)
        if (nv == 1) GL_LN2(1); else if (nv == 2) GL_LN2(2); else if (nv == 3) GL_LN2(3); else GL_LN2(4);
    } else if (y_f32) {
        if (!x_f32) return GL_ERR_UNSUPPORTED;            // fp32 rows out of an fp32 stream only
        if (nv == 1) GL_LNF(1); else if (nv == 2) GL_LNF(2); else if (nv == 3) GL_LNF(3); else GL_LNF(4);
    } else if (x_f32) {
        if (nv == 1) GL_LN(true, 1); else if (nv == 2) GL_LN(true, 2); else if (nv == 3) GL_LN(true, 3); else GL_LN(true, 4);
    } else {
        if (nv == 1) GL_LN(false, 1); else if (nv == 2) GL_LN(false, 2); else if (nv == 3) GL_LN(false, 3); else GL_LN(false, 4);
    }
#undef GL_LN
#undef GL_LNF
#undef GL_LN2
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_layernorm_stats(const float* x, int32_t ldx, int32_t rows, int32_t C, float eps, float* stats, void* stream) {
    if (!x || !stats || rows <= 0 || C <= 0 || (C % 8) || C > 2048 || (ldx % 4)) return GL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nv = gl_cdiv(C / 8, 64);
    const dim3 grid(gl_cdiv(rows, 4)), blk(256);
    if (nv == 1) ln_stats_kernel<1><<<grid, blk, 0, st>>>(x, ldx, rows, C, eps, stats);
    else if (nv == 2) ln_stats_kernel<2><<<grid, blk, 0, st>>>(x, ldx, rows, C, eps, stats);
    else if (nv == 3) ln_stats_kernel<3><<<grid, blk, 0, st>>>(x, ldx, rows, C, eps, stats);
    else ln_stats_kernel<4><<<grid, blk, 0, st>>>(x, ldx, rows, C, eps, stats);
    GL_CHECK_LAUNCH();
    return 0;
}
