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

// grid (max_objs, B), block 256: thread -> (pixel lane, 8-channel vector)
__global__ __launch_bounds__(256) void rela_pool_kernel(const half_t* __restrict__ hid, int H, int W, int C,
                                                        const int* __restrict__ rects, const int* __restrict__ nvalid,
                                                        const int* __restrict__ poison, int max_objs,
                                                        half_t* __restrict__ feat) {
    __shared__ float lacc[RELA_MAX_C];   // [plane][C] partial sums, nplanes * C <= 2048
    const int i = blockIdx.x;
    const int b = blockIdx.y;
    const int nvec = C / 8;
    half_t* frow = feat + ((size_t)b * max_objs + i) * C;
    if (i >= nvalid[b]) {
        for (int c = threadIdx.x; c < C; c += 256) frow[c] = (half_t)0.0f;
        return;
    }
    const int* r = rects + ((size_t)b * max_objs + i) * 4;
    const int top = r[0], bottom = r[1], left = r[2], right = r[3];
    const int rw = right - left;
    const int npix = (bottom - top) * rw;
    if (npix <= 0) {   // empty python slice: torch.mean of nothing = NaN (attention.py:343)
        for (int c = threadIdx.x; c < C; c += 256) frow[c] = (half_t)__builtin_nanf("");
        return;
    }
    const int vlanes = min(nvec, 256);
    const int nplanes = 256 / vlanes;
    const int plane = threadIdx.x / vlanes;
    const int v0 = threadIdx.x - plane * vlanes;
    if (plane < nplanes) {
        for (int vec = v0; vec < nvec; vec += vlanes) {
            float s[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = 0.0f;
            int q = plane;
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
#pragma unroll
            for (int j = 0; j < 8; ++j) lacc[plane * C + vec * 8 + j] = s[j];
        }
    }
    __syncthreads();
    const float inv = 1.0f / (float)npix;
    const bool bad = poison[b] != 0;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.0f;
        for (int pl = 0; pl < nplanes; ++pl) s += lacc[pl * C + c];
        frow[c] = bad ? (half_t)__builtin_nanf("") : (half_t)(s * inv);
    }
}

// elementwise over [B, H*W, C]: 8 channels per thread
__global__ __launch_bounds__(256) void rela_merge_kernel(const half_t* __restrict__ x, const half_t* __restrict__ hid,
                                                         const half_t* __restrict__ f, int H, int W, int C,
                                                         const int* __restrict__ rects, const int* __restrict__ nvalid,
                                                         const int* __restrict__ poison, int max_objs,
                                                         half_t* __restrict__ y, size_t total) {
    const int nvec = C / 8;
    const int HW = H * W;
    const float inv_mo = 1.0f / (float)max_objs;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t tok = idx / nvec;
        const int vec = (int)(idx - tok * nvec);
        const int b = (int)(tok / HW);
        const int p = (int)(tok - (size_t)b * HW);
        const int py = p / W, px = p - py * W;
        uint4 rx = ld16(x + tok * C + vec * 8);
        uint4 rh = ld16(hid + tok * C + vec * 8);
        const half8_t xv = *reinterpret_cast<half8_t*>(&rx);
        const half8_t hv = *reinterpret_cast<half8_t*>(&rh);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        const int nv = nvalid[b];
        const int* rb = rects + (size_t)b * max_objs * 4;
        for (int i = 0; i < nv; ++i) {
            const int top = rb[4 * i], bottom = rb[4 * i + 1], left = rb[4 * i + 2], right = rb[4 * i + 3];
            if (py >= top && py < bottom && px >= left && px < right) {
                uint4 rf = ld16(f + ((size_t)b * max_objs + i) * C + vec * 8);
                const half8_t fv = *reinterpret_cast<half8_t*>(&rf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)fv[j];
            }
        }
        const bool bad = poison[b] != 0;
        half8_t ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float o = 0.5f * (((float)hv[j] + acc[j] * inv_mo) + (float)xv[j]);
            if (bad) o = __builtin_nanf("");
            ov[j] = (half_t)o;
        }
        st16(y + tok * C + vec * 8, *reinterpret_cast<uint4*>(&ov));
    }
}

}  // namespace

extern "C" int gl_rela_pool(const void* hid, int32_t B, int32_t H, int32_t W, int32_t C, const int32_t* rects,
                            const int32_t* nvalid, const int32_t* poison, int32_t max_objs, void* feat, void* stream) {
    if (!hid || !rects || !nvalid || !poison || !feat || C <= 0 || (C % 8) || C > RELA_MAX_C) return GL_ERR_BAD_ARG;
    rela_pool_kernel<<<dim3(max_objs, B), dim3(256), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const half_t*>(hid), H, W, C, rects, nvalid, poison, max_objs, reinterpret_cast<half_t*>(feat));
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_rela_merge(const void* x, const void* hid, const void* f, int32_t B, int32_t H, int32_t W, int32_t C,
                             const int32_t* rects, const int32_t* nvalid, const int32_t* poison, int32_t max_objs,
                             void* y, void* stream) {
    if (!x || !hid || !f || !rects || !nvalid || !poison || !y || C <= 0 || (C % 8)) return GL_ERR_BAD_ARG;
    const size_t total = (size_t)B * H * W * (C / 8);
    int nblk = (int)((total + 255) / 256);
    if (nblk > 2048) nblk = 2048;
    rela_merge_kernel<<<dim3(nblk), dim3(256), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const half_t*>(x), reinterpret_cast<const half_t*>(hid), reinterpret_cast<const half_t*>(f), H,
        W, C, rects, nvalid, poison, max_objs, reinterpret_cast<half_t*>(y), total);
    GL_CHECK_LAUNCH();
    return 0;
}
