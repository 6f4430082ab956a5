// Image preprocessing of the reward stage (reference models/policy.py:108-111 calls the HuggingFace CLIP
// feature extractor on PIL images): decoded fp32 image -> uint8 (the arithmetic of GLIGEN/interface.py:543-547), Pillow's
// BICUBIC resize (shortest edge -> 224), centre crop, (x / 255 - mean) / std, channels first -- on the GPU, so the rollout's
// images never leave HBM between the VAE decoder and the CLIP vision tower.
//
// The resize is Pillow's own algorithm, bit for bit (src/libImaging/Resample.c, Pillow 9-12): two separable passes over
// 8-bit data, horizontal first, each output sample = clip8((2^21 + sum_k pixel[xmin + k] * coeff[k]) >> 22) with 22-bit
// fixed-point coefficients; the intermediate image is rounded to 8 bits like Pillow's.  The coefficient tables (a few KB,
// float64 arithmetic in Pillow's operation order) are built on the host (layoutllm_t2i_amd/preprocess.py) and passed in.
// Byte / integer work, HBM- and latency-bound (12.6 MB in, 9.6 MB out for 16 images): one thread per output pixel, the
// three channels of a pixel in one thread, neighbouring threads on neighbouring addresses.
#include "common.h"
#include "gligen_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Pillow: 22

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;                       // arithmetic shift, as Pillow's lookup index
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void image_to_u8_kernel(const float* __restrict__ img, int B, int HW, uint8_t* __restrict__ out) {
    // interface.py:543-547: clamp(x, -1, 1) * 0.5 + 0.5 (torch fp32, two roundings), then numpy fp32 * 255, astype(uint8) = truncation
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const size_t b = i / HW, p = i - b * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = img[(b * 3 + c) * HW + p];
        v = fminf(fmaxf(v, -1.0f), 1.0f);
        v = __fadd_rn(__fmul_rn(v, 0.5f), 0.5f);
        v = __fmul_rn(v, 255.0f);
        out[i * 3 + c] = (uint8_t)(int)v;       // v in [0, 255]: C truncation == numpy astype(uint8)
    }
}

__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ in, int B, int H, int W, const int* __restrict__ bounds,
                                                         const int* __restrict__ coeffs, int ksize, int Wout, uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (b, y, xx)
    if (i >= (size_t)B * H * Wout) return;
    const int xx = (int)(i % Wout);
    const size_t row = i / Wout;                                         // b * H + y
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = coeffs + (size_t)xx * ksize;
    const uint8_t* src = in + (row * W + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int w = k[x];
        s0 += (int)src[3 * x] * w;
        s1 += (int)src[3 * x + 1] * w;
        s2 += (int)src[3 * x + 2] * w;
    }
    uint8_t* dst = out + i * 3;
    dst[0] = (uint8_t)clip8(s0);
    dst[1] = (uint8_t)clip8(s1);
    dst[2] = (uint8_t)clip8(s2);
}

struct Norm3 { float mean[3], std[3]; };

__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ in, int B, int H, int W, const int* __restrict__ bounds,
                                                              const int* __restrict__ coeffs, int ksize, int top, int left, int ch, int cw,
                                                              Norm3 nm, float* __restrict__ out, uint8_t* __restrict__ out_u8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (b, cy, cx) inside the crop window
    if (i >= (size_t)B * ch * cw) return;
    const int cx = (int)(i % cw);
    const size_t t = i / cw;
    const int cy = (int)(t % ch);
    const size_t b = t / ch;
    const int yy = top + cy, xx = left + cx;
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = coeffs + (size_t)yy * ksize;
    const uint8_t* src = in + ((b * H + ymin) * W + xx) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
        const int w = k[y];
        const uint8_t* p = src + (size_t)y * W * 3;
        s0 += (int)p[0] * w;
        s1 += (int)p[1] * w;
        s2 += (int)p[2] * w;
    }
    const int v[3] = {clip8(s0), clip8(s1), clip8(s2)};
    if (out_u8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out_u8[i * 3 + c] = (uint8_t)v[c];
    }
    if (out) {
        // transformers 4.19.2 image_utils: float32(u8) / 255.0 (fp32 division), then (x - mean) / std in fp32
        const size_t plane = (size_t)ch * cw;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = __fdiv_rn((float)v[c], 255.0f);
            out[(b * 3 + c) * plane + (size_t)cy * cw + cx] = __fdiv_rn(__fsub_rn(x, nm.mean[c]), nm.std[c]);
        }
    }
}

}  // namespace

extern "C" int gl_image_to_u8(const float* img_nchw, int32_t B, int32_t H, int32_t W, uint8_t* out_hwc, void* stream) {
    if (!img_nchw || !out_hwc || B <= 0 || H <= 0 || W <= 0) return GL_ERR_BAD_ARG;
    const size_t n = (size_t)B * H * W;
    image_to_u8_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(img_nchw, B, H * W, out_hwc);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_resample_h_u8(const uint8_t* in, int32_t B, int32_t H, int32_t W, const int32_t* bounds, const int32_t* coeffs, int32_t ksize,
                                int32_t Wout, uint8_t* out, void* stream) {
    if (!in || !bounds || !coeffs || !out || B <= 0 || H <= 0 || W <= 0 || Wout <= 0 || ksize <= 0) return GL_ERR_BAD_ARG;
    const size_t n = (size_t)B * H * Wout;
    resample_h_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(in, B, H, W, bounds, coeffs, ksize, Wout, out);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_resample_v_norm(const uint8_t* in, int32_t B, int32_t H, int32_t W, const int32_t* bounds, const int32_t* coeffs, int32_t ksize,
                                  int32_t Hout, int32_t top, int32_t left, int32_t crop_h, int32_t crop_w, const float* mean3, const float* std3,
                                  float* out_nchw, uint8_t* out_u8_hwc, void* stream) {
    if (!in || !bounds || !coeffs || (!out_nchw && !out_u8_hwc) || B <= 0 || H <= 0 || W <= 0 || Hout <= 0 || ksize <= 0) return GL_ERR_BAD_ARG;
    if (top < 0 || left < 0 || crop_h <= 0 || crop_w <= 0 || top + crop_h > Hout || left + crop_w > W) return GL_ERR_BAD_ARG;
    if (out_nchw && (!mean3 || !std3)) return GL_ERR_BAD_ARG;
    Norm3 nm{{0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}};
    if (mean3 && std3)
        for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.std[c] = std3[c]; }      // HOST pointers: six floats by value
    const size_t n = (size_t)B * crop_h * crop_w;
    resample_v_norm_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(in, B, H, W, bounds, coeffs, ksize, top, left,
                                                                                                  crop_h, crop_w, nm, out_nchw, out_u8_hwc);
    GL_CHECK_LAUNCH();
    return 0;
}
