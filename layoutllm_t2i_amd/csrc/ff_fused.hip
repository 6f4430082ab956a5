// Fused FeedForward (attention.py:38-62: GEGLU projection -> erf-GELU gate -> output projection) for the narrow, long
// level of the UNet (C <= 320 channels, tens of thousands of tokens): y = res (+ gate *) (GEGLU(x W1^T + b1) W2^T + b2).
//
// As two GEMM launches this level moves an [M, 4C] intermediate through HBM (84 MB at 64x64x320, written by the GEGLU
// epilogue and read back by the second product) and runs two short-K pipelines (K = 320: five K-tiles per output
// tile, fill and drain dominate).  Here one block owns 128 tokens for the WHOLE layer:
//   * its X rows live in registers as MFMA operand fragments for the lifetime of the block (C/16 x 4 VGPRs per lane);
//   * the hidden dimension is walked in chunks of 32 units: S = X . W1c^T (x tile and gate tile, identical register
//     layouts, so the GEGLU product is formed in registers), H = fp16(S_x * gelu(S_g)) is ALREADY the B operand of the
//     second product (the MFMA output layout of 4 consecutive columns per register group becomes the K order of the
//     next MFMA; W2 fragments are read in the matching order), Y[32 x C] += H . W2c^T accumulates in registers;
//   * only weights stream: W1 chunk (64 packed rows x C, contiguous in the packed layout) + W2 chunk (C rows x 32
//     columns) = 60 KB per chunk through a two-slot LDS ring by LDS-DMA, shared by the four waves; one barrier per chunk.
// Per flop the block pulls 1/128 byte of weights (a 128 x 128 output tile of the two-launch form pulls 1/64) and the
// intermediate never leaves the CU.  One wave per SIMD (256 threads, up to 512 VGPRs: 80 X + 32 S + 160 Y).
//
// Results: phase 1 and the GEGLU arithmetic are those of gl_gemm's GL_EPI_GEGLU epilogue (same k-step partition, same
// bias / gelu_erf_f order), H is rounded to fp16 like the stored intermediate; phase 2 sums each 16-wide k-step in a
// permuted order, so outputs agree with the two-launch form to fp32 rounding of the accumulation, not bit for bit.
#include "common.h"
#include <atomic>
#include "gligen_hip.h"
#include "opts.h"

namespace {

__device__ __forceinline__ void ff_glds16(const half_t* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(src)),
        reinterpret_cast<__attribute__((address_space(3))) void*>(reinterpret_cast<uintptr_t>(dst)), 16, 0, 0);
}

__device__ __forceinline__ f32x16 ff_mfma(half8_t a, half8_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#define g_ff_enable gl_opt(27)  // default 1;   // gl_set_option 27: 0 = gl_ff_fused_applicable answers no (two-launch FeedForward everywhere)

template <int C>
struct FFGeom {
    static constexpr int KS1 = C / 16;               // phase-1 k-steps (K = C)
    static constexpr int NT2 = C / 32;               // phase-2 output tiles of 32 columns per wave
    static constexpr int NCH = 4 * C / 32;           // hidden chunks of 32 units
    static constexpr int ROWCH = C / 8 + 1;          // 16-byte chunks per W1 row in LDS: odd stride -> conflict-free b128 reads
    static constexpr int W1_BYTES = 64 * ROWCH * 16;
    static constexpr int W2_BYTES = C * 64;          // C rows x 32 hidden halfs, chunk-swizzled
    static constexpr int W2_BASE = 2 * W1_BYTES;     // W1 ring: 2 slots; W2 ring: 3 slots (chunk j's W2 is consumed one iteration late)
    static constexpr int B1_BASE = W2_BASE + 3 * W2_BYTES;
    static constexpr int W1_INSTR = ROWCH;           // wave-wide DMA instructions (64 x 16 B) for the W1 part: 64 * ROWCH / 64
    static constexpr int W2_INSTR = C / 16;          // C * 4 chunks / 64
    static constexpr int LDS = B1_BASE + 8 * C * 4;   // rings + the packed GEGLU bias
};

template <int C>
__global__ __launch_bounds__(256, 1) void ff_fused_kernel(gl_ff_args p) {
    using G = FFGeom<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int M = p.M;
    const int m0 = blockIdx.x * 128 + wave * 32;
    int mrow = m0 + ql;
    if (mrow >= M) mrow = M - 1;                       // ragged last block: clamp the loads, skip the stores
    const half_t* w1 = reinterpret_cast<const half_t*>(p.w1);
    const half_t* w2 = reinterpret_cast<const half_t*>(p.w2);

    // ---- weight stream: chunk j -> ring slot j & 1 (LDS-DMA: lane i of a wave instruction lands at dst + 16 i, so the
    // LDS layout is produced by choosing each lane's SOURCE address)
    // per-lane source offsets of this wave's DMA instructions are the same for every chunk: computed once
    constexpr int NI1 = (G::W1_INSTR + 3) / 4, NI2 = (G::W2_INSTR + 3) / 4;
    unsigned off1[NI1], off2[NI2];
#pragma unroll
    for (int ii = 0; ii < NI1; ++ii) {
        const int i = ii * 4 + wave;
        const int pos = i * 64 + lane;                                // chunk position in the padded [64][ROWCH] tile
        const int row = pos / G::ROWCH;
        int c = pos - row * G::ROWCH;
        if (c >= C / 8) c = C / 8 - 1;                                // pad chunk: any valid address
        off1[ii] = (unsigned)(row * C + c * 8);
    }
#pragma unroll
    for (int ii = 0; ii < NI2; ++ii) {
        const int i = ii * 4 + wave;
        const int pos = i * 64 + lane;
        const int n = pos >> 2;
        const int ch = (pos & 3) ^ ((n >> 2) & 3);                    // the chunk stored at this position
        off2[ii] = (unsigned)(n * (4 * C) + ch * 8);
    }
    auto issue = [&](int j) {
        unsigned char* st = smem + (j & 1) * G::W1_BYTES;
        const half_t* w1c = w1 + (size_t)j * 64 * C;                  // 64 packed rows [x 32 | gate 32] of this chunk
#pragma unroll
        for (int ii = 0; ii < NI1; ++ii) {
            const int i = ii * 4 + wave;
            if (i >= G::W1_INSTR) break;
            ff_glds16(w1c + off1[ii], st + i * 1024);
        }
        unsigned char* st2 = smem + G::W2_BASE + (j % 3) * G::W2_BYTES;
        const half_t* w2c = w2 + (size_t)j * 32;                      // columns [32 j, 32 j + 32) of every W2 row
#pragma unroll
        for (int ii = 0; ii < NI2; ++ii) {
            const int i = ii * 4 + wave;
            if (i >= G::W2_INSTR) break;
            ff_glds16(w2c + off2[ii], st2 + i * 1024);
        }
    };
    issue(0);

    // ---- this wave's 32 rows of X as B-operand fragments: lane (row ql, half hi) holds k = 16 s + 8 hi .. + 8
    half8_t xf[G::KS1];
    {
        const half_t* xr = reinterpret_cast<const half_t*>(p.x) + (size_t)mrow * p.ldx + 8 * hi;
#pragma unroll
        for (int s = 0; s < G::KS1; ++s) {
            uint4 raw = ld16(xr + 16 * s);
            xf[s] = *reinterpret_cast<half8_t*>(&raw);
        }
    }
    f32x16 y[G::NT2];
#pragma unroll
    for (int t = 0; t < G::NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[t][r] = 0.0f;

    // the packed GEGLU bias (8C floats) is kept in LDS for the whole block: a per-chunk global load would put one HBM / L2
    // round trip on the critical path of every chunk (one wave per SIMD: nothing else to run meanwhile)
    float* b1s = reinterpret_cast<float*>(smem + G::B1_BASE);
    for (int i = threadIdx.x; i < 8 * C / 4; i += 256)
        *reinterpret_cast<float4*>(b1s + 4 * i) = *reinterpret_cast<const float4*>(p.b1 + 4 * i);
    const float* b1 = b1s;                                     // visible after the first barrier below
    // Loop schedule (one wave per SIMD, so overlap has to come from inside the wave): iteration j runs phase 1 of chunk j,
    // then the erf arithmetic of chunk j (VALU) INTERLEAVED with phase 2 of chunk j - 1 (MFMA, independent of it); the
    // MFMAs execute in the matrix pipe while the wave keeps issuing the GELU instructions.
    constexpr int TG = 2;                                         // output tiles per phase-2 fragment group (NT2 is even)
    constexpr int NTG = G::NT2 / TG;
    half8_t hprev[2];
#pragma unroll
    for (int q = 0; q < 8; ++q) { hprev[0][q] = (half_t)0.0f; hprev[1][q] = (half_t)0.0f; }
    auto load_w2 = [&](const unsigned char* w2s, int tg, uint4 (&wv)[TG][2]) {
#pragma unroll
        for (int q = 0; q < TG; ++q) {
            const int row = (tg * TG + q) * 32 + ql;
            const int sw = (row >> 2) & 3;
            const unsigned char* rp = w2s + row * 64 + 8 * hi;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const uint2 lo = *reinterpret_cast<const uint2*>(rp + (((2 * s2) ^ sw) << 4));
                const uint2 hi2 = *reinterpret_cast<const uint2*>(rp + (((2 * s2 + 1) ^ sw) << 4));
                wv[q][s2] = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
            }
        }
    };
    // phase 2 of one chunk: Y[32 rows x C] += H . W2c^T; k-step s2 carries hidden units {16 s2 + 4 hi + e, 16 s2 + 8 + 4 hi + e}
    auto phase2_group = [&](int tg, const uint4 (&wv)[TG][2], const half8_t (&h)[2]) {
#pragma unroll
        for (int q = 0; q < TG; ++q)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                y[tg * TG + q] = ff_mfma(*reinterpret_cast<const half8_t*>(&wv[q][s2]), h[s2], y[tg * TG + q]);
    };
    int slot3 = 0;                                                // (j - 1) % 3 at the top of iteration j (unused for j = 0)
    for (int j = 0; j < G::NCH; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my share of chunk j has landed
        __syncthreads();                                       // ... everyone's has; W1 slot (j+1)&1 and W2 slot (j+1)%3 are free
        if (j + 1 < G::NCH) issue(j + 1);
        const unsigned char* st = smem + (j & 1) * G::W1_BYTES;
        const unsigned char* w2prev = smem + G::W2_BASE + slot3 * G::W2_BYTES;
        slot3 = (j == 0) ? 0 : (slot3 == 2 ? 0 : slot3 + 1);  // -> j % 3
        // ---- phase 1: S_x / S_g [32 hidden x 32 rows] = W1c . X^T
        f32x16 sx, sg;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sx[r] = 0.0f; sg[r] = 0.0f; }
        const unsigned char* wx = st + (size_t)(ql * G::ROWCH + hi) * 16;
        const unsigned char* wg = wx + (size_t)32 * G::ROWCH * 16;
        // nothing else hides the LDS read latency, so the W1 fragments are fetched a group of k-steps ahead of the MFMAs
        // that use them (two register sets, statically indexed)
        constexpr int GRP = 4;                               // KS1 = C / 16 is a multiple of 4 for every supported width
        constexpr int NG = G::KS1 / GRP;
        half8_t fx[2][GRP], fg[2][GRP];
#pragma unroll
        for (int q = 0; q < GRP; ++q) {
            fx[0][q] = *reinterpret_cast<const half8_t*>(wx + q * 32);
            fg[0][q] = *reinterpret_cast<const half8_t*>(wg + q * 32);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            __builtin_amdgcn_sched_barrier(0);      // keep the read-ahead where it is written (the scheduler folds it back otherwise)
            if (g + 1 < NG) {
#pragma unroll
                for (int q = 0; q < GRP; ++q) {
                    fx[(g + 1) & 1][q] = *reinterpret_cast<const half8_t*>(wx + ((g + 1) * GRP + q) * 32);
                    fg[(g + 1) & 1][q] = *reinterpret_cast<const half8_t*>(wg + ((g + 1) * GRP + q) * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                sx = ff_mfma(fx[g & 1][q], xf[g * GRP + q], sx);
                sg = ff_mfma(fg[g & 1][q], xf[g * GRP + q], sg);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEGLU of chunk j in registers (lane holds hidden units 8 rg + 4 hi + t of this chunk for its row), in the same
        // scheduling region as phase 2 of chunk j - 1: one fragment group is requested ahead, its MFMAs are spread over the
        // erf arithmetic (hprev is all zero on the first iteration: the products add nothing)
        uint4 wv[2][TG][2];
        load_w2(w2prev, 0, wv[0]);
        half8_t hf[2];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 bx = *reinterpret_cast<const float4*>(b1 + j * 64 + 8 * rg + 4 * hi);
            const float4 bg = *reinterpret_cast<const float4*>(b1 + j * 64 + 32 + 8 * rg + 4 * hi);
            const float bxs[4] = {bx.x, bx.y, bx.z, bx.w}, bgs[4] = {bg.x, bg.y, bg.z, bg.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float a = sx[rg * 4 + t] + bxs[t];
                const float b = sg[rg * 4 + t] + bgs[t];
                hf[rg >> 1][(rg & 1) * 4 + t] = (half_t)(a * gelu_erf_f(b));
            }
            // a quarter of the previous chunk's phase 2 after every quarter of the erf work; fragments one group ahead
#pragma unroll
            for (int tg = rg * NTG / 4; tg < (rg + 1) * NTG / 4; ++tg) {
                if (tg + 1 < NTG) load_w2(w2prev, tg + 1, wv[(tg + 1) & 1]);
                phase2_group(tg, wv[tg & 1], hprev);
            }
        }
        hprev[0] = hf[0];
        hprev[1] = hf[1];
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        // phase 2 of the last chunk
        const unsigned char* w2last = smem + G::W2_BASE + ((G::NCH - 1) % 3) * G::W2_BYTES;
        uint4 wva[TG][2];
#pragma unroll
        for (int tg = 0; tg < NTG; ++tg) {
            load_w2(w2last, tg, wva);
            phase2_group(tg, wva, hprev);
        }
    }

    // ---- epilogue: + b2, residual (fp32 stream or fp16), optional gate; lane holds columns 32 t + 8 rg + 4 hi + {0..3}
    const int m = m0 + ql;
    if (m >= M) return;
    float gate = 1.0f;
    if (p.gate != nullptr) gate = p.gate[0];
#pragma unroll
    for (int t = 0; t < G::NT2; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int n = t * 32 + 8 * rg + 4 * hi;
            const float4 bb = *reinterpret_cast<const float4*>(p.b2 + n);
            float v[4] = {y[t][rg * 4] + bb.x, y[t][rg * 4 + 1] + bb.y, y[t][rg * 4 + 2] + bb.z, y[t][rg * 4 + 3] + bb.w};
            float r[4];
            if (p.res_f32) {
                const float4 rr = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + (size_t)m * p.ldres + n);
                r[0] = rr.x; r[1] = rr.y; r[2] = rr.z; r[3] = rr.w;
            } else {
                const half4_t rr = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(p.res) + (size_t)m * p.ldres + n);
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = (float)rr[q];
            }
            if (p.gate != nullptr) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = r[q] + gate * v[q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += r[q];
            }
            if (p.out_mode == GL_OUT_F32_ROWMAJOR) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                half4_t o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (half_t)v[q];
                *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + n) = o;
                if (p.out_mode == GL_OUT_F16_HILO) {        // fp16 residual C columns to the right (split-fp16 operand of proj_out)
                    half4_t l;
#pragma unroll
                    for (int q = 0; q < 4; ++q) l[q] = (half_t)(v[q] - (float)o[q]);
                    *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + C + n) = l;
                }
            }
        }
}

template <int C>
int ff_launch(const gl_ff_args& a, hipStream_t st) {
    using G = FFGeom<C>;
    ff_fused_kernel<C><<<dim3(gl_cdiv(a.M, 128)), dim3(256), G::LDS, st>>>(a);
    GL_CHECK_LAUNCH();
    return 0;
}

}  // namespace

template <int C>
int ff_set_attr() {
    hipError_t e = hipFuncSetAttribute((const void*)ff_fused_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, FFGeom<C>::LDS);
    return e == hipSuccess ? 0 : (int)e;
}

// per-device one-time setup (dynamic LDS above 64 KB), called from gl_init
extern "C" int gl_init_ff(void) {
    int e;
    if ((e = ff_set_attr<64>())) return e;
    if ((e = ff_set_attr<128>())) return e;
    if ((e = ff_set_attr<192>())) return e;
    if ((e = ff_set_attr<256>())) return e;
    return ff_set_attr<320>();
}

extern "C" int gl_ff_fused_supported(int32_t C) { return C == 64 || C == 128 || C == 192 || C == 256 || C == 320; }

// Whether the fused form is the faster one for [M, C] on this device: a block holds 128 rows for the whole layer (long
// blocks, one per CU), so a grid that fills only part of its last round of CUs loses more than the fusion gains
// (288 blocks on 256 CUs = two rounds).  Callers that must agree on the choice (the engine and its Python mirror) ask here.
extern "C" int gl_ff_fused_applicable(int32_t C, int32_t M) {
    if (!g_ff_enable || !gl_ff_fused_supported(C) || M < 128) return 0;
    // CU count of the CURRENT device, cached per device id (relaxed atomics: concurrent first calls compute the same value)
    static std::atomic<int> cu_cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int cus = (dev >= 0 && dev < 64) ? cu_cache[dev].load(std::memory_order_relaxed) : 0;
    if (cus == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (dev >= 0 && dev < 64) cu_cache[dev].store(cus, std::memory_order_relaxed);
    }
    const int blocks = gl_cdiv(M, 128);
    const int rounds = gl_cdiv(blocks, cus);
    return blocks * 10 >= rounds * cus * 9;          // >= 90 % of the CU slots of its rounds
}

extern "C" int gl_ff_fused(const gl_ff_args* a, void* stream) {
    if (!a || !a->x || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->res || !a->out || a->M <= 0) return GL_ERR_BAD_ARG;
    if ((a->ldx % 8) != 0 || (a->ldres % 4) != 0 || (a->ldc % 4) != 0) return GL_ERR_BAD_ARG;
    if (a->out_mode != GL_OUT_F16_ROWMAJOR && a->out_mode != GL_OUT_F32_ROWMAJOR && a->out_mode != GL_OUT_F16_HILO) return GL_ERR_BAD_ARG;
    if (a->out_mode == GL_OUT_F16_HILO && a->ldc < 2 * a->C) return GL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (a->C) {
        case 64: return ff_launch<64>(*a, st);
        case 128: return ff_launch<128>(*a, st);
        case 192: return ff_launch<192>(*a, st);
        case 256: return ff_launch<256>(*a, st);
        case 320: return ff_launch<320>(*a, st);
        default: return GL_ERR_UNSUPPORTED;
    }
}

extern "C" int gl_sizeof_ff_args(void) { return (int)sizeof(gl_ff_args); }
