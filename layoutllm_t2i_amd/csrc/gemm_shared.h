// Pieces shared by the two GEMM / implicit-GEMM-conv translation units (gemm_conv.hip: 4-wave 2-stage kernels, skinny
// kernel, dispatch; gemm8.hip: 8-wave deep-pipelined 256-row kernel).  Device code is compiled per translation unit
// (no relocatable device code), so everything here is inline / static.
#pragma once
#include "common.h"
#include "gligen_hip.h"

// 16 zero bytes in global memory: the source of masked lanes of the direct-to-LDS loads (one copy per translation unit).
static __device__ uint4 g_zero16[4];

struct ConvGeom {
    const half_t* in;
    int B, Hin, Win, Cin, Hout, Wout, stride, ups;
    const half_t* zero;        // gemm8.hip: device address of 16 zero bytes (filled by its launcher)
    int cwrap;                 // split-fp16 input with a third pass (gl_conv_args.in_split == 3): channel blocks >= cwrap read the input's block
                               // (cblk - cwrap), i.e. the K walk [hi | lo | hi] over pixel rows that hold [hi | lo]; 0 = no wrap.  Cin above is the
                               // pixel ROW STRIDE in channels (2 x the conv's channels for a split input); the channel blocks walked are K / 576
};

namespace {

__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void glds16(const half_t* src, half_t* dst) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(src)),
        reinterpret_cast<__attribute__((address_space(3))) void*>(reinterpret_cast<uintptr_t>(dst)), 16, 0, 0);
}

// One 8-column piece of one output row: bias / activation / residual, then the stores.  Shared by the GEMM epilogues
// and the split-K reduction so that all of them produce bit-identical results from the same fp32 sums.
// Two halves, so that a caller with several pieces in hand can issue ALL their global loads before the first dependent
// store (the stores may alias the loads as far as the compiler knows, which otherwise serialises load -> store -> load).
struct Fin8Aux {
    float b[8];     // bias (zeros when absent)
    float r[8];     // residual or row bias, as fp32
};

__device__ __forceinline__ void fin8_load(const gl_gemm_args& p, int m, int n, Fin8Aux& a) {
    const int epi = p.epi;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a.b[j] = 0.0f; a.r[j] = 0.0f; }
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        a.b[0] = b0.x; a.b[1] = b0.y; a.b[2] = b0.z; a.b[3] = b0.w; a.b[4] = b1.x; a.b[5] = b1.y; a.b[6] = b1.z; a.b[7] = b1.w;
    }
    if (epi == GL_EPI_RES || epi == GL_EPI_GATE_RES) {
        if (p.res_f32) {
            const float* rp = reinterpret_cast<const float*>(p.res) + (size_t)m * p.ldres + n;
            const float4 r0 = *reinterpret_cast<const float4*>(rp);
            const float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
            a.r[0] = r0.x; a.r[1] = r0.y; a.r[2] = r0.z; a.r[3] = r0.w; a.r[4] = r1.x; a.r[5] = r1.y; a.r[6] = r1.z; a.r[7] = r1.w;
        } else {
            uint4 raw = ld16(reinterpret_cast<const half_t*>(p.res) + (size_t)m * p.ldres + n);
            const half8_t rv = *reinterpret_cast<half8_t*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) a.r[j] = (float)rv[j];
        }
    } else if (epi == GL_EPI_ROWBIAS) {
        const int sidx = m / p.rows_per_sample;
        if (p.rowbias_f32) {
            const float* rp = reinterpret_cast<const float*>(p.rowbias) + (size_t)sidx * p.ld_rowbias + n;
            const float4 r0 = *reinterpret_cast<const float4*>(rp);
            const float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
            a.r[0] = r0.x; a.r[1] = r0.y; a.r[2] = r0.z; a.r[3] = r0.w; a.r[4] = r1.x; a.r[5] = r1.y; a.r[6] = r1.z; a.r[7] = r1.w;
        } else {
            uint4 raw = ld16(reinterpret_cast<const half_t*>(p.rowbias) + (size_t)sidx * p.ld_rowbias + n);
            const half8_t rv = *reinterpret_cast<half8_t*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) a.r[j] = (float)rv[j];
        }
    }
}

__device__ __forceinline__ void fin8_store(const gl_gemm_args& p, float gate, int m, int n, float (&v)[8], const Fin8Aux& a) {
    const int epi = p.epi;
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += a.b[j];
    }
    if (epi == GL_EPI_SILU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
    } else if (epi == GL_EPI_RES || epi == GL_EPI_ROWBIAS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += a.r[j];
    } else if (epi == GL_EPI_GATE_RES) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = a.r[j] + gate * v[j];
    }
    half_t* o16 = reinterpret_cast<half_t*>(p.out);
    int ld16o = p.ldc;
    if (p.out_mode == GL_OUT_F32_ROWMAJOR) {
        float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        o16 = reinterpret_cast<half_t*>(p.out2);
        ld16o = p.ldc2;
        if (o16 == nullptr) return;
    }
    if (p.out_mode == GL_OUT_F16_HILO) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pin_value(v[j]);      // hi and lo from the same value (common.h)
    }
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
    st16(o16 + (size_t)m * ld16o + n, *reinterpret_cast<uint4*>(&o));
    if (p.out_mode == GL_OUT_F16_HILO) {
        // split-fp16 operand for the next 1x1 product: the fp16 residual goes N columns to the right
        half8_t l;
#pragma unroll
        for (int j = 0; j < 8; ++j) l[j] = (half_t)(v[j] - (float)o[j]);
        st16(o16 + (size_t)m * ld16o + p.N + n, *reinterpret_cast<uint4*>(&l));
    }
}

__device__ __forceinline__ void finish8(const gl_gemm_args& p, float gate, int m, int n, float (&v)[8]) {
    Fin8Aux a;
    fin8_load(p, m, n, a);
    fin8_store(p, gate, m, n, v, a);
}

}  // namespace

// gemm8.hip: 8-wave, 256 x {160,128} tile, 3-stage LDS ring filled by LDS-DMA with counted vmcnt and raw s_barrier.
// Returns < 0 (GL_ERR_UNSUPPORTED) when the problem is outside what it implements; the caller then uses the 4-wave kernels.
// zs > 1: fp32 partial tiles go to g.workspace[z][M][N] and the caller runs the reduction.
int gl8_supported(const gl_gemm_args& g, bool conv, int* bn_out);
// s3: the dedicated three-pass main loop for the split-fp16 product (g in its K-walk form, K = 3 * kwrap; kper in 32-wide stages of kwrap)
int gl8_launch(const gl_gemm_args& g, const ConvGeom& cg, bool conv, int bm, int bn, int zs, int kper, int order_m, hipStream_t st, bool s3 = false);
int gl8_init(void);
int gl8_read_stamps(void* dst, int64_t bytes);
