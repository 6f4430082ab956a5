// Whole-chip issue-rate probe (wall clock, HIP events): every SIMD of the 256 CUs runs W waves of ONE instruction kind;
// prints ns per wave-instruction per SIMD and the ratio to v_mfma_f32_32x32x16_f16 (32 clocks at the dense fp16 peak).
// Complements isa_rate.hip, whose s_memtime ticks could not be tied to core clocks.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/isa_rate2 tools/probes/isa_rate2.hip && /tmp/isa_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, float seed, int iters) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = seed + 0.001f * (float)(threadIdx.x + i);
    float p0 = seed, p1 = seed * 0.5f;
    unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    half8_t ha, hb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(seed + i); hb[i] = (_Float16)(seed - i); }
    f32x16 acc[2];
    f32x4 acc4[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (KIND == 0) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(r[i]), "v"(p0));
                REP8(X)
#undef X
            } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(p0), "v"(p1));
                REP8(X)
#undef X
            } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(p0), "v"(p1));
                REP8(X)
#undef X
            } else if constexpr (KIND == 4) {
                // 8 x mfma 32x32x16 on two accumulators
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[1], 0, 0, 0);
                }
            } else if constexpr (KIND == 5) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    acc4[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[0], 0, 0, 0);
                    acc4[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[1], 0, 0, 0);
                    acc4[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[2], 0, 0, 0);
                    acc4[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[3], 0, 0, 0);
                }
            } else if constexpr (KIND == 6) {
                // mixed: 4 mfma 32x32x16 + 8 v_exp per group (do they overlap across waves?)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[1], 0, 0, 0);
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == 7) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 1) & 7]));
                REP8(X)
#undef X
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i] + (float)pk[i];
    s += acc[0][0] + acc[1][3] + acc4[0][0] + acc4[1][1] + acc4[2][2] + acc4[3][3];
    if (s == 123.456f) out[0] = s;
}

template <int KIND>
void run(const char* name, int instr_per_u, double* ref) {
    float* d;
    (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;             // 256-thread blocks: one wave per SIMD each; wps blocks per CU
        const int iters = 4096 / wps;
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, 0.37f, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms < best) best = ms;
        }
        const double per_simd = (double)wps * iters * 8 * instr_per_u;       // wave-instructions issued on one SIMD
        const double ns = best * 1e6 / per_simd;
        if (KIND == 4 && wps == 4) *ref = ns;
        printf("%-34s waves/SIMD %d: %8.3f ms  -> %7.3f ns per wave-instruction per SIMD%s\n", name, wps, best, ns, "");
    }
    (void)hipFree(d);
}

int main() {
    double ref = 0;
    run<4>("v_mfma_f32_32x32x16_f16", 8, &ref);
    printf("# 32x32x16 at 4 waves/SIMD = %.3f ns = 32 clocks at the dense peak -> implied clock %.2f GHz; rows below in those clocks\n", ref, 32.0 / ref);
    run<5>("v_mfma_f32_16x16x32_f16", 8, &ref);
    run<2>("v_fma_f32", 8, &ref);
    run<3>("v_max3_f32", 8, &ref);
    run<0>("v_exp_f32", 8, &ref);
    run<1>("v_cvt_pkrtz_f16_f32", 8, &ref);
    run<7>("v_permlane32_swap", 8, &ref);
    run<6>("4 mfma32 + 8 v_exp (12 instr)", 12, &ref);
    return 0;
}
