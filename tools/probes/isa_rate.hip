// Issue-rate micro-benchmark for the VALU / MFMA instructions the d = 40 attention kernel is made of (the
// "206 -> 180 us floor" argument of DESIGN.md rests on v_exp_f32 = 16 clocks per wave instruction; this measures it on gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/isa_rate tools/probes/isa_rate.hip && /tmp/isa_rate
// One block of W waves per SIMD on ONE CU (grid = 1), each wave runs ITER x 64 independent instructions of one kind (8 independent
// register chains, so the result latency is hidden) between two s_memtime reads; clocks per wave instruction = cycles / count, and
// with 2 or 4 waves per SIMD the SIMD-level throughput (clocks per instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
constexpr int ITER = 512;

template <int KIND>
__global__ void probe(unsigned long long* out, float seed) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = seed + 0.001f * (float)(threadIdx.x + i);
    float p0 = seed, p1 = seed * 0.5f;
    half8_t ha, hb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(seed + i); hb[i] = (_Float16)(seed - i); }
    f32x4 acc4[8];
    f32x16 acc16[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
    unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 pkd[8], pkc = f32x2{seed, seed * 0.25f};
#pragma unroll
    for (int i = 0; i < 8; ++i) pkd[i] = f32x2{seed + i, seed - i};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
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
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(p0));
                REP8(X)
#undef X
            } else if constexpr (KIND == 4) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(p0));
                REP8(X)
#undef X
            } else if constexpr (KIND == 5) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == 6) {
#define X(i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[i], 0, 0, 0);
                REP8(X)
#undef X
            } else if constexpr (KIND == 7) {
                acc16[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[0], 0, 0, 0);
                acc16[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[1], 0, 0, 0);
                acc16[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[2], 0, 0, 0);
                acc16[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[3], 0, 0, 0);
                acc16[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[0], 0, 0, 0);
                acc16[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[1], 0, 0, 0);
                acc16[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[2], 0, 0, 0);
                acc16[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[3], 0, 0, 0);
            } else if constexpr (KIND == 8) {
                // the attention inner mix per score pair: exp2 argument already in the MFMA output -> v_exp_f32 x2 + one pack
#define X(i) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_cvt_pkrtz_f16_f32 %2, %0, %1" : "+v"(r[i]), "+v"(r[(i + 4) & 7]), "=v"(pk[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == 9) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pkd[i]) : "v"(pkc));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r[i] + (float)pk[i] + acc4[i][0] + acc4[i][2] + pkd[i][0] + pkd[i][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc16[i][0];
    if (s == 123.456f) out[1023] = 1;            // keep the results alive
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter_instr, double flop_per_instr) {
    unsigned long long* d;
    (void)hipMalloc(&d, 1024 * sizeof(unsigned long long));
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;
        hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(threads), 0, 0, d, 0.37f);      // warm-up (clocks ramp)
        hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(threads), 0, 0, d, 0.37f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        (void)hipMemcpy(h.data(), d, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double cyc = 0;
        for (int w = 0; w < 4 * waves_per_simd; ++w) cyc += (double)h[w];
        cyc /= 4 * waves_per_simd;
        const double n = (double)ITER * 8 * per_iter_instr;
        // s_memtime ticks at a fixed 100 MHz on this part; convert through the measured v_fma rate below instead of trusting a clock guess:
        printf("%-44s waves/SIMD %d: %9.0f ticks for %7.0f instr/wave -> %7.4f ticks per wave-instruction, %7.4f ticks per instruction per SIMD\n",
               name, waves_per_simd, cyc, n, cyc / n, cyc / n / waves_per_simd);
    }
    (void)hipFree(d);
}

int main() {
    printf("# counter = __builtin_readcyclecounter() (s_memtime); compare every row with v_fma_f32 (a full-rate VALU op: 4 clocks per wave64\n"
           "# instruction on a 16-lane SIMD... measured ratio is what matters): ratio to v_fma_f32 = relative issue cost\n");
    run<2>("v_fma_f32", 8, 128);
    run<4>("v_sub_f32", 8, 64);
    run<3>("v_max_f32", 8, 64);
    run<0>("v_exp_f32", 8, 64);
    run<5>("v_rcp_f32", 8, 64);
    run<1>("v_cvt_pkrtz_f16_f32", 8, 64);
    run<8>("2 x v_exp_f32 + v_cvt_pkrtz (per score pair)", 24, 64);
    run<9>("v_pk_fma_f32", 8, 256);
    run<6>("v_mfma_f32_16x16x32_f16", 8, 16384);
    run<7>("v_mfma_f32_32x32x16_f16", 8, 32768);
    return 0;
}
