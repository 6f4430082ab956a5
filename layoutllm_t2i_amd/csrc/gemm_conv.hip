// fp16 MFMA GEMM and implicit-GEMM 3x3 convolution for gfx950 (wave64, v_mfma_f32_32x32x16_f16).
//
//   out[M, N] = A[M, K] . W[N, K]^T  (+ fused epilogue)
//
// A is either a plain (optionally two-source, K-split) row-major matrix, or the implicit im2col view
// of an NHWC feature map for a 3x3 / pad 1 convolution (stride 1, stride 2, or nearest-2x-upsampled
// input).  One K-tile (64) never straddles a filter tap because Cin % 64 == 0.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN x 64, each wave owns a 64 x 64 sub-tile as 2x2
// MFMA 32x32 tiles (64 fp32 accumulators / lane).  Tiles are staged global -> registers -> LDS with
// 16-byte accesses into a double-buffered, XOR-swizzled LDS image (chunk' = chunk ^ ((row >> 1) & 7),
// conflict-free for the 16-lane groups of ds_read_b128 on 128-byte rows); the loads of tile t+1 are
// issued before the MFMAs of tile t and written to LDS after them (one barrier per K-tile).
//
// The MFMA is issued "swapped" (weights as the row operand, activations as the column operand), so a
// lane ends up holding 4 consecutive output channels n for one output row m: the epilogue does 8-byte
// loads/stores and the GEGLU pairing (x | gate in adjacent 32-wide MFMA tiles) is lane-local.
#include "common.h"
#include "gligen_hip.h"

namespace {

constexpr int BK = 64;

struct ConvGeom {
    const half_t* in;
    int B, Hin, Win, Cin, Hout, Wout, stride, ups;
};

template <int BM, int BN, int WAVES_M, int WAVES_N, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(gl_gemm_args p, ConvGeom cg) {
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    constexpr int ACH = BM * (BK / 8) / 256;   // 16-byte chunks of the A tile per thread
    constexpr int BCH = BN * (BK / 8) / 256;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1 && ACH >= 1 && BCH >= 1, "tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* As = reinterpret_cast<half_t*>(smem);   // [2][BM][BK]
    half_t* Bs = As + 2 * BM * BK;                  // [2][BN][BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int M = p.M, N = p.N, K = p.K;
    const int nk = K / BK;

    const int srow = tid >> 3;   // staging row 0..31 (+32 i)
    const int skc = tid & 7;     // 16-byte chunk within the 128-byte tile row

    const half_t* __restrict__ Ag = reinterpret_cast<const half_t*>(p.a);
    const half_t* __restrict__ A2g = reinterpret_cast<const half_t*>(p.a2);
    const half_t* __restrict__ Wg = reinterpret_cast<const half_t*>(p.w);

    // per-thread staging rows
    int cb[ACH], coy[ACH], cox[ACH];   // conv: sample, out-y, out-x (cb < 0: row out of range)
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int m = m0 + srow + 32 * i;
            if (m < M) {
                const int hw = cg.Hout * cg.Wout;
                const int b = m / hw;
                const int r = m - b * hw;
                cb[i] = b;
                coy[i] = r / cg.Wout;
                cox[i] = r - coy[i] * cg.Wout;
            } else {
                cb[i] = -1; coy[i] = 0; cox[i] = 0;
            }
        }
    }

    uint4 ra[ACH], rb[BCH];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if constexpr (CONV) {
            const int tap = k0 / cg.Cin;
            const int ci0 = k0 - tap * cg.Cin;
            const int ky = tap / 3;
            const int kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (cb[i] >= 0) {
                    int iy, ix;
                    bool ok;
                    if (cg.ups) {
                        const int uy = coy[i] + ky - 1, ux = cox[i] + kx - 1;
                        ok = (uy >= 0) && (uy < cg.Hout) && (ux >= 0) && (ux < cg.Wout);
                        iy = uy >> 1; ix = ux >> 1;
                    } else {
                        iy = coy[i] * cg.stride + ky - 1;
                        ix = cox[i] * cg.stride + kx - 1;
                        ok = (iy >= 0) && (iy < cg.Hin) && (ix >= 0) && (ix < cg.Win);
                    }
                    if (ok) {
                        const size_t off = ((size_t)(cb[i] * cg.Hin + iy) * cg.Win + ix) * cg.Cin + ci0 + skc * 8;
                        v = ld16(cg.in + off);
                    }
                }
                ra[i] = v;
            }
        } else {
            const bool second = (A2g != nullptr) && (k0 >= p.ksplit);
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int m = m0 + srow + 32 * i;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (m < M) {
                    const half_t* src = second ? (A2g + (size_t)m * p.lda2 + (k0 - p.ksplit))
                                               : (Ag + (size_t)m * p.lda + k0);
                    v = ld16(src + skc * 8);
                }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int n = n0 + srow + 32 * i;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (n < N) v = ld16(Wg + (size_t)n * K + k0 + skc * 8);
            rb[i] = v;
        }
    };

    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int r = srow + 32 * i;
            st16(As + (size_t)(buf * BM + r) * BK + ((skc ^ ((r >> 1) & 7)) << 3), ra[i]);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int r = srow + 32 * i;
            st16(Bs + (size_t)(buf * BN + r) * BK + ((skc ^ ((r >> 1) & 7)) << 3), rb[i]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const int frow = lane & 31;
    const int fhi = lane >> 5;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8_t xf[TM], wf[TN];
            const int c = ks * 2 + fhi;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int r = wm * (TM * 32) + mi * 32 + frow;
                xf[mi] = *reinterpret_cast<const half8_t*>(As + (size_t)(buf * BM + r) * BK + ((c ^ ((r >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int r = wn * (TN * 32) + ni * 32 + frow;
                wf[ni] = *reinterpret_cast<const half8_t*>(Bs + (size_t)(buf * BN + r) * BK + ((c ^ ((r >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = mfma32(wf[ni], xf[mi], acc[mi][ni]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    // lane holds output row m = ... + (lane & 31) and, per 4-register group rg, the 4 consecutive
    // channels n = ... + 8*rg + 4*(lane >> 5) + {0,1,2,3}.
    const float* __restrict__ bias = p.bias;
    const half_t* __restrict__ res = reinterpret_cast<const half_t*>(p.res);
    const half_t* __restrict__ rowbias = reinterpret_cast<const half_t*>(p.rowbias);
    const int epi = p.epi;
    float gate = 1.0f;
    if (epi == GL_EPI_GATE_RES) gate = p.gate[0];

#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + wm * (TM * 32) + mi * 32 + frow;
        if (m >= M) continue;
        if (epi == GL_EPI_GEGLU) {
            if constexpr (TN % 2 == 0) {
                half_t* out = reinterpret_cast<half_t*>(p.out);
#pragma unroll
                for (int ni = 0; ni < TN; ni += 2) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int nl = 8 * rg + 4 * fhi;
                        const int nx = n0 + wn * (TN * 32) + ni * 32 + nl;   // packed row of x
                        if (nx >= N) continue;
                        const int ng = nx + 32;                               // packed row of gate
                        const int oc = ((n0 + wn * (TN * 32)) >> 1) + (ni >> 1) * 32 + nl;
                        half4_t o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float xv = acc[mi][ni][rg * 4 + j];
                            float gv = acc[mi][ni + 1][rg * 4 + j];
                            if (bias) { xv += bias[nx + j]; gv += bias[ng + j]; }
                            o[j] = (half_t)(xv * gelu_erf_f(gv));
                        }
                        *reinterpret_cast<half4_t*>(out + (size_t)m * p.ldc + oc) = o;
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + wn * (TN * 32) + ni * 32 + 8 * rg + 4 * fhi;
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][rg * 4 + j];
                if (bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (epi == GL_EPI_SILU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
                } else if (epi == GL_EPI_RES) {
                    const half4_t rv = *reinterpret_cast<const half4_t*>(res + (size_t)m * p.ldres + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
                } else if (epi == GL_EPI_GATE_RES) {
                    const half4_t rv = *reinterpret_cast<const half4_t*>(res + (size_t)m * p.ldres + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (float)rv[j] + gate * v[j];
                } else if (epi == GL_EPI_ROWBIAS) {
                    const int s = m / p.rows_per_sample;
                    const half4_t rv = *reinterpret_cast<const half4_t*>(rowbias + (size_t)s * p.ld_rowbias + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
                }
                if (p.out_mode == GL_OUT_F32_NCHW) {
                    float* out = reinterpret_cast<float*>(p.out);
                    const int b = m / p.hw;
                    const int pix = m - b * p.hw;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (n + j < N) out[((size_t)b * N + n + j) * p.hw + pix] = v[j];
                } else {
                    half_t* out = reinterpret_cast<half_t*>(p.out);
                    half4_t o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
                    *reinterpret_cast<half4_t*>(out + (size_t)m * p.ldc + n) = o;
                }
            }
        }
    }
}

template <int BM, int BN>
constexpr int lds_bytes() { return 2 * (BM + BN) * BK * (int)sizeof(half_t); }

template <int BM, int BN, int WM, int WN, bool CONV>
int launch(const gl_gemm_args& g, const ConvGeom& cg, hipStream_t st) {
    dim3 grid(gl_cdiv(g.M, BM), gl_cdiv(g.N, BN));
    constexpr int lds = lds_bytes<BM, BN>();
    gemm_kernel<BM, BN, WM, WN, CONV><<<grid, dim3(256), lds, st>>>(g, cg);
    GL_CHECK_LAUNCH();
    return 0;
}

template <bool CONV>
int dispatch(const gl_gemm_args& g, const ConvGeom& cg, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % BK) != 0 || (g.N % 4) != 0) return GL_ERR_BAD_ARG;
    if (g.epi == GL_EPI_GEGLU && (g.N % 64) != 0) return GL_ERR_BAD_ARG;
    if (g.a2 != nullptr && (g.ksplit % BK) != 0) return GL_ERR_BAD_ARG;
    if ((g.epi == GL_EPI_RES || g.epi == GL_EPI_GATE_RES) && g.res == nullptr) return GL_ERR_BAD_ARG;
    if (g.epi == GL_EPI_GATE_RES && g.gate == nullptr) return GL_ERR_BAD_ARG;
    if (g.epi == GL_EPI_ROWBIAS && (g.rowbias == nullptr || g.rows_per_sample <= 0)) return GL_ERR_BAD_ARG;
    if ((g.N % 128) == 0) return launch<128, 128, 2, 2, CONV>(g, cg, st);
    return launch<256, 64, 4, 1, CONV>(g, cg, st);
}

}  // namespace

extern "C" int gl_gemm(const gl_gemm_args* a, void* stream) {
    if (!a || !a->a || !a->w || !a->out) return GL_ERR_BAD_ARG;
    ConvGeom cg{};
    return dispatch<false>(*a, cg, (hipStream_t)stream);
}

extern "C" int gl_conv3x3(const gl_conv_args* a, void* stream) {
    if (!a || !a->in || !a->g.w || !a->g.out) return GL_ERR_BAD_ARG;
    if ((a->Cin % BK) != 0) return GL_ERR_BAD_ARG;
    if (a->stride != 1 && a->stride != 2) return GL_ERR_BAD_ARG;
    if (a->upsample2x && (a->stride != 1 || a->Hout != 2 * a->Hin || a->Wout != 2 * a->Win)) return GL_ERR_BAD_ARG;
    gl_gemm_args g = a->g;
    g.a = a->in;
    g.a2 = nullptr;
    g.M = a->B * a->Hout * a->Wout;
    g.K = 9 * a->Cin;
    ConvGeom cg{reinterpret_cast<const half_t*>(a->in), a->B, a->Hin, a->Win, a->Cin, a->Hout, a->Wout, a->stride,
                a->upsample2x};
    return dispatch<true>(g, cg, (hipStream_t)stream);
}

extern "C" int gl_init_gemm(void) {
    hipError_t e;
    e = hipFuncSetAttribute((const void*)gemm_kernel<128, 128, 2, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<128, 128>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_kernel<128, 128, 2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<128, 128>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_kernel<256, 64, 4, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<256, 64>());
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_kernel<256, 64, 4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<256, 64>());
    if (e != hipSuccess) return (int)e;
    return 0;
}
