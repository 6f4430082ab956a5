// fp16 MFMA GEMM and implicit-GEMM 3x3 convolution for gfx950 (wave64, v_mfma_f32_32x32x16_f16).
//
//   out[M, N] = A[M, K] . W[N, K]^T  (+ fused epilogue)
//
// A is either a plain (optionally two-source, K-split) row-major matrix, or the implicit im2col view
// of an NHWC feature map for a 3x3 / pad 1 convolution (stride 1, stride 2, or nearest-2x-upsampled
// input).  A K-tile never straddles a filter tap because Cin % 64 == 0.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN (128x128, 128x160, 256x64, 64x128, 256x128/160), MFMA 32x32
// tiles, fp32 accumulators.  Operand tiles go global -> LDS directly (global_load_lds_dwordx4: a wave's 64
// lanes fill 1 KiB of consecutive LDS = 8 or 16 tile rows), so the LDS image is lane-linear and the
// bank-conflict swizzle is applied to the per-lane GLOBAL source chunk and, identically, to the
// fragment reads (chunk' = chunk ^ f(row); f = (row>>1)&7 for 128-byte rows, (row>>2)&3 for 64-byte
// rows: conflict-free for the 16-lane groups of ds_read_b128).  Masked lanes (rows past M/N, the conv
// halo) read a 16-byte zero page, since LDS-DMA cannot be predicated per lane.
//
// Pipeline: BK = 64 (BK = 32 for the short-K GEGLU projections and the 256-row tiles), 2 LDS stages: tile t+1 is
// requested before the MFMAs of tile t; s_waitcnt vmcnt(0) + barrier per K-tile.  The round-1 experiments that lost
// to it (BK 32 / 3-4 stage rings with counted vmcnt, BK 128, 8-wave 256-row tiles, a halo-resident conv) are recorded
// in DESIGN.md section 3 and no longer compiled.
//
// The MFMA is issued "swapped" (weights as the row operand, activations as the column operand) and the
// epilogue restages the fp32 tile through LDS so that outputs, residuals and row-biases move as
// 16-byte row-contiguous accesses; GEGLU pairs x|gate columns that the weight packing interleaved.
//
// Residual stream: with res_f32 / GL_OUT_F32_ROWMAJOR the residual is read and the sum written in fp32 (plus an
// optional fp16 copy for matrix-core consumers), so chained residual adds are never rounded to fp16.
#include <atomic>
#include <cstring>
#include "common.h"
#include "gligen_hip.h"
#include "gemm_shared.h"
#include "opts.h"

namespace {

#define g_opt_ksplit gl_opt(13)  // default 3;        // intra-block K-split variants (64-row wave tiles): 0 off, 1 convs + long-K GEMMs, 2 always, 3 convs only
                             // (default: same-box full-forward A/B, profiles/r2_ab_dispatch_knobs.txt: the plain-GEMM use costs 0.1 ms), 4 GEMMs only
#define g_opt_geglu32 gl_opt(8)  // default 1;       // 1 = short-K GEGLU GEMMs use the 4-blocks/CU BK 32 variant
#define g_opt_big gl_opt(7)  // default 300;         // problems with >= this many 256-row tiles use the 256-row variant (B=4: neutral; B=16: +5-7 %); 0 = off
#define g_opt_small gl_opt(4)  // default 400;       // use 64x128 tiles when the 128-row grid has fewer tiles than this (0 = never)
#define g_opt_splitk_tiles gl_opt(5)  // default 300; // split K only below this many tiles (plain GEMM) ...
#define g_opt_splitk_tiles_conv gl_opt(GL_OPT_SPLITK_TILES_CONV)  // default 450; // ... (conv)
#define g_opt_splitk_nk gl_opt(6)  // default 16;    // ... and at least this many 64-wide K tiles
#define g_opt_skinny gl_opt(24)  // default 64;       // skinny-GEMM kernel while its operand re-reads stay below this many MiB (0 = off)
#define g_opt_order gl_opt(23)  // default 1;         // tile order: 0 = N-tiles fastest, 1 = M-tiles fastest when the weights are the larger operand, 2 = always M
#define g_opt_tile gl_opt(2)  // default 0;          // 0 = auto; 1 = force 128x128 (N >= 256); 2 = prefer 128x160 whenever N % 160 == 0
#define g_opt_g8 gl_opt(30)  // default 1;            // 8-wave deep-pipelined 256-row kernel (gemm8.hip): 0 off, 1 auto (enough tiles), 2 whenever it applies
#define g_opt_g8_tiles gl_opt(31)  // default 200;    // auto: at least this many 256-row tiles (x K slices)
#define g_opt_g8_tapmajor gl_opt(33)  // default 0;   // conv K order of the 8-wave kernel: 1 = (tap, channel block), 0 = (channel block, tap); measured equal in time,
                             // tap-major re-fetches the input 9x from beyond L2 once a level's slab outgrows the 4 MiB L2 (r3 PMC)
#define g_opt_g8_minkt gl_opt(34)  // default 11;     // split-K of the 8-wave kernel: at least this many K-tiles per slice
std::atomic<uint64_t> g8_launches{0};   // launches that went to the 8-wave kernel (tests read it: gl_debug_read(9))
#define g_opt_g8_shortk gl_opt(37)  // default 1: 8-wave kernel also for short-K multi-round grids that fill >= 80 % of their rounds
#define g_opt_g8_bm128 gl_opt(46)  // default set in misc.hip: half-height (128-row) tiles of the 8-wave kernel for under-filled grids: bit 0 convs, bit 1 plain GEMMs
#define g_opt_g8_minblk gl_opt(47)  // default 100: plain GEMMs use the 8-wave kernel from this many blocks (tiles x slices) on
#define g_opt_g8_s3 gl_opt(52)  // default 1: three-pass split-fp16 products (K = 3 * kwrap) run the dedicated three-pass loop of the 8-wave kernel (0 = K-walk)
#define g_opt_g8_minnk gl_opt(35)  // default 5;      // 8-wave kernel only for K >= 64 * this

template <int BM, int BN, int BKT, int NW = 4>
constexpr int lds_bytes() {
    constexpr int pipe = 2 * (BM + BN) * BKT * (int)sizeof(half_t);
    constexpr int epi = NW * 32 * 68 * (int)sizeof(float);    // epilogue staging: NW waves x 32 rows x (64+4) fp32
    return pipe > epi ? pipe : epi;
}

// occupancy hints (waves per SIMD the register allocator must leave room for)
template <int BM, int BN, int BKT, int WK = 1>
constexpr int min_waves() {
    if (WK > 1) return 2;
    if (BKT != 32) return 1;
    if (BM * BN >= 256 * 128) return 2;
    if (BM * BN <= 128 * 128) return 4;
    return 3;
}

// WK = 2: intra-block K split.  The waves form two groups that own the SAME output rows/columns but alternate
// halves of every K-tile's k-steps, so a wave's tile is twice as tall (64 x BN instead of 32 x BN at 4 waves):
// (TM + TN) / (TM * TN) LDS fragment reads per MFMA drop from 1.2 to 0.7 (128x160) -- the CU's LDS port, shared by
// the LDS-DMA writes and the fragment reads, is what bounds the main loop (loop ablation).  The two
// partial accumulators are exchanged through LDS in the epilogue: each wave ends up finalising 32 rows.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool CONV, int BKT, int WK = 1>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WK, (min_waves<BM, BN, BKT, WK>())) void gemm_kernel(gl_gemm_args p, ConvGeom cg, int splitk, int kt_per_split, int order_m) {
    constexpr int NTHR = 64 * WAVES_M * WAVES_N * WK;
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    constexpr int CPR = BKT / 8;                 // 16-byte chunks per tile row (8 or 4)
    constexpr int RPP = NTHR / CPR;              // tile rows covered by one pass of the block's threads
    constexpr int RPW = 64 / CPR;                // tile rows covered by one wave instruction (1 KiB)
    constexpr int APASS = (BM + RPP - 1) / RPP;
    constexpr int BPASS = (BN + RPP - 1) / RPP;
    constexpr int KSTEPS = BKT / 16;
    static_assert(WAVES_M * WAVES_N * WK == 4, "4 waves");
    static_assert(WK == 1 || (WK == 2 && TM == 2 && KSTEPS % 2 == 0), "K-split: two groups, two 32-row tiles per wave");
    static_assert(BM % RPW == 0 && BN % RPW == 0, "whole wave instructions");
    static_assert(BKT == 64 || BKT == 32, "BK");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* As = reinterpret_cast<half_t*>(smem);   // [2][BM][BKT]
    half_t* Bs = As + 2 * BM * BKT;                 // [2][BN][BKT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wk = wave / (WAVES_M * WAVES_N);          // K group (0 when WK == 1)
    const int wmn = wave - wk * (WAVES_M * WAVES_N);
    const int wm = wmn / WAVES_N;
    const int wn = wmn % WAVES_N;
    const int M = p.M, N = p.N, K = p.K;
    // Tile order: N-tiles fastest (tiles sharing an A panel / the same input pixels are co-scheduled), and
    // an XCD-aware bijective remap of the hardware block id (block b runs on XCD b % 8, each XCD has its own
    // 4 MiB L2): XCD x gets one CONTIGUOUS run of logical tiles, so an A panel is fetched into one L2 instead
    // of eight and neighbouring conv tiles share their halo rows there.  Speed only, never correctness.
    const int nt = (N + BN - 1) / BN;
    int tile;
    {
        const int nwg = gridDim.x;
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, local = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    // N-tiles fastest by default: the tiles sharing an A panel run on one XCD.  When the WEIGHTS are the larger operand (the
    // 16x16 / 8x8 levels: 30-59 MB of conv weights against a few MB of activations) M-tiles go fastest instead, so that an
    // XCD owns a contiguous range of N and streams only ITS slice of the weights into its L2 (order_m).
    const int mt_ = (M + BM - 1) / BM;
    const int tm = order_m ? tile % mt_ : tile / nt;
    const int m0 = tm * BM;
    const int n0 = (order_m ? tile / mt_ : tile - tm * nt) * BN;
    const int kt_begin = blockIdx.z * kt_per_split;
    const int kt_end = min(K / BKT, kt_begin + kt_per_split);
    const int nkt = kt_end - kt_begin;

    const int srow = tid / CPR;      // staging row within a pass
    const int skc = tid % CPR;       // LDS chunk slot within the row
    auto swz = [](int r) -> int { return BKT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); };

    const half_t* __restrict__ Ag = reinterpret_cast<const half_t*>(p.a);
    const half_t* __restrict__ A2g = reinterpret_cast<const half_t*>(p.a2);
    const half_t* __restrict__ Wg = reinterpret_cast<const half_t*>(p.w);
    const half_t* zsrc = reinterpret_cast<const half_t*>(g_zero16);

    // ---- per-thread staging state, computed ONCE: every K-tile then costs one pointer bump (plain) or one
    // uniform tap offset + select (conv) per 16-byte LDS-DMA instead of rebuilding 64-bit addresses.
    // Masked rows point at the zero page with a zero increment.
    const half_t* aptr[APASS];       // plain: &A[m][k0 + chunk]   conv: &in[b][oy*stride][ox*stride][chunk]
    unsigned amask = 0u, bmask = 0u; // bit i set <=> pass i stages a real row (its pointer advances by BKT per K-tile)
    unsigned cmask[APASS];           // conv: bit t set <=> filter tap t reads an in-bounds pixel
    int cbyx[APASS];                 // conv + upsample: (sample << 20) | (oy << 10) | ox, -1 for masked rows
    const int k_first = kt_begin * BKT;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int r = srow + RPP * i;
        const int gc = (skc ^ swz(r)) << 3;
        const int m = m0 + r;
        const bool rowok = (r < BM) && (m < M);
        aptr[i] = zsrc; cmask[i] = 0u; cbyx[i] = -1;
        if constexpr (CONV) {
            if (rowok) {
                const int hw = cg.Hout * cg.Wout;
                const int b = m / hw;
                const int rr = m - b * hw;
                const int oy = rr / cg.Wout;
                const int ox = rr - oy * cg.Wout;
                cbyx[i] = (b << 20) | (oy << 10) | ox;
                if (!cg.ups) {
                    unsigned mk = 0u;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int iy = oy * cg.stride + t / 3 - 1, ix = ox * cg.stride + t % 3 - 1;
                        if (iy >= 0 && iy < cg.Hin && ix >= 0 && ix < cg.Win) mk |= 1u << t;
                    }
                    cmask[i] = mk;
                    aptr[i] = cg.in + ((size_t)(b * cg.Hin + oy * cg.stride) * cg.Win + ox * cg.stride) * cg.Cin + gc;
                }
            }
        } else {
            if (rowok) {
                if (A2g != nullptr && k_first >= p.ksplit) aptr[i] = A2g + (size_t)m * p.lda2 + (k_first - p.ksplit) + gc;
                else aptr[i] = Ag + (size_t)m * p.lda + k_first + gc;
                amask |= 1u << i;
            }
        }
    }
    const half_t* bptr[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        const int r = srow + RPP * i;
        const int gc = (skc ^ swz(r)) << 3;
        const int n = n0 + r;
        const bool ok = (r < BN) && (n < N);
        // (gl_conv3x3 sets ldw / kwrap as well: a split-fp16 input walks the same weight rows twice)
        bptr[i] = ok ? (Wg + (size_t)n * p.ldw + (p.kwrap != 0 && k_first >= p.kwrap ? k_first - p.kwrap : k_first) + gc) : zsrc;
        if (ok) bmask |= 1u << i;
    }

    auto issue_tile = [&](int kt, int buf) {
        const int k0 = kt * BKT;
        if constexpr (CONV) {
            // K is ordered (64-channel block, tap, channel): the 9 taps of one channel block are consecutive
            // K-tiles, so they re-read the same [pixels + halo] x 64-channel slab while it is L1/L2-hot.
            const int t64 = k0 >> 6;
            const int cblk = t64 / 9;
            const int tap = t64 - cblk * 9;
            const int acb = (cg.cwrap != 0 && cblk >= cg.cwrap) ? cblk - cg.cwrap : cblk;      // third pass of a split input: hi again
            const int ci0 = (acb << 6) + (k0 & 63);
            const int ky = tap / 3;
            const int kx = tap - ky * 3;
            if (!cg.ups) {
                const int off = ((ky - 1) * cg.Win + (kx - 1)) * cg.Cin + ci0;    // wave-uniform
#pragma unroll
                for (int i = 0; i < APASS; ++i) {
                    if ((BM % RPP) != 0 && RPP * i + wave * RPW >= BM) continue;      // wave-uniform: pass has no rows for this wave
                    const half_t* src = ((cmask[i] >> tap) & 1u) ? (aptr[i] + off) : zsrc;
                    glds16(src, As + (size_t)(buf * BM + RPP * i + wave * RPW) * BKT);
                }
            } else {
#pragma unroll
                for (int i = 0; i < APASS; ++i) {
                    if ((BM % RPP) != 0 && RPP * i + wave * RPW >= BM) continue;
                    const int r = srow + RPP * i;
                    const int gc = (skc ^ swz(r)) << 3;
                    const half_t* src = zsrc;
                    if (cbyx[i] >= 0) {
                        const int uy = ((cbyx[i] >> 10) & 1023) + ky - 1, ux = (cbyx[i] & 1023) + kx - 1;
                        if ((uy >= 0) && (uy < cg.Hout) && (ux >= 0) && (ux < cg.Wout))
                            src = cg.in + ((size_t)((cbyx[i] >> 20) * cg.Hin + (uy >> 1)) * cg.Win + (ux >> 1)) * cg.Cin + ci0 + gc;
                    }
                    glds16(src, As + (size_t)(buf * BM + RPP * i + wave * RPW) * BKT);
                }
            }
        } else {
            if (A2g != nullptr && k0 == p.ksplit && k0 != k_first) {
                // two-source A: crossing into the second matrix (th.cat folded into the GEMM), once per block
                // (ksplit % BKT == 0 is validated by the launcher)
#pragma unroll
                for (int i = 0; i < APASS; ++i) {
                    const int r = srow + RPP * i;
                    const int m = m0 + r;
                    if ((r < BM) && (m < M)) aptr[i] = A2g + (size_t)m * p.lda2 + ((skc ^ swz(r)) << 3);
                }
            }
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                if ((BM % RPP) != 0 && RPP * i + wave * RPW >= BM) continue;
                glds16(aptr[i], As + (size_t)(buf * BM + RPP * i + wave * RPW) * BKT);
                aptr[i] += ((amask >> i) & 1u) ? BKT : 0;
            }
        }
        {
            if (p.kwrap != 0 && k0 == p.kwrap && k0 != k_first) {
                // weight reuse along K ([hi | lo] activations against the same W): back to column 0 of the weight rows, once per block
#pragma unroll
                for (int i = 0; i < BPASS; ++i) bptr[i] -= ((bmask >> i) & 1u) ? p.kwrap : 0;
            }
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            if ((BN % RPP) != 0 && RPP * i + wave * RPW >= BN) continue;
            glds16(bptr[i], Bs + (size_t)(buf * BN + RPP * i + wave * RPW) * BKT);
            bptr[i] += ((bmask >> i) & 1u) ? BKT : 0;
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

    auto compute_tile = [&](int buf) {
#pragma unroll
        for (int kq = 0; kq < KSTEPS / WK; ++kq) {
            half8_t xf[TM], wf[TN];
            const int ks = wk * (KSTEPS / WK) + kq;
            const int c = ks * 2 + fhi;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int r = wm * (TM * 32) + mi * 32 + frow;
                xf[mi] = *reinterpret_cast<const half8_t*>(As + (size_t)(buf * BM + r) * BKT + ((c ^ swz(r)) << 3));
            }
            if constexpr (WK == 2) {
                // 160 accumulator registers: keep ONE weight fragment live at a time (each feeds both m-tiles)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const int r = wn * (TN * 32) + ni * 32 + frow;
                    wf[0] = *reinterpret_cast<const half8_t*>(Bs + (size_t)(buf * BN + r) * BKT + ((c ^ swz(r)) << 3));
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) acc[mi][ni] = mfma32(wf[0], xf[mi], acc[mi][ni]);
                }
            } else {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const int r = wn * (TN * 32) + ni * 32 + frow;
                    wf[ni] = *reinterpret_cast<const half8_t*>(Bs + (size_t)(buf * BN + r) * BKT + ((c ^ swz(r)) << 3));
                }
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = mfma32(wf[ni], xf[mi], acc[mi][ni]);
            }
        }
    };

    issue_tile(kt_begin, 0);
    wait_vmcnt0();
    __syncthreads();
    for (int it = 0; it < nkt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nkt) issue_tile(kt_begin + it + 1, buf ^ 1);
        compute_tile(buf);
        wait_vmcnt0();
        __syncthreads();
    }

    // K-split: group wk finalises m-tile wk of the shared 64 rows.  Swap the two m-tiles' accumulators in group 1
    // (160 v_cndmask, once per tile) so that EVERY wave sends acc[1] and finalises acc[0]: one uniform code path,
    // block barriers executed convergently.
    if constexpr (WK == 2) {
        const bool sw = (wk == 1);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a0 = acc[0][ni][r], a1 = acc[1][ni][r];
                acc[0][ni][r] = sw ? a1 : a0;
                acc[1][ni][r] = sw ? a0 : a1;
            }
    }
    constexpr int TMF = (WK == 2) ? 1 : TM;       // m-tiles a wave finalises
    const int mi_off = (WK == 2) ? wk : 0;        // ... starting at this one

    // K-split exchange of one 64-column pass (2 MFMA tiles): hand the partner group the half it finalises (my
    // acc[1 - mi]), take its acc[mi] and add -- both in the MFMA register layout, so every lane meets exactly its own
    // elements.  Called with compile-time (mi, np) from unrolled loops; two block barriers per call.
    constexpr int XEPS = 64 + 4;
    auto kgroup_exchange = [&](const int mi, const int np) __attribute__((always_inline)) {
        constexpr int WMN = WAVES_M * WAVES_N;
        float* mine = reinterpret_cast<float*>(smem) + wave * (32 * XEPS);
        const float* theirs = reinterpret_cast<const float*>(smem) + (wave < WMN ? wave + WMN : wave - WMN) * (32 * XEPS);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ni = np * 2 + t;
            if (ni < TN) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *reinterpret_cast<float4*>(mine + frow * XEPS + t * 32 + 8 * rg + 4 * fhi) =
                        make_float4(acc[1 - mi][ni][rg * 4], acc[1 - mi][ni][rg * 4 + 1], acc[1 - mi][ni][rg * 4 + 2], acc[1 - mi][ni][rg * 4 + 3]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ni = np * 2 + t;
            if (ni < TN) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const float4 o4 = *reinterpret_cast<const float4*>(theirs + frow * XEPS + t * 32 + 8 * rg + 4 * fhi);
                    acc[mi][ni][rg * 4] += o4.x; acc[mi][ni][rg * 4 + 1] += o4.y;
                    acc[mi][ni][rg * 4 + 2] += o4.z; acc[mi][ni][rg * 4 + 3] += o4.w;
                }
            }
        }
        __syncthreads();      // the partner has read my slab: it can be reused (staging area / next pass)
    };

    if (splitk > 1) {
        // split-K slice: raw fp32 partial tile -> workspace[z][m][n]; epilogue happens in splitk_reduce_kernel.
        // With K groups the two halves are first combined through LDS, so a block still writes ONE slice.
        float* ws = reinterpret_cast<float*>(p.workspace) + (size_t)blockIdx.z * M * N;
        if constexpr (WK == 2) __syncthreads();          // every wave is done reading the operand buffers
#pragma unroll
        for (int mi = 0; mi < TMF; ++mi) {
            const int m = m0 + wm * (TM * 32) + (mi + mi_off) * 32 + frow;
            if constexpr (WK == 2) {
#pragma unroll
                for (int np = 0; np < (TN + 1) / 2; ++np) kgroup_exchange(mi, np);
            }
            if (m >= M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = n0 + wn * (TN * 32) + ni * 32 + 8 * rg + 4 * fhi;
                    if (n >= N) continue;
                    *reinterpret_cast<float4*>(ws + (size_t)m * N + n) =
                        make_float4(acc[mi][ni][rg * 4], acc[mi][ni][rg * 4 + 1], acc[mi][ni][rg * 4 + 2], acc[mi][ni][rg * 4 + 3]);
                }
        }
        return;
    }

    // ------------------------------------------------------------------ epilogue
    const float* __restrict__ bias = p.bias;
    const int epi = p.epi;
    float gate = 1.0f;
    if (epi == GL_EPI_GATE_RES) gate = p.gate[0];

    if (WK == 1 && p.out_mode == GL_OUT_F32_NCHW) {
        // out conv only (N = 4): lane holds row m = ..+(lane&31) and channels 8*rg + 4*(lane>>5) + {0..3}
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m0 + wm * (TM * 32) + mi * 32 + frow;
            if (m >= M) continue;
            float* out = reinterpret_cast<float*>(p.out);
            const int b = m / p.hw;
            const int pix = m - b * p.hw;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = n0 + wn * (TN * 32) + ni * 32 + 8 * rg + 4 * fhi;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (n + j < N) out[((size_t)b * N + n + j) * p.hw + pix] = acc[mi][ni][rg * 4 + j] + (bias ? bias[n + j] : 0.0f);
                }
        }
        return;
    }

    // Row-major output: the fp32 accumulator tile of each wave is staged through LDS (the operand
    // buffers are dead by now), 32 rows x 64 columns (two MFMA tiles) at a time, so that every lane then
    // owns 8 CONSECUTIVE channels of one row: residual / row-bias reads and the output stores are 16-byte
    // accesses, 128 contiguous bytes per row per 8 lanes, instead of 8-byte pieces scattered over 32 rows
    // straight from the MFMA register layout.  For GEGLU a pass is exactly one [x(32) | gate(32)] pair.
    constexpr int EPS = 64 + 4;                            // padded fp32 row stride (conflict-free float4 writes)
    constexpr int NPASS = (TN + 1) / 2;
    constexpr bool VT_OK = !(BM == 256 && BN == 160);      // that tile sits at the 256-register cap: no V^T tail there
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * EPS);
    half_t* outp = reinterpret_cast<half_t*>(p.out);
    const bool geglu = (epi == GL_EPI_GEGLU);
    // one block barrier (every wave is done reading the operand buffers); after it each wave only touches its own
    // staging slab, and LDS operations of one wave execute in order, so the passes need no further barriers
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < TMF; ++mi) {
        const int mbase = m0 + wm * (TM * 32) + (mi + mi_off) * 32;
#pragma unroll
        for (int np = 0; np < NPASS; ++np) {
            if constexpr (WK == 2) kgroup_exchange(mi, np);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ni = np * 2 + t;
                if (ni < TN) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        *reinterpret_cast<float4*>(stage + frow * EPS + t * 32 + 8 * rg + 4 * fhi) =
                            make_float4(acc[mi][ni][rg * 4], acc[mi][ni][rg * 4 + 1], acc[mi][ni][rg * 4 + 2], acc[mi][ni][rg * 4 + 3]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int nbase = n0 + wn * (TN * 32) + np * 64;   // first (packed) column of this pass
            if (VT_OK && p.vt != nullptr && nbase >= p.vt_col0) {
                // V^T tail of a fused QKV projection: lane = one channel column of the pass, 8 consecutive tokens per store
                // (read down the staged tile: consecutive lanes hit consecutive banks), so V never exists row-major
                const int ncols = (np * 2 + 1 < TN) ? 64 : 32;
                const int n = nbase + lane;
                if (lane < ncols && n < N) {
                    const int nv = n - p.vt_col0;
                    const int hh = nv / p.vt_d;
                    const int cc = nv - hh * p.vt_d;
                    const float bv = bias ? bias[n] : 0.0f;
                    half_t* vtp = reinterpret_cast<half_t*>(p.vt);
#pragma unroll
                    for (int tg = 0; tg < 4; ++tg) {
                        const int m = mbase + tg * 8;
                        if (m >= M) continue;
                        half8_t o;
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = (half_t)(stage[(tg * 8 + k) * EPS + lane] + bv);
                        if ((p.vt_rows & 7) == 0) {
                            const int b = m / p.vt_rows;
                            const int key = m - b * p.vt_rows;
                            st16(vtp + ((size_t)(b * p.vt_H + hh) * p.vt_d + cc) * p.vt_ld + key, *reinterpret_cast<uint4*>(&o));
                        } else {
                            // ragged rows per sample (the fuser's N + 30 keys): 8 tokens may straddle two samples
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const int mm = m + k;
                                if (mm < M) {
                                    const int b = mm / p.vt_rows;
                                    const int key = mm - b * p.vt_rows;
                                    vtp[((size_t)(b * p.vt_H + hh) * p.vt_d + cc) * p.vt_ld + key] = o[k];
                                }
                            }
                        }
                    }
                }
            } else if (geglu) {
                // 32 output columns per pass, 8 per lane: 4 lanes per row, 16 rows per sweep
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int r = ps * 16 + (lane >> 2);
                    const int pc = (lane & 3) * 8;
                    const int m = mbase + r;
                    const int nx = nbase + pc;
                    if (m < M && nx < N) {
                        const float4 x0 = *reinterpret_cast<const float4*>(stage + r * EPS + pc);
                        const float4 x1 = *reinterpret_cast<const float4*>(stage + r * EPS + pc + 4);
                        const float4 g0 = *reinterpret_cast<const float4*>(stage + r * EPS + pc + 32);
                        const float4 g1 = *reinterpret_cast<const float4*>(stage + r * EPS + pc + 36);
                        float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                        half8_t o, lo8 = {};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float a = xv[j], b = gv[j];
                            if (bias) { a += bias[nx + j]; b += bias[nx + 32 + j]; }
                            float y = a * gelu_erf_f(b);
                            if (p.out_mode == GL_OUT_F16_HILO) y = pin_value(y);      // hi and lo from ONE value; the default mode keeps its plain code
                            o[j] = (half_t)y;
                            if (p.out_mode == GL_OUT_F16_HILO) lo8[j] = (half_t)(y - (float)o[j]);
                        }
                        st16(outp + (size_t)m * p.ldc + (nbase >> 1) + pc, *reinterpret_cast<uint4*>(&o));
                        // [hi | lo] rows for a split-fp16 FeedForward output projection: lo goes N / 2 (the output width) columns to the right
                        if (p.out_mode == GL_OUT_F16_HILO) st16(outp + (size_t)m * p.ldc + (N >> 1) + (nbase >> 1) + pc, *reinterpret_cast<uint4*>(&lo8));
                    }
                }
            } else {
                const int ncols = (np * 2 + 1 < TN) ? 64 : 32;  // a trailing odd tile fills only half the pass
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int r = ps * 8 + (lane >> 3);
                    const int c = (lane & 7) * 8;
                    const int m = mbase + r;
                    const int n = nbase + c;
                    if (c < ncols && m < M && n < N) {
                        const float4 a0 = *reinterpret_cast<const float4*>(stage + r * EPS + c);
                        const float4 a1 = *reinterpret_cast<const float4*>(stage + r * EPS + c + 4);
                        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                        finish8(p, gate, m, n, v);
                    }
                }
            }
        }
    }
}

// Sums the split-K partial tiles and applies the epilogue (row-major outputs only).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(gl_gemm_args p, int splitk) {
    const int M = p.M, N = p.N;
    const int nq = N / 8;
    const size_t total = (size_t)M * nq;
    const float* ws = reinterpret_cast<const float*>(p.workspace);
    float gate = 1.0f;
    if (p.epi == GL_EPI_GATE_RES) gate = p.gate[0];
    const size_t zstride = (size_t)M * N;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * 256u) {
        const int m = (int)(idx / (unsigned)nq);
        const int n = (int)(idx - (unsigned)m * (unsigned)nq) * 8;
        const float* src = ws + (size_t)m * N + n;
        float4 a = *reinterpret_cast<const float4*>(src);
        float4 c = *reinterpret_cast<const float4*>(src + 4);
        // slices are added in index order (deterministic); independent loads in flight instead of a load -> wait -> add chain
        int z = 1;
        for (; z + 1 < splitk; z += 2) {
            const float4 b0 = *reinterpret_cast<const float4*>(src + (size_t)z * zstride);
            const float4 d0 = *reinterpret_cast<const float4*>(src + (size_t)z * zstride + 4);
            const float4 b1 = *reinterpret_cast<const float4*>(src + (size_t)(z + 1) * zstride);
            const float4 d1 = *reinterpret_cast<const float4*>(src + (size_t)(z + 1) * zstride + 4);
            a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
            c.x += d0.x; c.y += d0.y; c.z += d0.z; c.w += d0.w;
            a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
            c.x += d1.x; c.y += d1.y; c.z += d1.z; c.w += d1.w;
        }
        for (; z < splitk; ++z) {
            const float4 b = *reinterpret_cast<const float4*>(src + (size_t)z * zstride);
            const float4 d = *reinterpret_cast<const float4*>(src + (size_t)z * zstride + 4);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            c.x += d.x; c.y += d.y; c.z += d.z; c.w += d.w;
        }
        float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        finish8(p, gate, m, n, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Skinny GEMM (M <= ~1000 rows: the 30-slot relation / grounding-token chains, the timestep MLP).  With 4-20 tiles
// of the LDS-staged kernel these problems are pure latency: every K-step of its 2-stage ring exposes one cold
// HBM / cross-XCD round trip (12-18 us for a 240 x 320 x 320 product).  Here a block owns one 32 x 32 output tile,
// its four waves split K, and each wave loads its MFMA operand fragments straight from global memory into registers
// -- up to 20 k-steps (K = 1280 per block) in flight at once, no LDS staging, no barrier in the loop: one round trip.
// The four partial tiles are added in wave order through LDS (deterministic), then the shared finish8 epilogue.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SK_BATCH = 20;            // k-steps of 16 whose operand loads are issued back to back (160 VGPRs)
constexpr int SK_RS = 36;               // fp32 row stride of a staged partial tile

__global__ __launch_bounds__(256) void gemm_skinny_kernel(gl_gemm_args p) {
    __shared__ __attribute__((aligned(16))) float part[4][32 * SK_RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int M = p.M, K = p.K;
    const int steps = K / 64;                                   // k-steps of 16 per wave (K % 64 == 0)
    const int kw = wave * steps * 16;                           // this wave's K slice
    int mr = m0 + (lane & 31);
    if (mr >= M) mr = M - 1;                                    // ragged last tile: clamp (rows discarded below)
    const half_t* ap = reinterpret_cast<const half_t*>(p.a) + (size_t)mr * p.lda + kw + 8 * (lane >> 5);
    // weight reuse along K: K == 2 * kwrap, so the four K slices never straddle the wrap
    const half_t* wp = reinterpret_cast<const half_t*>(p.w) + (size_t)(n0 + (lane & 31)) * p.ldw + (p.kwrap != 0 && kw >= p.kwrap ? kw - p.kwrap : kw) +
                       8 * (lane >> 5);
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    for (int s0 = 0; s0 < steps; s0 += SK_BATCH) {
        uint4 af[SK_BATCH], wf[SK_BATCH];
#pragma unroll
        for (int j = 0; j < SK_BATCH; ++j)
            if (s0 + j < steps) {
                af[j] = ld16(ap + (s0 + j) * 16);
                wf[j] = ld16(wp + (s0 + j) * 16);
            }
#pragma unroll
        for (int j = 0; j < SK_BATCH; ++j)
            if (s0 + j < steps)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8_t*>(&wf[j]), *reinterpret_cast<half8_t*>(&af[j]), acc, 0, 0, 0);
    }
    // lane holds row m = lane & 31, columns 8 * rg + 4 * (lane >> 5) + {0..3}
    float* mine = part[wave] + (lane & 31) * SK_RS + 4 * (lane >> 5);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
        *reinterpret_cast<float4*>(mine + 8 * rg) = make_float4(acc[rg * 4], acc[rg * 4 + 1], acc[rg * 4 + 2], acc[rg * 4 + 3]);
    __syncthreads();
    if (threadIdx.x < 128) {
        const int r = threadIdx.x >> 2, c = (threadIdx.x & 3) * 8;
        const int m = m0 + r;
        if (m < M) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[0][r * SK_RS + c + j];
#pragma unroll
            for (int w = 1; w < 4; ++w)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += part[w][r * SK_RS + c + j];
            float gate = 1.0f;
            if (p.epi == GL_EPI_GATE_RES) gate = p.gate[0];
            finish8(p, gate, m, n0 + c, v);
        }
    }
}

// rows x (N / 32) tiles, every tile streams (32 + 32) x K operand halves: worth it while that stays cache-sized
inline bool skinny_ok(const gl_gemm_args& g) {
    if (!g_opt_skinny || g.M > 1024 || (g.N % 32) != 0 || g.epi == GL_EPI_GEGLU || g.vt != nullptr || g.a2 != nullptr) return false;
    if (g.kwrap != 0 && g.K != 2 * g.kwrap) return false;
    if (g.out_mode == GL_OUT_F32_NCHW || (g.lda % 8) != 0) return false;
    // a block walks the whole K with four waves: beyond K = 4096 (the relation chain's FF2 at 1280 channels: 31.6 us on 64 rows) the
    // LDS-staged kernels with split-K are faster (12 + 5 us)
    if (g.K > 4096 && g.workspace != nullptr) return false;
    const long tiles = (long)gl_cdiv(g.M, 32) * (g.N / 32);
    return tiles * 64L * g.K * 2L <= (long)g_opt_skinny << 20;
}

// How many K slices: only when the tile grid underfills the chip and K is long enough to amortise
// the fp32 partial round trip.
inline int choose_splitk(const gl_gemm_args& g, int tiles, bool conv) {
    const int nk = g.K / 64;
    if (!g.workspace || g.epi == GL_EPI_GEGLU || g.out_mode == GL_OUT_F32_NCHW || g.vt != nullptr) return 1;
    // convs (long K, weights streamed once per row tile) profit up to ~1.7 tiles per CU: the 32x32-level
    // convs launch exactly 256 tiles and went 604 -> 694 TF/s with 2 K-slices; plain GEMMs only below ~300
    if (tiles >= (conv ? g_opt_splitk_tiles_conv : g_opt_splitk_tiles) || nk < g_opt_splitk_nk) return 1;
    int s = (480 + tiles - 1) / tiles;
    if (s > nk / 8) s = nk / 8;
    if (s > 16) s = 16;
    while (s > 1 && (int64_t)s * g.M * g.N * 4 > g.workspace_bytes) --s;
    return s < 2 ? 1 : s;
}

template <int BM, int BN, int WM, int WN, bool CONV, int BKT, int WK = 1>
int launch(const gl_gemm_args& g, const ConvGeom& cg, hipStream_t st) {
    if (g.a2 != nullptr && (g.ksplit % BKT) != 0) return GL_ERR_BAD_ARG;    // the two-source crossover happens on a K-tile edge
    if (g.vt != nullptr && ((g.vt_col0 % BN) % 64) != 0) return GL_ERR_UNSUPPORTED;   // the V^T tail must start on an epilogue pass
    const int mt = gl_cdiv(g.M, BM), nt = gl_cdiv(g.N, BN);
    const int nk = g.K / BKT;
    int splitk = choose_splitk(g, mt * nt, CONV);
    int kper = gl_cdiv(nk, splitk);
    const int zs = gl_cdiv(nk, kper);          // slices that actually have work
    dim3 grid(mt * nt, 1, zs);
    constexpr int lds = lds_bytes<BM, BN, BKT, WM * WN * WK>();
    // weights (N x K) larger than the activations they meet (M x K, or M x K / 9 distinct bytes for a conv)?
    const int order_m = g_opt_order == 1 ? ((CONV ? 9L : 1L) * g.N > (long)g.M) : (g_opt_order == 2);
    gemm_kernel<BM, BN, WM, WN, CONV, BKT, WK><<<grid, dim3(64 * WM * WN * WK), lds, st>>>(g, cg, zs, kper, order_m);
    GL_CHECK_LAUNCH();
    if (zs > 1) {
        const size_t total = (size_t)g.M * (g.N / 8);
        int nblk = (int)((total + 255) / 256);
        if (nblk > 2048) nblk = 2048;
        splitk_reduce_kernel<<<dim3(nblk), dim3(256), 0, st>>>(g, zs);
        GL_CHECK_LAUNCH();
    }
    return 0;
}

template <bool CONV, int BKT>
int dispatch_shape(const gl_gemm_args& g, const ConvGeom& cg, hipStream_t st) {
    // tile shape: 128x160 (4 waves x 32x160) when it divides N exactly -- N = 320/640/960/... are the
    // channel widths of this UNet and 128-wide tiles would waste up to 17 % of the MFMA work there
    // (convs, whose K is long, prefer it even when 128 also divides N: measured 442 vs 408 and 607 vs 536 TF/s
    // at the 32x32 and 16x16 levels); otherwise 128x128; 256x64 only for narrow outputs.
    const bool geglu = (g.epi == GL_EPI_GEGLU);
    const bool rowmajor = (g.out_mode != GL_OUT_F32_NCHW);
    int shape = 0;                                   // 0: 128x128, 1: 128x160, 2: 256x64, 3: 64x128
    if (g.N < 256 && (g.N % 128) != 0) shape = 2;
    else if (!geglu && (g.N % 160) == 0 && ((g.N % 128) != 0 || CONV || g_opt_tile == 2)) shape = 1;
    if (g_opt_tile == 1 && g.N >= 256) shape = 0;
    // few tiles (mid / low resolution levels): halve the tile height so the grid covers the 256 CUs at
    // 2-3 blocks each instead of leaving half of them idle (M=8192,N=640 -> 256 tiles of 128x160 vs 640 of 64x128)
    // (measured: helps the K <= 1280 projections, 22 -> 18 us; hurts long-K GEMMs and convs, whose
    //  weight stream then gets re-read by twice as many row tiles)
    if (g_opt_small && !CONV && g.K <= 1280 && shape != 2 && (g.N % 128) == 0) {
        const long t128 = (long)gl_cdiv(g.M, 128) * gl_cdiv(g.N, shape == 1 ? 160 : 128);
        if (t128 < g_opt_small) shape = 3;
    }
    if constexpr (BKT == 64) {
        // intra-block K-split (64-row wave tiles, 0.7-0.75 LDS fragment reads per MFMA): +7-18 % on every conv of the
        // UNet (with or without split-K slices: the two K groups are combined in LDS before a partial slice is written)
        // and on K >= 1024 GEMMs; its accumulator exchange in the epilogue costs 5-20 % on short K (the K = 320 / 640
        // projections), which stay on the 4 x (32 x BN) kernels (per-shape A/B in DESIGN.md)
        if (g_opt_ksplit && rowmajor && (shape == 0 || shape == 1)) {
            const int nk = g.K / 64;
            bool use = (g_opt_ksplit == 2);
            // (128-wide conv tiles only occur in the VAE decoder, M = 0.26-1 M pixels x 128/256 channels: measured 2 % slower)
            if (g_opt_ksplit == 1 || g_opt_ksplit == 3 || g_opt_ksplit == 4) use = CONV ? (shape == 1 && nk >= 20) : (nk >= 16);
            if (g_opt_ksplit == 3 && !CONV) use = false;      // A/B: convs only
            if (g_opt_ksplit == 4 && CONV) use = false;       // A/B: plain GEMMs only
            if (use) {
                if (shape == 1) return launch<128, 160, 2, 1, CONV, 64, 2>(g, cg, st);
                return launch<128, 128, 2, 1, CONV, 64, 2>(g, cg, st);
            }
        }
        // 256-row tiles (4 waves x 64-row wave tiles, BK 32) once the problem has >= g_opt_big of them
        if (g_opt_big && shape != 2 && shape != 3 && g.M >= 256 && !(shape == 1 && g.vt != nullptr)) {
            const int bn = (shape == 1) ? 160 : 128;
            const long t256 = (long)gl_cdiv(g.M, 256) * gl_cdiv(g.N, bn);
            if (t256 >= g_opt_big) {
                if (shape == 1) return launch<256, 160, 4, 1, CONV, 32>(g, cg, st);
                return launch<256, 128, 4, 1, CONV, 32>(g, cg, st);
            }
        }
        if constexpr (!CONV) {
            // K-split for the small-tile shape too (64x64 instead of 32x64 wave tiles): 22.4 -> 20.4 us at
            // M = 2048, N = K = 1280; neutral at K = 640, which stays on the plain kernel
            if (shape == 3 && (g_opt_ksplit == 2 || ((g_opt_ksplit == 1 || g_opt_ksplit == 4 || g_opt_ksplit == 5) && g.K >= 1024)))
                return launch<64, 128, 1, 2, false, 64, 2>(g, cg, st);
        }
        if (shape == 3) return launch<64, 128, 2, 2, CONV, 64>(g, cg, st);
        if (shape == 1) return launch<128, 160, 4, 1, CONV, 64>(g, cg, st);
        if (shape == 0) return launch<128, 128, 2, 2, CONV, 64>(g, cg, st);
        return launch<256, 64, 4, 1, CONV, 64>(g, cg, st);
    } else {
        // BK 32 (short-K GEGLU projections only): 128x128 / 64x128 at 4 blocks per CU, so that one block's erf epilogue
        // overlaps the other blocks' main loops (256-row tiles measured 13 % slower here: 181 vs 157 ms over the trace)
        static_assert(!CONV, "BK 32 dispatch is for plain GEMMs");
        if (shape == 3) return launch<64, 128, 2, 2, false, 32>(g, cg, st);
        return launch<128, 128, 2, 2, false, 32>(g, cg, st);
    }
}

template <bool CONV>
int dispatch(const gl_gemm_args& g, const ConvGeom& cg, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % 64) != 0) return GL_ERR_BAD_ARG;
    if (g.out_mode < 0 || g.out_mode > GL_OUT_F16_HILO) return GL_ERR_BAD_ARG;
    if (g.out_mode == GL_OUT_F16_HILO && g.ldc < (g.epi == GL_EPI_GEGLU ? g.N : 2 * g.N)) return GL_ERR_BAD_ARG;
    // the transposed tail of a [hi | lo] output takes both halves (vt_lo, ABI 15); a plain output has no residual to write
    if ((g.vt != nullptr && (g.out_mode == GL_OUT_F16_HILO) != (g.vt_lo != nullptr)) || (g.vt == nullptr && g.vt_lo != nullptr)) return GL_ERR_BAD_ARG;
    if (g.out_mode != GL_OUT_F32_NCHW && ((g.N % 8) != 0 || (g.ldc % 8) != 0)) return GL_ERR_BAD_ARG;
    if (g.out_mode == GL_OUT_F32_ROWMAJOR && g.out2 != nullptr && (g.ldc2 % 8) != 0) return GL_ERR_BAD_ARG;
    if (g.out_mode == GL_OUT_F32_ROWMAJOR && g.epi == GL_EPI_GEGLU) return GL_ERR_UNSUPPORTED;
    if (g.res != nullptr && (g.ldres % 8) != 0) return GL_ERR_BAD_ARG;
    if (g.rowbias != nullptr && (g.ld_rowbias % 8) != 0) return GL_ERR_BAD_ARG;
    if (g.epi == GL_EPI_GEGLU && (g.N % 64) != 0) return GL_ERR_BAD_ARG;
    if (g.a2 != nullptr && (g.ksplit % 64) != 0) return GL_ERR_BAD_ARG;
    if ((g.epi == GL_EPI_RES || g.epi == GL_EPI_GATE_RES) && g.res == nullptr) return GL_ERR_BAD_ARG;
    if (g.epi == GL_EPI_GATE_RES && g.gate == nullptr) return GL_ERR_BAD_ARG;
    if (g.epi == GL_EPI_ROWBIAS && (g.rowbias == nullptr || g.rows_per_sample <= 0)) return GL_ERR_BAD_ARG;
    if (g.vt != nullptr && (g.epi != GL_EPI_BIAS || (g.out_mode != GL_OUT_F16_ROWMAJOR && g.out_mode != GL_OUT_F16_HILO) || (g.vt_col0 % 64) != 0 || g.vt_col0 <= 0 ||
                            g.vt_col0 >= g.N || g.vt_rows <= 0 || g.vt_d <= 0 || ((g.N - g.vt_col0) % g.vt_d) != 0 || (g.vt_ld % 8) != 0 ||
                            g.vt_ld < g.vt_rows || g.vt_H * g.vt_d != g.N - g.vt_col0))
        return GL_ERR_BAD_ARG;
    // GEGLU with a short K (levels 0/1: K = 320/640, 5-10 k-tiles) spends a large share of each block in its
    // erf epilogue; BK 32 / 2-stage needs 35 KiB of LDS and 114 registers, so 4 blocks/CU are resident and
    // one block's epilogue overlaps the others' main loops (measured 150 -> 135 us and 109 -> 100 us; long-K
    // GEMMs and convs lose 10-20 % to the doubled barrier count, so they stay on BK 64)
    if constexpr (!CONV) {
        if (g.vt_lo == nullptr && skinny_ok(g)) {
            gemm_skinny_kernel<<<dim3(g.N / 32, gl_cdiv(g.M, 32)), dim3(256), 0, st>>>(g);
            GL_CHECK_LAUNCH();
            return 0;
        }
    }
    // the [hi | lo] transposed tail (vt_lo, ABI 15) is implemented by the 8-wave kernel's epilogue only: the 4-wave instantiations sit at their
    // register caps (adding the second 8-token vector cost gemm_kernel<256,128,4,1,.,32> 576 bytes of scratch and the default mode 0.3 ms per forward)
    const bool need8 = g.vt_lo != nullptr;
    if (need8) {
        int bn_ = 0;
        if (!g_opt_g8 || !gl8_supported(g, CONV, &bn_)) return GL_ERR_UNSUPPORTED;
    }
    if (g_opt_g8) {
        int bn = 0;
        if (gl8_supported(g, CONV, &bn)) {
            const int nk = g.K / 64;
            // One block per CU: when the 256-row grid underfills the chip (32x32 levels and below), K is cut into slices up to ONE
            // round of blocks (256), never below g_opt_g8_minkt K-tiles per slice (prologue + epilogue cost ~4 K-tiles of time).
            auto plan_split = [&](const int tiles_) {
                int sk = 1;
                if (tiles_ < g_opt_g8_tiles && g.workspace && g.epi != GL_EPI_GEGLU && g.vt == nullptr) {
                    sk = 256 / tiles_;
                    if (sk > nk / g_opt_g8_minkt) sk = nk / g_opt_g8_minkt;
                    while (sk > 1 && (int64_t)sk * g.M * g.N * 4 > g.workspace_bytes) --sk;
                    if (sk < 1) sk = 1;
                }
                return sk;
            };
            int bm = 256;
            int tiles = gl_cdiv(g.M, 256) * gl_cdiv(g.N, bn);
            int splitk = plan_split(tiles);
            // Half-height (128-row) tiles where the 256-row grid covers at most half the chip (32x32 maps and below at 2B = 8): twice the
            // blocks, so half the K slices (none at 32x32) and a smaller or no reduction.  The 128-row main loop issues the same B-side
            // LDS-DMA per K-tile for half the MFMAs and runs ~1.4x slower per FLOP, so it only pays where the per-block fixed costs and
            // the reduction dominate (profiles/r4_bm128_shapes.txt): every plain GEMM on such a grid (-20...-32 % at 8192 rows, -3...-17 %
            // at 2048), and the convs whose 256-row plan leaves <= 16 K-tiles per slice (8x8 maps, stride-2 convs: -3...-10 %; long-K 3x3
            // convs at 16x16 / 32x32 lose 8-25 %).  Key 46: bit 0 convs (by that rule), bit 1 plain GEMMs, bit 2 every conv (A/B).
            if (tiles * 2 <= 256 && g.epi != GL_EPI_GEGLU && g.vt == nullptr) {
                const bool want = CONV ? ((g_opt_g8_bm128 & 4) || ((g_opt_g8_bm128 & 1) && nk / splitk <= 16)) : ((g_opt_g8_bm128 & 2) != 0);
                if (want) {
                    bm = 128;
                    tiles = gl_cdiv(g.M, 128) * gl_cdiv(g.N, bn);
                    splitk = plan_split(tiles);
                }
            }
            // multi-round plain GEMMs whose 256-row grid ends in a mostly empty round (8192 x 1920 x 640: 384 tiles = 1.5 rounds) while the
            // 128-row grid fills its rounds (768 = 3.0): half-height tiles as well (key 46 bit 3)
            if (!CONV && (g_opt_g8_bm128 & 8) && bm == 256 && splitk == 1 && tiles > 256 && g.epi != GL_EPI_GEGLU) {
                const int t128 = gl_cdiv(g.M, 128) * gl_cdiv(g.N, bn);
                auto fills = [](const int b) { return b * 5 >= ((b + 255) / 256) * 256 * 4; };
                if (!fills(tiles) && fills(t128) && t128 <= 2048) {
                    bm = 128;
                    tiles = t128;
                }
            }
            // plain GEMMs: measured per shape (profiles/r3_g8_probe.txt) -- multi-round grids with a short K stay on the 4-wave kernels
            // (several resident blocks hide each other's prologue / epilogue), split-K slices need >= 20 K-tiles to pay for the reduction
            if (!CONV && splitk > 1 && nk / splitk < 20) splitk = nk / 20 > 0 ? nk / 20 : 1;
            // (plain GEMMs: the deep ring pays from ~100 blocks on -- 2048 x 1280 x 1280 on 128 half-height tiles 18.8 -> 17.0 us against the
            // 4-wave 64 x 128 tiles, M = 512 shapes -15...-23 %; key 47.  The split-K decision above keeps its own threshold, key 31.)
            bool enough = tiles * splitk >= (CONV ? g_opt_g8_tiles : (g_opt_g8_minblk < g_opt_g8_tiles ? g_opt_g8_minblk : g_opt_g8_tiles)) && nk >= g_opt_g8_minnk;
            if (!CONV && enough) {
                // multi-round grids with a short K: nothing hides a block's prologue / epilogue at one block per CU, so the last, partly
                // filled round must not cost more than the deeper loop wins -- only grids that fill >= 80 % of their rounds (measured
                // with the hoisted-load epilogue, tools/g8_probe.py time with G8_SHORTK=1, profiles/r3_g8_shortk.txt: GEGLU
                // 32768x2560x320 104 -> 84 us, QKV 32768x960x320 49 -> 41 us, 9216x1920x640 55 -> 34 us; the ragged fuser QKV, 774 blocks =
                // 3.02 rounds, loses 5 %); plain GEMMs beyond 2048 blocks lose 3 % (131072 rows) and stay on the 4-wave kernels
                const int blocks = tiles * splitk;
                const bool full_rounds = g_opt_g8_shortk && blocks * 5 >= ((blocks + 255) / 256) * 256 * 4;
                enough = nk >= 16 || blocks <= 256 || (g.epi == GL_EPI_GEGLU && (nk >= 10 || full_rounds)) || (full_rounds && blocks <= 2048);
            }
            if (g_opt_g8 == 2 || enough || need8) {
                // the three-pass product xhi.Whi + xlo.Whi + xhi.Wlo in its K-walk description (K = 3 * kwrap; GEMM: the third A segment is
                // the first one again; conv: in_split == 3) -> the dedicated three-pass loop: same slices, counted in 32-wide stages of kwrap
                const bool s3 = g_opt_g8_s3 != 0 && g.kwrap != 0 && g.K == 3 * g.kwrap &&
                                (CONV ? cg.cwrap != 0 : (g.a2 == g.a && g.lda2 == g.lda && g.ksplit == 2 * g.kwrap));
                const int nks = s3 ? g.kwrap / 32 : nk;
                const int kper = gl_cdiv(nks, splitk);
                const int zs = gl_cdiv(nks, kper);
                const int order_m = g_opt_order == 1 ? ((CONV ? 9L : 1L) * g.N > (long)g.M) : (g_opt_order == 2);
                const int e = gl8_launch(g, cg, CONV, bm, bn, zs, kper, order_m | (CONV && g_opt_g8_tapmajor ? 2 : 0), st, s3);
                if (e) return e;
                g8_launches.fetch_add(1, std::memory_order_relaxed);
                if (zs > 1) {
                    const size_t total = (size_t)g.M * (g.N / 8);
                    int nblk = (int)((total + 255) / 256);
                    if (nblk > 2048) nblk = 2048;
                    splitk_reduce_kernel<<<dim3(nblk), dim3(256), 0, st>>>(g, zs);
                    GL_CHECK_LAUNCH();
                }
                return 0;
            }
        }
    }
    if constexpr (!CONV) {
        if (g_opt_geglu32 && g.epi == GL_EPI_GEGLU && g.K <= 640 && g.a2 == nullptr) return dispatch_shape<false, 32>(g, cg, st);
    }
    return dispatch_shape<CONV, 64>(g, cg, st);
}

}  // namespace

extern "C" int gl_gemm(const gl_gemm_args* a, void* stream) {
    if (!a || !a->a || !a->w || !a->out) return GL_ERR_BAD_ARG;
    ConvGeom cg{};
    gl_gemm_args g = *a;
    // kwrap: the weight column of K index k is k for k < kwrap and k - kwrap beyond (ONE step back): K = 2 * kwrap walks the same kwrap weight
    // columns twice ([hi | lo] activations), K = 3 * kwrap walks columns [0, kwrap) twice and then [kwrap, 2 * kwrap) -- [Whi | Wlo] weight rows
    // for the three-pass product xhi.Whi + xlo.Whi + xhi.Wlo, whose third A segment the caller supplies as the second source (a2 = xhi)
    if (g.kwrap != 0 && ((g.kwrap % 64) != 0 || g.kwrap >= g.K || g.K > 3 * g.kwrap)) return GL_ERR_BAD_ARG;
    if (g.ldw == 0) g.ldw = g.kwrap != 0 ? g.kwrap : g.K;
    if (g.ldw < (g.kwrap != 0 ? (g.K > 2 * g.kwrap ? 2 * g.kwrap : g.kwrap) : g.K) || (g.ldw % 8) != 0) return GL_ERR_BAD_ARG;
    return dispatch<false>(g, cg, (hipStream_t)stream);
}

extern "C" int gl_conv3x3(const gl_conv_args* a, void* stream) {
    if (!a || !a->in || !a->g.w || !a->g.out) return GL_ERR_BAD_ARG;
    if ((a->Cin % 64) != 0) return GL_ERR_BAD_ARG;
    if (a->stride != 1 && a->stride != 2) return GL_ERR_BAD_ARG;
    if (a->upsample2x && (a->stride != 1 || a->Hout != 2 * a->Hin || a->Wout != 2 * a->Win)) return GL_ERR_BAD_ARG;
    if (a->B >= 2048 || a->Hout >= 1024 || a->Wout >= 1024) return GL_ERR_BAD_ARG;     // packed (b, oy, ox) row coordinates
    if (a->in_split != 0 && a->in_split != 2 && a->in_split != 3) return GL_ERR_BAD_ARG;
    if (a->in_split == 3 && !a->w_split) return GL_ERR_BAD_ARG;      // the third pass multiplies by Wlo
    gl_gemm_args g = a->g;
    g.a = a->in;
    g.a2 = nullptr;
    g.M = a->B * a->Hout * a->Wout;
    // split-fp16 input (strict mode): pixel rows [hi | lo] of 2 Cin channels; the K walk visits hi, lo (, hi) against W, W (, Wlo)
    g.K = 9 * a->Cin * (a->in_split ? a->in_split : 1);
    g.ldw = 9 * a->Cin * (a->w_split ? 2 : 1);
    g.kwrap = a->in_split ? 9 * a->Cin : 0;
    ConvGeom cg{reinterpret_cast<const half_t*>(a->in), a->B, a->Hin, a->Win, a->Cin * (a->in_split ? 2 : 1), a->Hout, a->Wout, a->stride,
                a->upsample2x, nullptr, a->in_split == 3 ? 2 * (a->Cin >> 6) : 0};
    return dispatch<true>(g, cg, (hipStream_t)stream);
}

template <int BM, int BN, int WM, int WN, bool CONV, int BKT, int WK = 1>
int set_lds_attr() {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, WM, WN, CONV, BKT, WK>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<BM, BN, BKT, WM * WN * WK>());
    return e == hipSuccess ? 0 : (int)e;
}
template <int BM, int BN, int WM, int WN, int BKT, int WK = 1>
int set_lds_attr2() {
    int e = set_lds_attr<BM, BN, WM, WN, false, BKT, WK>();
    return e ? e : set_lds_attr<BM, BN, WM, WN, true, BKT, WK>();
}

extern "C" int gl_init_gemm(void) {
    int e;
    if ((e = set_lds_attr2<64, 128, 2, 2, 64>())) return e;
    if ((e = set_lds_attr2<128, 128, 2, 2, 64>())) return e;
    if ((e = set_lds_attr2<128, 160, 4, 1, 64>())) return e;
    if ((e = set_lds_attr2<256, 64, 4, 1, 64>())) return e;
    if ((e = set_lds_attr2<128, 160, 2, 1, 64, 2>())) return e;
    if ((e = set_lds_attr2<128, 128, 2, 1, 64, 2>())) return e;
    if ((e = set_lds_attr<64, 128, 1, 2, false, 64, 2>())) return e;
    if ((e = set_lds_attr2<256, 160, 4, 1, 32>())) return e;
    if ((e = set_lds_attr2<256, 128, 4, 1, 32>())) return e;
    if ((e = set_lds_attr<64, 128, 2, 2, false, 32>())) return e;
    if ((e = set_lds_attr<128, 128, 2, 2, false, 32>())) return e;
    if ((e = gl8_init())) return e;
    return 0;
}

// measurement hook: per-block cycle stamps of the timestamping 8-wave kernel (gl_set_option(32, 1)); 4 x uint64 per block
extern "C" int gl_debug_read(int what, void* dst, int64_t bytes) {
    if (what == 8) return gl8_read_stamps(dst, bytes);
    if (what == 9) {
        if (!dst || bytes < (int64_t)sizeof(uint64_t)) return GL_ERR_BAD_ARG;
        const uint64_t n = g8_launches.load(std::memory_order_relaxed);
        memcpy(dst, &n, sizeof n);
        return 0;
    }
    return GL_ERR_BAD_ARG;
}
