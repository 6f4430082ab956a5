// fp16 MFMA GEMM and implicit-GEMM 3x3 convolution for gfx950 (wave64, v_mfma_f32_32x32x16_f16).
//
//   out[M, N] = A[M, K] . W[N, K]^T  (+ fused epilogue)
//
// A is either a plain (optionally two-source, K-split) row-major matrix, or the implicit im2col view
// of an NHWC feature map for a 3x3 / pad 1 convolution (stride 1, stride 2, or nearest-2x-upsampled
// input).  A K-tile never straddles a filter tap because Cin % 64 == 0.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN (128x128, 128x160 or 256x64), MFMA 32x32 tiles,
// fp32 accumulators.  Operand tiles go global -> LDS directly (global_load_lds_dwordx4: a wave's 64
// lanes fill 1 KiB of consecutive LDS = 8 or 16 tile rows), so the LDS image is lane-linear and the
// bank-conflict swizzle is applied to the per-lane GLOBAL source chunk and, identically, to the
// fragment reads (chunk' = chunk ^ f(row); f = (row>>1)&7 for 128-byte rows, (row>>2)&3 for 64-byte
// rows: conflict-free for the 16-lane groups of ds_read_b128).  Masked lanes (rows past M/N, the conv
// halo) read a 16-byte zero page, since LDS-DMA cannot be predicated per lane.
//
// Two pipelines (gl_set_option key 1):
//   BK=64, 2 LDS stages : (default) tile t+1 is requested before the MFMAs of tile t; s_waitcnt vmcnt(0) +
//                         barrier per K-tile.  PMC on the level-0 convs: waves wait 39 % of their lifetime,
//                         MFMA pipe 31 % busy.
//   BK=32, 3 LDS stages : tiles t+1 and t+2 are in flight during the MFMAs of tile t; the wait before each
//                         barrier is a COUNTED vmcnt (only tile t must have landed) and the barrier is a raw
//                         s_barrier (a __syncthreads would drain the LDS-DMA queue).  Measured 15-25 % SLOWER
//                         than BK=64 on every shape of this UNet (twice the barriers and fragment-read
//                         restarts per K outweigh the deeper prefetch), so it is kept only as an A/B knob.
//
// The MFMA is issued "swapped" (weights as the row operand, activations as the column operand) and the
// epilogue restages the fp32 tile through LDS so that outputs, residuals and row-biases move as
// 16-byte row-contiguous accesses; GEGLU pairs x|gate columns that the weight packing interleaved.
#include "common.h"
#include "gligen_hip.h"

// 16 zero bytes in global memory: the source of masked lanes of the direct-to-LDS loads.
__device__ uint4 g_zero16[4];

namespace {

int g_opt_big_kind = 1;     // which 256-row variant g_opt_big selects: 0 = 8 waves BK 64 / 3-stage, 1 = 4 waves BK 32 / 2-stage
int g_opt_halo = 0;          // halo-resident conv kernel: 0 off, 1 auto (enough tiles), 2 whenever the geometry allows
int g_opt_halo_tiles = 64;
int g_opt_ksplit = 1;        // intra-block K-split variants (64-row wave tiles): 0 off, 1 auto (long K, no split-K), 2 always
int g_opt_dbg = 0;           // measurement-only loop ablation, see the NST == 2 main loop
int g_opt_geglu32 = 1;      // 1 = short-K GEGLU GEMMs use the 4-blocks/CU BK 32 variant
int g_opt_pipe = 0;          // 0 = BK 64 / 2-stage (default, faster), 1 = BK 32 / 3-stage counted-vmcnt pipeline
int g_opt_big = 300;          // problems with >= this many 256-row tiles use the 256-row variant (B=4: neutral; B=16: +5-7 %); 0 = off
int g_opt_small = 400;       // use 64x128 tiles when the 128-row grid has fewer tiles than this (0 = never)
int g_opt_splitk_tiles = 300; // split K only below this many tiles (plain GEMM) ...
int g_opt_splitk_tiles_conv = 450; // ... (conv)
int g_opt_splitk_nk = 16;    // ... and at least this many 64-wide K tiles
int g_opt_tile = 0;          // 0 = auto; 1 = force 128x128 (N >= 256); 2 = prefer 128x160 whenever N % 160 == 0

struct ConvGeom {
    const half_t* in;
    int B, Hin, Win, Cin, Hout, Wout, stride, ups;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const half_t* src, half_t* dst) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(src)),
        reinterpret_cast<__attribute__((address_space(3))) void*>(reinterpret_cast<uintptr_t>(dst)), 16, 0, 0);
}

template <int BM, int BN, int BKT, int NST, int NW = 4>
constexpr int lds_bytes() {
    constexpr int pipe = NST * (BM + BN) * BKT * (int)sizeof(half_t);
    constexpr int epi = NW * 32 * 68 * (int)sizeof(float);    // epilogue staging: NW waves x 32 rows x (64+4) fp32
    return pipe > epi ? pipe : epi;
}

// occupancy hint: the 3-stage BK=32 128x128 kernel needs 48 KiB of LDS (3 blocks/CU) but ~178 registers;
// asking for 3 waves/SIMD makes the compiler fit 170 so that the third block is actually resident.
template <int BM, int BN, int BKT, int NST, int WK = 1>
constexpr int min_waves() {
    if (WK > 1) return 2;
    if (BM == 128 && BN == 160 && BKT == 32 && NST == 2 && WK == 1) return 2;   // (2-wave 64x160 variant shares this key: needs 2)
    if (BKT != 32) return 1;
    if (BM * BN >= 256 * 128) return NST == 2 ? 2 : 1;
    if (BM * BN <= 128 * 128) return NST == 2 ? 4 : 3;
    return NST == 2 ? 3 : 1;
}

// WK = 2: intra-block K split.  The waves form two groups that own the SAME output rows/columns but alternate
// halves of every K-tile's k-steps, so a wave's tile is twice as tall (64 x BN instead of 32 x BN at 4 waves):
// (TM + TN) / (TM * TN) LDS fragment reads per MFMA drop from 1.2 to 0.7 (128x160) -- the CU's LDS port, shared by
// the LDS-DMA writes and the fragment reads, is what bounds the main loop (loop ablation).  The two
// partial accumulators are exchanged through LDS in the epilogue: each wave ends up finalising 32 rows.
//
// HALO = true (stride-1 3x3 convs whose 256-pixel row tile is a whole number of image rows): instead of one A tile per
// (channel block, tap) -- nine shifted copies of the same pixels, each fetched from L2 into LDS -- the block keeps ONE
// input patch per 64-channel block resident in LDS, (rows + 2) x (W + 2) pixels including the zero halo, and the nine
// taps read their A fragments from it at a shifted pixel offset.  The patch of the next channel block streams in
// during taps 1..7 (one 64-pixel pass per K-step).  L2 -> LDS traffic per channel block drops from
// 9 x (256 + 160) to 448 + 9 x 160 rows of 128 B (-50 %), LDS-DMA writes likewise; 8 waves, K-split wave tiles.
constexpr int PATCH_PX = 448;      // patch rows reserved in LDS: (256/W + 2) * (W + 2) <= 396 for W = 64, 32, 16
template <int BM, int BN, int WAVES_M, int WAVES_N, bool CONV, int BKT, int NST, int WK = 1, bool HALO = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WK, (min_waves<BM, BN, BKT, NST, WK>())) void gemm_kernel(gl_gemm_args p, ConvGeom cg, int splitk, int kt_per_split, int dbg) {
    constexpr int NTHR = 64 * WAVES_M * WAVES_N * WK;  // 4 waves (256 threads) or 8 waves (512 threads, 256-row tiles)
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    constexpr int CPR = BKT / 8;                 // 16-byte chunks per tile row (8 or 4)
    constexpr int RPP = NTHR / CPR;              // tile rows covered by one pass of the block's threads
    constexpr int RPW = 64 / CPR;                // tile rows covered by one wave instruction (1 KiB)
    constexpr int APASS = HALO ? PATCH_PX / RPP : (BM + RPP - 1) / RPP;
    constexpr int BPASS = (BN + RPP - 1) / RPP;
    constexpr int KSTEPS = BKT / 16;
    static_assert(WAVES_M * WAVES_N * WK == 2 || WAVES_M * WAVES_N * WK == 4 || WAVES_M * WAVES_N * WK == 8, "2, 4 or 8 waves");
    static_assert(WK == 1 || (WK == 2 && TM == 2 && KSTEPS % 2 == 0), "K-split: two groups, two 32-row tiles per wave");
    static_assert(BM % RPW == 0 && BN % RPW == 0, "whole wave instructions");
    static_assert(BKT == 128 || BKT == 64 || BKT == 32, "BK");
    static_assert(!HALO || (CONV && WK == 2 && BM == 256 && BKT == 64 && NST == 2 && NTHR == 512), "halo conv geometry");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* As = reinterpret_cast<half_t*>(smem);   // [NST][BM][BKT]
    half_t* Bs = As + (HALO ? 2 * PATCH_PX : NST * BM) * BKT;   // [NST][BN][BKT]; HALO: As = [2][PATCH_PX][64]

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
    const int tm = tile / nt;
    const int m0 = tm * BM;
    const int n0 = (tile - tm * nt) * BN;
    const int kt_begin = blockIdx.z * kt_per_split;
    const int kt_end = min(K / BKT, kt_begin + kt_per_split);
    const int nkt = kt_end - kt_begin;

    const int srow = tid / CPR;      // staging row within a pass
    const int skc = tid % CPR;       // LDS chunk slot within the row
    auto swz = [](int r) -> int { return BKT == 128 ? (r & 15) : (BKT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3)); };

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
        if constexpr (HALO) {
            // pass i stages patch pixels 64 i .. 64 i + 63 (this thread: pixel r, 16-byte chunk skc of its 64 channels)
            const int hw = cg.Hin * cg.Win;
            const int b = m0 / hw;
            const int oy0 = (m0 - b * hw) / cg.Win;
            const int pw = cg.Win + 2;
            const int py = r / pw;
            const int px = r - py * pw;
            const int iy = oy0 + py - 1, ix = px - 1;
            if (r < (BM / cg.Win + 2) * pw && iy >= 0 && iy < cg.Hin && ix >= 0 && ix < cg.Win) {
                aptr[i] = cg.in + ((size_t)(b * cg.Hin + iy) * cg.Win + ix) * cg.Cin + gc;
                amask |= 1u << i;
            }
        } else if constexpr (CONV) {
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
        bptr[i] = ok ? (Wg + (size_t)n * K + k_first + gc) : zsrc;
        if (ok) bmask |= 1u << i;
    }

    // number of LDS-DMA instructions THIS wave issues per tile (passes whose rows exist for this wave)
    int my_loads = 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) my_loads += (RPP * i + wave * RPW < BM) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) my_loads += (RPP * i + wave * RPW < BN) ? 1 : 0;

    auto issue_tile = [&](int kt, int buf) {
        const int k0 = kt * BKT;
        if constexpr (CONV) {
            // K is ordered (64-channel block, tap, channel): the 9 taps of one channel block are consecutive
            // K-tiles, so they re-read the same [pixels + halo] x 64-channel slab while it is L1/L2-hot.
            const int t64 = k0 >> 6;
            const int cblk = t64 / 9;
            const int tap = t64 - cblk * 9;
            const int ci0 = (cblk << 6) + (k0 & 63);
            const int ky = tap / 3;
            const int kx = tap - ky * 3;
            if constexpr (HALO) {
                // with the W tile of (cblk, tap) goes patch pass tap-1 of channel block cblk+1 (taps 1..7): its buffer,
                // (cblk+1) & 1, was last read while (cblk-1, 8) was computed, one barrier before (cblk, 1) is issued
                if (tap >= 1 && tap <= APASS && (cblk + 1) * 9 < kt_end) {
                    half_t* dst = As + (size_t)(((cblk + 1) & 1) * PATCH_PX + wave * RPW) * BKT;
                    const int coff = (cblk + 1) << 6;
#pragma unroll
                    for (int i = 0; i < APASS; ++i)
                        if (i == tap - 1)
                            glds16(((amask >> i) & 1u) ? aptr[i] + coff : zsrc, dst + (size_t)(RPP * i) * BKT);
                }
            } else if (!cg.ups) {
                const int off = ((ky - 1) * cg.Win + (kx - 1)) * cg.Cin + ci0;    // wave-uniform
#pragma unroll
                for (int i = 0; i < APASS; ++i) {
                    if (RPP * i + wave * RPW >= BM) continue;      // wave-uniform: pass has no rows for this wave
                    const half_t* src = ((cmask[i] >> tap) & 1u) ? (aptr[i] + off) : zsrc;
                    glds16(src, As + (size_t)(buf * BM + RPP * i + wave * RPW) * BKT);
                }
            } else {
#pragma unroll
                for (int i = 0; i < APASS; ++i) {
                    if (RPP * i + wave * RPW >= BM) continue;
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
#pragma unroll
                for (int i = 0; i < APASS; ++i) {
                    const int r = srow + RPP * i;
                    const int m = m0 + r;
                    if ((r < BM) && (m < M)) aptr[i] = A2g + (size_t)m * p.lda2 + ((skc ^ swz(r)) << 3);
                }
            }
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                if (RPP * i + wave * RPW >= BM) continue;
                glds16(aptr[i], As + (size_t)(buf * BM + RPP * i + wave * RPW) * BKT);
                aptr[i] += ((amask >> i) & 1u) ? BKT : 0;
            }
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            if (RPP * i + wave * RPW >= BN) continue;
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

    // HALO: patch row of output pixel q = wm*64 + mi*32 + frow at tap (0,0): (q / W) * (W + 2) + q % W
    int prow0[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        prow0[mi] = 0;
        if constexpr (HALO) {
            const int q = wm * (TM * 32) + mi * 32 + frow;
            const int dy = q / cg.Win;
            prow0[mi] = dy * (cg.Win + 2) + (q - dy * cg.Win);
        }
    }
    auto compute_tile = [&](int buf, int kt) {
        int abase = 0;                                   // HALO: first LDS row of this tile's A operand
        if constexpr (HALO) {
            const int cblk = kt / 9;
            const int tap = kt - cblk * 9;
            const int ky = tap / 3;
            abase = (cblk & 1) * PATCH_PX + ky * (cg.Win + 2) + (tap - ky * 3);
        }
        if (dbg & 4) __builtin_amdgcn_s_setprio(1);      // A/B: priority over the co-resident block's DMA issue
#pragma unroll
        for (int kq = 0; kq < KSTEPS / WK; ++kq) {
            half8_t xf[TM], wf[TN];
            const int ks = wk * (KSTEPS / WK) + kq;
            const int c = ks * 2 + fhi;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                if constexpr (HALO) {
                    const int r = abase + prow0[mi];     // swizzle follows the LDS row, i.e. the patch pixel
                    xf[mi] = *reinterpret_cast<const half8_t*>(As + (size_t)r * BKT + ((c ^ swz(r)) << 3));
                } else {
                    const int r = wm * (TM * 32) + mi * 32 + frow;
                    xf[mi] = *reinterpret_cast<const half8_t*>(As + (size_t)(buf * BM + r) * BKT + ((c ^ swz(r)) << 3));
                }
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
        if (dbg & 4) __builtin_amdgcn_s_setprio(0);
    };

    if constexpr (NST == 2) {
        // dbg (gl_set_option 12, measurement only -- results are garbage): bit 0 skips the MFMA / fragment-read
        // half of the loop, bit 1 skips the global -> LDS half; isolates which side bounds a shape
        if constexpr (HALO) {
            // whole patch of the first channel block of this K range
            const int cb0 = kt_begin / 9;
#pragma unroll
            for (int i = 0; i < APASS; ++i)
                glds16(((amask >> i) & 1u) ? aptr[i] + (cb0 << 6) : zsrc,
                       As + (size_t)((cb0 & 1) * PATCH_PX + RPP * i + wave * RPW) * BKT);
        }
        issue_tile(kt_begin, 0);
        wait_vmcnt<0>();
        __syncthreads();
        for (int it = 0; it < nkt; ++it) {
            const int buf = it & 1;
            if (it + 1 < nkt && !(dbg & 2)) issue_tile(kt_begin + it + 1, buf ^ 1);
            if (!(dbg & 1)) compute_tile(buf, kt_begin + it);
            wait_vmcnt<0>();
            __syncthreads();
        }
    } else {
        // NST-stage ring (NST = 3 or 4), prefetch distance D = NST - 1 tiles.  Invariant at the top of iteration `it`:
        // tiles it .. it+D-1 have been requested.  The counted wait leaves the newest D-1 tiles' loads (my_loads each,
        // per wave) in flight; loads retire in issue order, so tile `it` has landed for this wave, and after the
        // barrier for all waves.  The buffer refilled right after the barrier, (it+D) % NST, was last read in
        // iteration it-1, which every wave has finished before arriving at this barrier.
        constexpr int D = NST - 1;
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < nkt) issue_tile(kt_begin + j, j);
        int buf = 0;
        for (int it = 0; it < nkt; ++it) {
            if (it + D - 1 < nkt) {
                switch (my_loads * (D - 1)) {
                    case 2: wait_vmcnt<2>(); break;
                    case 3: wait_vmcnt<3>(); break;
                    case 4: wait_vmcnt<4>(); break;
                    case 5: wait_vmcnt<5>(); break;
                    case 6: wait_vmcnt<6>(); break;
                    case 7: wait_vmcnt<7>(); break;
                    case 8: wait_vmcnt<8>(); break;
                    case 10: wait_vmcnt<10>(); break;
                    case 12: wait_vmcnt<12>(); break;
                    case 14: wait_vmcnt<14>(); break;
                    case 16: wait_vmcnt<16>(); break;
                    case 18: wait_vmcnt<18>(); break;
                    default: wait_vmcnt<0>(); break;
                }
            } else {
                wait_vmcnt<0>();          // tail: fewer than D-1 younger tiles exist; draining is exact enough
            }
            __builtin_amdgcn_s_barrier();
            int nbuf = buf + D;
            if (nbuf >= NST) nbuf -= NST;
            if (it + D < nkt) issue_tile(kt_begin + it + D, nbuf);
            compute_tile(buf, kt_begin + it);
            buf = (buf == NST - 1) ? 0 : buf + 1;
        }
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
    const half_t* __restrict__ res = reinterpret_cast<const half_t*>(p.res);
    const half_t* __restrict__ rowbias = reinterpret_cast<const half_t*>(p.rowbias);
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

    // Row-major fp16 output: the fp32 accumulator tile of each wave is staged through LDS (the operand
    // buffers are dead by now), 32 rows x 64 columns (two MFMA tiles) at a time, so that every lane then
    // owns 8 CONSECUTIVE channels of one row: residual / row-bias reads and the output stores are 16-byte
    // accesses, 128 contiguous bytes per row per 8 lanes, instead of 8-byte pieces scattered over 32 rows
    // straight from the MFMA register layout.  For GEGLU a pass is exactly one [x(32) | gate(32)] pair.
    constexpr int EPS = 64 + 4;                            // padded fp32 row stride (conflict-free float4 writes)
    constexpr int NPASS = (TN + 1) / 2;
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
            if (geglu) {
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
                        half8_t o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float a = xv[j], b = gv[j];
                            if (bias) { a += bias[nx + j]; b += bias[nx + 32 + j]; }
                            o[j] = (half_t)(a * gelu_erf_f(b));
                        }
                        st16(outp + (size_t)m * p.ldc + (nbase >> 1) + pc, *reinterpret_cast<uint4*>(&o));
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
                        if (bias) {
                            const float4 b0 = *reinterpret_cast<const float4*>(bias + n);
                            const float4 b1 = *reinterpret_cast<const float4*>(bias + n + 4);
                            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                        }
                        if (epi == GL_EPI_SILU) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
                        } else if (epi == GL_EPI_RES || epi == GL_EPI_GATE_RES) {
                            uint4 raw = ld16(res + (size_t)m * p.ldres + n);
                            const half8_t rv = *reinterpret_cast<half8_t*>(&raw);
                            if (epi == GL_EPI_RES) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
                            } else {
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[j] = (float)rv[j] + gate * v[j];
                            }
                        } else if (epi == GL_EPI_ROWBIAS) {
                            const int sidx = m / p.rows_per_sample;
                            uint4 raw = ld16(rowbias + (size_t)sidx * p.ld_rowbias + n);
                            const half8_t rv = *reinterpret_cast<half8_t*>(&raw);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
                        }
                        half8_t o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
                        st16(outp + (size_t)m * p.ldc + n, *reinterpret_cast<uint4*>(&o));
                    }
                }
            }
        }
    }
}

// Sums the split-K partial tiles and applies the epilogue (fp16 row-major outputs only).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(gl_gemm_args p, int splitk) {
    const int M = p.M, N = p.N;
    const int nq = N / 4;
    const size_t total = (size_t)M * nq;
    const float* ws = reinterpret_cast<const float*>(p.workspace);
    const half_t* res = reinterpret_cast<const half_t*>(p.res);
    const half_t* rowbias = reinterpret_cast<const half_t*>(p.rowbias);
    half_t* out = reinterpret_cast<half_t*>(p.out);
    float gate = 1.0f;
    if (p.epi == GL_EPI_GATE_RES) gate = p.gate[0];
    const size_t zstride = (size_t)M * N;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * 256u) {
        const int m = (int)(idx / (unsigned)nq);
        const int n = (int)(idx - (unsigned)m * (unsigned)nq) * 4;
        const float* src = ws + (size_t)m * N + n;
        float4 a = *reinterpret_cast<const float4*>(src);
        // slices are added in index order (deterministic); 4 independent loads in flight per step instead of a
        // load -> wait -> add chain per slice
        int z = 1;
        for (; z + 3 < splitk; z += 4) {
            const float4 b0 = *reinterpret_cast<const float4*>(src + (size_t)z * zstride);
            const float4 b1 = *reinterpret_cast<const float4*>(src + (size_t)(z + 1) * zstride);
            const float4 b2 = *reinterpret_cast<const float4*>(src + (size_t)(z + 2) * zstride);
            const float4 b3 = *reinterpret_cast<const float4*>(src + (size_t)(z + 3) * zstride);
            a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
            a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
            a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
            a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
        }
        for (; z < splitk; ++z) {
            const float4 b = *reinterpret_cast<const float4*>(src + (size_t)z * zstride);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float v[4] = {a.x, a.y, a.z, a.w};
        if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (p.epi == GL_EPI_SILU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
        } else if (p.epi == GL_EPI_RES) {
            const half4_t rv = *reinterpret_cast<const half4_t*>(res + (size_t)m * p.ldres + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
        } else if (p.epi == GL_EPI_GATE_RES) {
            const half4_t rv = *reinterpret_cast<const half4_t*>(res + (size_t)m * p.ldres + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (float)rv[j] + gate * v[j];
        } else if (p.epi == GL_EPI_ROWBIAS) {
            const int sidx = m / p.rows_per_sample;
            const half4_t rv = *reinterpret_cast<const half4_t*>(rowbias + (size_t)sidx * p.ld_rowbias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
        }
        half4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
        *reinterpret_cast<half4_t*>(out + (size_t)m * p.ldc + n) = o;
    }
}

// How many K slices: only when the tile grid underfills the chip and K is long enough to amortise
// the fp32 partial round trip.
inline int choose_splitk(const gl_gemm_args& g, int tiles, bool conv) {
    const int nk = g.K / 64;
    if (!g.workspace || g.epi == GL_EPI_GEGLU || g.out_mode != GL_OUT_F16_ROWMAJOR) return 1;
    // convs (long K, weights streamed once per row tile) profit up to ~1.7 tiles per CU: the 32x32-level
    // convs launch exactly 256 tiles and went 604 -> 694 TF/s with 2 K-slices; plain GEMMs only below ~300
    if (tiles >= (conv ? g_opt_splitk_tiles_conv : g_opt_splitk_tiles) || nk < g_opt_splitk_nk) return 1;
    int s = (480 + tiles - 1) / tiles;
    if (s > nk / 8) s = nk / 8;
    if (s > 16) s = 16;
    while (s > 1 && (int64_t)s * g.M * g.N * 4 > g.workspace_bytes) --s;
    return s < 2 ? 1 : s;
}

template <int BM, int BN, int WM, int WN, bool CONV, int BKT, int NST, int WK = 1>
int launch(const gl_gemm_args& g, const ConvGeom& cg, hipStream_t st) {
    const int mt = gl_cdiv(g.M, BM), nt = gl_cdiv(g.N, BN);
    const int nk = g.K / BKT;
    int splitk = choose_splitk(g, mt * nt, CONV);
    int kper = gl_cdiv(nk, splitk);
    const int zs = gl_cdiv(nk, kper);          // slices that actually have work
    dim3 grid(mt * nt, 1, zs);
    constexpr int lds = lds_bytes<BM, BN, BKT, NST, WM * WN * WK>();
    gemm_kernel<BM, BN, WM, WN, CONV, BKT, NST, WK><<<grid, dim3(64 * WM * WN * WK), lds, st>>>(g, cg, zs, kper, g_opt_dbg);
    GL_CHECK_LAUNCH();
    if (zs > 1) {
        const size_t total = (size_t)g.M * (g.N / 4);
        int nblk = (int)((total + 255) / 256);
        if (nblk > 2048) nblk = 2048;
        splitk_reduce_kernel<<<dim3(nblk), dim3(256), 0, st>>>(g, zs);
        GL_CHECK_LAUNCH();
    }
    return 0;
}

// halo-resident 3x3 conv (see gemm_kernel, HALO): 256-pixel row tiles, 8 waves, one block per CU
constexpr int lds_bytes_halo(int bn) { return (2 * PATCH_PX + 2 * bn) * 64 * (int)sizeof(half_t); }

inline bool halo_conv_ok(const gl_gemm_args& g, const ConvGeom& cg) {
    const int hw = cg.Hin * cg.Win;
    return cg.stride == 1 && !cg.ups && cg.Hin == cg.Hout && cg.Win == cg.Wout && cg.Win >= 16 && (256 % cg.Win) == 0 &&
           (hw % 256) == 0 && (256 / cg.Win + 2) * (cg.Win + 2) <= PATCH_PX && (cg.Cin % 64) == 0 && (g.N % 160) == 0 &&
           g.out_mode == GL_OUT_F16_ROWMAJOR && g.epi != GL_EPI_GEGLU;
}

int launch_halo(const gl_gemm_args& g, const ConvGeom& cg, hipStream_t st) {
    constexpr int BN = 160;
    const int mt = g.M / 256, nt = gl_cdiv(g.N, BN);
    const int ncb = cg.Cin / 64;                       // K-tiles come in groups of 9 taps per 64-channel block
    gl_gemm_args gk = g;
    int splitk = choose_splitk(gk, mt * nt, true);
    if (splitk > ncb) splitk = ncb;
    const int cb_per = gl_cdiv(ncb, splitk);
    const int zs = gl_cdiv(ncb, cb_per);
    dim3 grid(mt * nt, 1, zs);
    gemm_kernel<256, BN, 4, 1, true, 64, 2, 2, true><<<grid, dim3(512), lds_bytes_halo(BN), st>>>(g, cg, zs, 9 * cb_per, g_opt_dbg);
    GL_CHECK_LAUNCH();
    if (zs > 1) {
        const size_t total = (size_t)g.M * (g.N / 4);
        int nblk = (int)((total + 255) / 256);
        if (nblk > 2048) nblk = 2048;
        splitk_reduce_kernel<<<dim3(nblk), dim3(256), 0, st>>>(g, zs);
        GL_CHECK_LAUNCH();
    }
    return 0;
}

template <bool CONV, int BKT, int NST>
int dispatch_shape(const gl_gemm_args& g, const ConvGeom& cg, hipStream_t st) {
    // tile shape: 128x160 (4 waves x 32x160) when it divides N exactly -- N = 320/640/960/... are the
    // channel widths of this UNet and 128-wide tiles would waste up to 17 % of the MFMA work there
    // (convs, whose K is long, prefer it even when 128 also divides N: measured 442 vs 408 and 607 vs 536 TF/s
    // at the 32x32 and 16x16 levels); otherwise 128x128; 256x64 only for narrow outputs.
    const bool geglu = (g.epi == GL_EPI_GEGLU);
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
    // intra-block K-split (64-row wave tiles, 0.7-0.75 LDS fragment reads per MFMA): +7-18 % on every conv of the
    // UNet (with or without split-K slices: the two K groups are combined in LDS before a partial slice is written)
    // and on K >= 1024 GEMMs; its accumulator exchange in the epilogue costs 5-20 % on short K (the K = 320 / 640
    // projections), which stay on the 4 x (32 x BN) kernels (per-shape A/B in DESIGN.md)
    if constexpr (BKT == 64 && NST == 2) {
        // experiment (key 7 = tile threshold, key 9 = 2): 256-row K-split tile, 8 waves, 3-stage ring -- one block per
        // CU with TWO K-tiles of loads in flight (106 KB) instead of two blocks with one each (74 KB)
        if (g_opt_big && g_opt_big_kind == 2 && g.out_mode == GL_OUT_F16_ROWMAJOR && (shape == 0 || shape == 1) && g.M >= 256) {
            const long t256 = (long)gl_cdiv(g.M, 256) * gl_cdiv(g.N, shape == 1 ? 160 : 128);
            if (t256 >= g_opt_big) {
                if (shape == 1) return launch<256, 160, 4, 1, CONV, 64, 3, 2>(g, cg, st);
                return launch<256, 128, 4, 1, CONV, 64, 3, 2>(g, cg, st);
            }
        }
        if (g_opt_ksplit && g.out_mode == GL_OUT_F16_ROWMAJOR && (shape == 0 || shape == 1)) {
            const int nk = g.K / 64;
            bool use = (g_opt_ksplit == 2);
            // (128-wide conv tiles only occur in the VAE decoder, M = 0.26-1 M pixels x 128/256 channels: measured 2 % slower)
            if (g_opt_ksplit == 1) use = CONV ? (shape == 1 && nk >= 20) : (nk >= 16);
            if (use) {
                if (shape == 1) return launch<128, 160, 2, 1, CONV, 64, 2, 2>(g, cg, st);
                return launch<128, 128, 2, 1, CONV, 64, 2, 2>(g, cg, st);
            }
        }
    }
    // deep-prefetch variant for the big level-0 / wide-N problems: 8 waves share a 256-row tile, BK 64 with a
    // 3-stage ring (loads get TWO tile-times to land instead of one) at the same 2 waves/SIMD occupancy
    if constexpr (BKT == 64 && NST == 2) {
        if (g_opt_big && shape != 2 && g.M >= 256) {
            const int bn = (shape == 1) ? 160 : 128;
            const long t256 = (long)gl_cdiv(g.M, 256) * gl_cdiv(g.N, bn);
            if (t256 >= g_opt_big) {
                if (g_opt_big_kind == 1) {   // 4 waves with 64-row wave tiles: 0.7-0.75 LDS fragment reads per MFMA
                    if (shape == 1) return launch<256, 160, 4, 1, CONV, 32, 2>(g, cg, st);
                    if (shape == 0) return launch<256, 128, 4, 1, CONV, 32, 2>(g, cg, st);
                }
                if (shape == 1) return launch<256, 160, 8, 1, CONV, 64, 3>(g, cg, st);
                if (shape == 0) return launch<256, 128, 4, 2, CONV, 64, 3>(g, cg, st);
            }
        }
    }
    if constexpr (BKT == 64 && NST == 2 && !CONV) {
        // K-split for the small-tile shape too (64x64 instead of 32x64 wave tiles): 22.4 -> 20.4 us at
        // M = 2048, N = K = 1280; neutral at K = 640, which stays on the plain kernel
        if (shape == 3 && (g_opt_ksplit == 2 || (g_opt_ksplit == 1 && g.K >= 1024)))
            return launch<64, 128, 1, 2, false, 64, 2, 2>(g, cg, st);
    }
    if (shape == 3) return launch<64, 128, 2, 2, CONV, BKT, NST>(g, cg, st);
    if (shape == 1) return launch<128, 160, 4, 1, CONV, BKT, NST>(g, cg, st);
    if (shape == 0) return launch<128, 128, 2, 2, CONV, BKT, NST>(g, cg, st);
    if constexpr (BKT == 128) return GL_ERR_UNSUPPORTED; else return launch<256, 64, 4, 1, CONV, BKT, NST>(g, cg, st);
}

template <bool CONV>
int dispatch(const gl_gemm_args& g, const ConvGeom& cg, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % 64) != 0) return GL_ERR_BAD_ARG;
    if (g.out_mode == GL_OUT_F16_ROWMAJOR && ((g.N % 8) != 0 || (g.ldc % 8) != 0)) return GL_ERR_BAD_ARG;
    if (g.res != nullptr && (g.ldres % 8) != 0) return GL_ERR_BAD_ARG;
    if (g.rowbias != nullptr && (g.ld_rowbias % 8) != 0) return GL_ERR_BAD_ARG;
    if (g.epi == GL_EPI_GEGLU && (g.N % 64) != 0) return GL_ERR_BAD_ARG;
    if (g.a2 != nullptr && (g.ksplit % 64) != 0) return GL_ERR_BAD_ARG;
    if ((g.epi == GL_EPI_RES || g.epi == GL_EPI_GATE_RES) && g.res == nullptr) return GL_ERR_BAD_ARG;
    if (g.epi == GL_EPI_GATE_RES && g.gate == nullptr) return GL_ERR_BAD_ARG;
    if (g.epi == GL_EPI_ROWBIAS && (g.rowbias == nullptr || g.rows_per_sample <= 0)) return GL_ERR_BAD_ARG;
    if constexpr (CONV) {
        if (g_opt_halo && halo_conv_ok(g, cg) && (g_opt_halo == 2 || (g.M / 256) * gl_cdiv(g.N, 160) >= g_opt_halo_tiles))
            return launch_halo(g, cg, st);
    }
    if (g_opt_pipe == 1) return dispatch_shape<CONV, 32, 3>(g, cg, st);
    if (g_opt_pipe == 3) return dispatch_shape<CONV, 32, 2>(g, cg, st);
    if (g_opt_pipe == 5 && g.out_mode == GL_OUT_F16_ROWMAJOR && g.epi != GL_EPI_GEGLU && (g.N % 160) == 0) {
        // experiment: 2-wave blocks with 64 x 160 wave tiles (0.7 fragment reads per MFMA WITHOUT a K-split exchange),
        // BK 32 so that four blocks (8 waves) fit a CU
        return launch<128, 160, 2, 1, CONV, 32, 2>(g, cg, st);
    }
    if (g_opt_pipe == 4 && g.out_mode == GL_OUT_F16_ROWMAJOR && g.epi != GL_EPI_GEGLU && (g.N % 160) == 0 && g.K >= 1024) {
        // experiment: K-split tile with BK 32 and a 4-stage ring (3 sub-tiles = 96 K-columns in flight per block)
        return launch<128, 160, 2, 1, CONV, 32, 4, 2>(g, cg, st);
    }
    // GEGLU with a short K (levels 0/1: K = 320/640, 5-10 k-tiles) spends a large share of each block in its
    // erf epilogue; BK 32 / 2-stage needs 35 KiB of LDS and 114 registers, so 4 blocks/CU are resident and
    // one block's epilogue overlaps the others' main loops (measured 150 -> 135 us and 109 -> 100 us; long-K
    // GEMMs and convs lose 10-20 % to the doubled barrier count, so they stay on BK 64)
    if (!CONV && g_opt_pipe == 0 && g_opt_geglu32 && g.epi == GL_EPI_GEGLU && g.K <= 640)
        return dispatch_shape<CONV, 32, 2>(g, cg, st);
    if constexpr (!CONV) {
        if (g_opt_pipe == 2 && (g.K % 128) == 0 && g.N >= 256) return dispatch_shape<false, 128, 2>(g, cg, st);
    }
    return dispatch_shape<CONV, 64, 2>(g, cg, st);
}

}  // namespace

extern "C" int gl_gemm(const gl_gemm_args* a, void* stream) {
    if (!a || !a->a || !a->w || !a->out) return GL_ERR_BAD_ARG;
    ConvGeom cg{};
    return dispatch<false>(*a, cg, (hipStream_t)stream);
}

extern "C" int gl_conv3x3(const gl_conv_args* a, void* stream) {
    if (!a || !a->in || !a->g.w || !a->g.out) return GL_ERR_BAD_ARG;
    if ((a->Cin % 64) != 0) return GL_ERR_BAD_ARG;
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

template <int BM, int BN, int WM, int WN, int BKT, int NST, int WK = 1>
int set_lds_attr() {
    hipError_t e;
    const int lds = lds_bytes<BM, BN, BKT, NST, WM * WN * WK>();
    e = hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, WM, WN, false, BKT, NST, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, WM, WN, true, BKT, NST, WK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    return 0;
}

template <int BM, int BN, int WM, int WN, int BKT, int NST, int WK = 1>
int set_lds_attr_plain() {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, WM, WN, false, BKT, NST, WK>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<BM, BN, BKT, NST, WM * WN * WK>());
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int gl_init_gemm(void) {
    int e;
    if ((e = set_lds_attr_plain<64, 128, 2, 2, 128, 2>())) return e;
    if ((e = set_lds_attr_plain<128, 128, 2, 2, 128, 2>())) return e;
    if ((e = set_lds_attr_plain<128, 160, 4, 1, 128, 2>())) return e;
    if ((e = set_lds_attr<256, 160, 8, 1, 64, 3>())) return e;
    if ((e = set_lds_attr<256, 128, 4, 2, 64, 3>())) return e;
    if ((e = set_lds_attr<64, 128, 2, 2, 32, 3>())) return e;
    if ((e = set_lds_attr<64, 128, 2, 2, 64, 2>())) return e;
    if ((e = set_lds_attr<64, 128, 2, 2, 32, 2>())) return e;
    if ((e = set_lds_attr<128, 128, 2, 2, 32, 2>())) return e;
    if ((e = set_lds_attr<128, 160, 4, 1, 32, 2>())) return e;
    if ((e = set_lds_attr<256, 64, 4, 1, 32, 2>())) return e;
    if ((e = set_lds_attr<256, 160, 4, 1, 32, 2>())) return e;
    if ((e = set_lds_attr<256, 128, 4, 1, 32, 2>())) return e;
    if ((e = set_lds_attr<128, 128, 2, 2, 32, 3>())) return e;
    if ((e = set_lds_attr<128, 160, 4, 1, 32, 3>())) return e;
    if ((e = set_lds_attr<256, 64, 4, 1, 32, 3>())) return e;
    if ((e = set_lds_attr<128, 128, 2, 2, 64, 2>())) return e;
    if ((e = set_lds_attr<128, 160, 4, 1, 64, 2>())) return e;
    if ((e = set_lds_attr<256, 64, 4, 1, 64, 2>())) return e;
    if ((e = set_lds_attr<128, 160, 2, 1, 64, 2, 2>())) return e;
    if ((e = set_lds_attr<128, 128, 2, 1, 64, 2, 2>())) return e;
    if ((e = set_lds_attr<128, 160, 2, 1, 32, 2>())) return e;
    if ((e = set_lds_attr_plain<64, 128, 1, 2, 64, 2, 2>())) return e;
    if ((e = set_lds_attr<256, 160, 4, 1, 64, 3, 2>())) return e;
    if ((e = set_lds_attr<256, 128, 4, 1, 64, 3, 2>())) return e;
    if ((e = set_lds_attr<128, 160, 2, 1, 32, 4, 2>())) return e;
    {
        hipError_t he = hipFuncSetAttribute((const void*)gemm_kernel<256, 160, 4, 1, true, 64, 2, 2, true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_halo(160));
        if (he != hipSuccess) return (int)he;
    }
    return 0;
}

extern "C" int gl_set_option_gemm(int key, int value) {
    if (key == 1) { g_opt_pipe = value; return 0; }
    if (key == 2) { g_opt_tile = value; return 0; }
    if (key == 4) { g_opt_small = value; return 0; }
    if (key == 7) { g_opt_big = value; return 0; }
    if (key == 8) { g_opt_geglu32 = value; return 0; }
    if (key == 9) { g_opt_big_kind = value; return 0; }
    if (key == 12) { g_opt_dbg = value; return 0; }
    if (key == 13) { g_opt_ksplit = value; return 0; }
    if (key == 14) { g_opt_halo = value; return 0; }
    if (key == 15) { g_opt_halo_tiles = value; return 0; }
    if (key == 5) { g_opt_splitk_tiles = value; g_opt_splitk_tiles_conv = value; return 0; }
    if (key == 6) { g_opt_splitk_nk = value; return 0; }
    return GL_ERR_BAD_ARG;
}
