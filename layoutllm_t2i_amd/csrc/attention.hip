// Flash-style fused attention for gfx950: out = softmax(scale * Q K^T) V without the N x N tensor.
//
// Replaces SelfAttention.forward (attention.py:164-176) and CrossAttention.forward (:128-141), which
// materialise `sim` as (B*8) x N x N fp32 (2.1 GB at the 64x64 level for B=4).
//
// Work decomposition: one block per (128- or 256-query slab, head, batch), 4 or 8 waves, each wave owns 32
// queries; the 1-D grid is remapped so that all slabs of one (batch, head) run on the same XCD / L2.
// Per 64-key tile the block stages K [64][d] and V^T [d][64] in LDS (register-prefetched one tile
// ahead), then every wave runs
//   S^T = K . Q^T   as mfma_32x32x16(A = K rows, B = Q rows): lane (q = lane&31, hi = lane>>5) ends up
//                   with 32 scores of ITS query (keys (r&3) + 8*(r>>2) + 4*hi of each 32-key half), so
//                   the row max / row sum are 31 in-lane ops + one cross-half shuffle (lane ^ 32);
//   O^T += V^T . P^T  as mfma_32x32x16(A = V^T rows (head-dim), B = P rows (queries)): the K-index of
//                   this product is the key; MFMA only needs A and B to agree on which key sits in
//                   (lane-half, element e), so P is fed STRAIGHT from the score registers (key order
//                   (e&3) + 8*(e>>2) + 4*hi) and V^T is read from LDS in that same order (two 8-byte
//                   reads).  P never touches LDS and no cross-lane permutation is needed.
// Online softmax in fp32 with exp2 (scale * log2(e) folded); O rescaled per tile; fp16 P and V.
// Head dims 40/80/160 are zero-padded to 48/80/160 for QK^T (K = 16 per MFMA) and to 64/96/160 rows
// of V^T (32 per MFMA tile).
#include "common.h"
#include "gligen_hip.h"
#include "opts.h"
#include <type_traits>

namespace {

#define g_attn_qt2 gl_opt(3)  // default 0;          // block shape: 0 auto, 3 always 8 waves, 4 always 4 waves
#define g_attn_padmax gl_opt(29)  // default 1;       // running max carried in the operands' padding column where the head dim has one (gl_set_option 29; 0 = FMA path)
#define g_attn_split_var gl_opt(53)  // split-fp16 attention, d <= 48: 0 = software-pipelined kernel (d = 32 / 40 / 48), 2 = the round-5 8-wave kernel, 1 = its 4-wave form, two blocks per CU (A/B)
#define g_attn_setprio gl_opt(10)  // default -1;     // s_setprio(1) around the MFMA clusters: -1 auto (head dim <= 48: +6 %; d = 80: -4 %), 0 off, 1 on
constexpr int KT = 64;          // keys per tile
constexpr int VSTR2 = KT + 8;   // main kernel: 144 B rows = 9 x 16 B, conflict-free 16-byte fragment reads

// build knobs of attn_split_pipe_kernel (A/B builds: csrc/build.py --variant)
#ifndef ATTN_PIPE_KRING
#define ATTN_PIPE_KRING 2
#endif
#ifndef ATTN_PIPE_SLOTS
#define ATTN_PIPE_SLOTS 4     // LDS slots per operand: 4 = tiles staged one iteration further ahead, ONE BARRIER PER TWO TILES (the waves of a block may drift by a
#endif                        // tile); 2 = one barrier per tile (A/B)
#ifndef ATTN_PIPE_PRIO
#define ATTN_PIPE_PRIO 0      // 1 = static s_setprio 1 for the second-dispatched half of the block (waves 4-7), A/B
#endif
#ifndef ATTN_PIPE_DBG
#define ATTN_PIPE_DBG 0       // timing probes (results invalid): bit 0 = no staging stores / barrier in the loop, bit 1 = one K fragment group reused for all Q.K^T MFMAs,
#endif                        // bit 2 = no staging global loads

// PRE: the softmax scale AND log2(e) are already folded into Q (q_prescaled: the packer folds them into the q projection
// weights), so the logits leave the MFMA in exp2 units.  The running max is then subtracted by the MFMA itself -- the
// accumulator is initialised with -m instead of 0 -- and the common (no-rescale) path of the online softmax is ONE v_exp
// per score instead of FMA + v_exp: the kernel is VALU-bound (PMC: VALU 68 % busy vs MFMA 43 % at d = 40), and this
// removes 32 of its ~146 VALU instructions per 64-key tile.
// PRE == 2 (head dims with a spare padded column, d % 16 == 8: d = 40 in this UNet): the running max rides in the PADDING of
// the operands instead of the accumulator init -- K's pad column d holds 1.0 for every key, Q's pad column d holds -m_run, so
// the QK^T MFMA that is issued anyway delivers logit - m_run; no extra registers (the C-init form costs 16 and an occupancy
// step at d = 40).  -m_run sits in an fp16 operand, so m_run is kept fp16-representable; softmax is invariant to WHICH
// reference value is subtracted as long as P, the row sum and the rescale factors all use the same one.
template <int DQK, int QT, int NW, int PRE>
__global__ __launch_bounds__(64 * NW) void attn_kernel(gl_attn_args p, int flags) {
    // NW = waves per block: 4 (128 queries share each staged K/V tile) or 8 (256 queries: half the L2 -> LDS
    // traffic per query at the same registers per wave)
    constexpr int NTHR = 64 * NW;
    const bool SETPRIO = (flags & 1) != 0;
    // QT = 32-query sub-tiles per wave: 2 for small head dims (256 queries per block: every K/V tile
    // staged in LDS and every K/V fragment read from LDS is reused twice), 1 where the O accumulator
    // (NDT x 16 registers per sub-tile) is too large.
    constexpr int NKS = DQK / 16;            // MFMA k-steps over the head dim
    constexpr int NDT = (DQK + 31) / 32;     // 32-wide head-dim tiles of O
    constexpr int KSTR = DQK + 8;            // K LDS row stride in halfs (odd number of 16-byte slots)
    constexpr int KCH = DQK / 8;             // 16-byte chunks per K row
    constexpr int K_ITEMS = KT * KCH;        // chunks in a K tile
    constexpr int K_PER_T = (K_ITEMS + NTHR - 1) / NTHR;
    constexpr int V_ITEMS = NDT * 32 * (KT / 8);
    constexpr int V_PER_T = (V_ITEMS + NTHR - 1) / NTHR;
    constexpr int QBLK = NW * 32 * QT;        // queries per block
    // If the padded head-dim tile has a spare row, V^T row OC is kept at 1.0 so that the P.V MFMA also
    // produces the softmax denominator (sum_k P) in O column OC: saves 32 VALU adds per tile.
    constexpr bool ONES = (NDT * 32 > DQK);
    constexpr int OC = NDT * 32 - 1;

    // double-buffered K / V^T tiles: one barrier per 64-key tile
    constexpr int KBUF = KT * KSTR;
    constexpr int VBUF = NDT * 32 * VSTR2;
    __shared__ __attribute__((aligned(16))) half_t Ksm[2 * KBUF];
    __shared__ __attribute__((aligned(16))) half_t Vsm[2 * VBUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ql = lane & 31;
    const int hi = lane >> 5;
    // 1-D grid, XCD-aware: hardware block id L goes to XCD L % 8, each with its own 4 MB L2.  Give every XCD a
    // contiguous range of logical blocks (bijective also when the count is not a multiple of 8) so that all
    // query blocks of one (batch, head) -- which stream the same K / V^T -- hit the same L2.
    const int nqb = (p.Nq + QBLK - 1) / QBLK;
    int logical;
    {
        const int total = gridDim.x, L = blockIdx.x;
        const int xcd = L & 7, qd = total >> 3, rm = total & 7;
        logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (L >> 3);
    }
    const int qb = logical % nqb;
    const int bh = logical / nqb;
    const int h = bh % p.H;
    const int b = bh / p.H;
    const int q0 = qb * QBLK + wave * (32 * QT);
    const int d = p.d, Nq = p.Nq, Nk = p.Nk;

    const half_t* __restrict__ Qg = reinterpret_cast<const half_t*>(p.q) + (size_t)b * p.q_bstride + (size_t)h * d;
    const half_t* __restrict__ Kg = reinterpret_cast<const half_t*>(p.k) + (size_t)b * p.k_bstride + (size_t)h * d;
    const half_t* __restrict__ Vg = reinterpret_cast<const half_t*>(p.vt) + (size_t)(b * p.H + h) * d * p.ldvt;

    // Q fragments (B operand: row = query, k = head-dim chunk 2*ks + hi)
    half8_t qf[QT][NKS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int q = q0 + qt * 32 + ql;
        if (q >= Nq) q = Nq - 1;
        const half_t* qrow = Qg + (size_t)q * p.ldq;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int c0 = (2 * ks + hi) * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (c0 < d) v = ld16(qrow + c0);
            qf[qt][ks] = *reinterpret_cast<half8_t*>(&v);
        }
    }

    f32x16 o[QT][NDT];
    float m_run[QT], l_run[QT];
    f32x16 negm[QT];                   // PRE: -m_run in all 16 accumulator slots = the C operand of the first QK^T MFMA
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = PRE ? 0.0f : -INFINITY;
        l_run[qt] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qt][r] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][dt][r] = 0.0f;
    }
    // q_prescaled input on the FMA path (small head dims, where the 16 extra registers of the C-init form would cost
    // occupancy): the scale is already in the logits
    const float c_scale = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    uint4 rk[K_PER_T], rv[V_PER_T];
    // tile-invariant staging coordinates (no per-tile divisions / 64-bit address rebuilds)
    const half_t* kptr[K_PER_T];
    const half_t* vptr[V_PER_T];
    int krow[K_PER_T], klds[K_PER_T], vlds[V_PER_T], vkey[V_PER_T];
    bool kok[K_PER_T], kone[K_PER_T], vok[V_PER_T], vone[V_PER_T];
#pragma unroll
    for (int i = 0; i < K_PER_T; ++i) {
        const int idx = tid + NTHR * i;
        const int row = idx / KCH;
        const int c = idx - row * KCH;
        krow[i] = row;
        kok[i] = (idx < K_ITEMS) && (c * 8 < d);
        kone[i] = (PRE == 2) && (idx < K_ITEMS) && (c * 8 == d);      // the pad chunk whose first column carries 1.0
        klds[i] = row * KSTR + c * 8;
        kptr[i] = Kg + (size_t)row * p.ldk + c * 8;
    }
#pragma unroll
    for (int i = 0; i < V_PER_T; ++i) {
        const int idx = tid + NTHR * i;
        const int row = idx >> 3;      // head-dim column
        const int c = idx & 7;         // 8-key chunk
        vkey[i] = c * 8;
        vok[i] = (idx < V_ITEMS) && (row < d);
        vone[i] = ONES && (idx < V_ITEMS) && (row == OC);
        // LDS image of a V^T row: per 16-key group [keys 0-3, 8-11 | keys 4-7, 12-15], i.e. the 8 keys a lane
        // half feeds to one P.V MFMA k-step are contiguous -> ONE 16-byte read per fragment
        vlds[i] = row * VSTR2 + 16 * (c >> 1) + 4 * (c & 1);
        vptr[i] = Vg + (size_t)row * p.ldvt + c * 8;
    }
    const size_t kstep = (size_t)KT * p.ldk;

    auto load_tile = [&](int key0) {
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (kone[i]) v.x = 0x00003C00u;                          // fp16 1.0 in column d
            if (kok[i] && key0 + krow[i] < Nk) v = ld16(kptr[i]);
            kptr[i] += kstep;
            rk[i] = v;
        }
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (vone[i]) v = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);   // fp16 1.0 x 8
            if (vok[i]) {
                v = ld16(vptr[i]);
                // keys >= Nk of the last tile: P is 0 there, but 0 * (stale NaN / Inf) would poison O, so the V^T pad
                // is masked here and its contents never matter (the producer does not have to zero-fill it)
                const int kfirst = key0 + vkey[i];
                if (kfirst + 8 > Nk) {
                    const int keep = Nk - kfirst;            // valid keys in this 8-key chunk (<= 7, possibly <= 0)
                    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (2 * q >= keep) w[q] = 0u;
                        else if (2 * q + 1 >= keep) w[q] &= 0xFFFFu;
                    }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            vptr[i] += KT;
            rv[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        half_t* Ksw = Ksm + buf * KBUF;
        half_t* Vsw = Vsm + buf * VBUF;
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i)
            if (tid + NTHR * i < K_ITEMS) st16(Ksw + klds[i], rk[i]);
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i)
            if (tid + NTHR * i < V_ITEMS) {
                uint2* dst = reinterpret_cast<uint2*>(Vsw + vlds[i]);
                dst[0] = make_uint2(rv[i].x, rv[i].y);      // keys 8c .. 8c+3  -> lane half 0
                dst[2] = make_uint2(rv[i].z, rv[i].w);      // keys 8c+4 .. 8c+7 -> lane half 1 (+16 bytes)
            }
    };

    const int ntiles = (Nk + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KT;
        const half_t* Ks = Ksm + (t & 1) * KBUF;
        const half_t* Vs = Vsm + (t & 1) * VBUF;
        if (t + 1 < ntiles) load_tile(key0 + KT);      // global -> registers, in flight during the MFMAs below
        const bool tail = (key0 + KT > Nk);

        // ---- S^T = K . Q^T : two 32-key halves, K fragments shared by the QT query sub-tiles
        f32x16 s[QT][2];
        const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (kh * 32 + ql) * KSTR + (2 * ks + hi) * 8);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[qt][kh] = mfma32(kf, qf[qt][ks], ks == 0 ? (PRE == 1 ? negm[qt] : zero16) : s[qt][kh]);
            }
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        // ---- online softmax (this lane: query ql of each sub-tile, keys kh*32 + (r&3) + 8*(r>>2) + 4*hi).
        // VALU-lean form (the first version spent 32 VALU instructions per MFMA): the row max is taken on the
        // raw scores, the softmax scale rides in one FMA with the exp2 argument, pairs are converted with
        // v_cvt_pk_f16_f32, and the O / l rescale is DEFERRED: the running max is only raised (and the 16*NDT
        // accumulator registers rescaled through AGPR read/mul/write) when some row's tile max exceeds it by
        // more than 2^DEFER.  Until then P <= 2^DEFER, exactly representable-range for fp16 with the same
        // relative precision, and l / O accumulate in fp32, so results are unchanged to rounding.
        constexpr float DEFER = 8.0f;
        uint4 pf[QT][4];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (tail) {
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (key >= Nk) s[qt][kh][r] = -INFINITY;
                    }
            }
            auto sv = [&](int i) -> float { return s[qt][i >> 4][i & 15]; };
            float tmax = max3f(sv(0), sv(1), sv(2));
#pragma unroll
            for (int i = 3; i < 31; i += 2) tmax = max3f(tmax, sv(i), sv(i + 1));
            tmax = fmaxf(tmax, sv(31));
            float psum = 0.0f;
            if constexpr (PRE != 0) {
                // scores are s' = logit - m_run (exp2 units).  Rescale when some row's tile max exceeds the running max
                // by more than 2^DEFER -- and always on the first tile, which establishes the true row max (m_run = 0
                // until then, so a row of very negative logits cannot underflow its whole first tile)
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                float inc = 0.0f;
                if (t == 0 || __any(tmax > DEFER)) {
                    inc = (t == 0) ? tmax : fmaxf(tmax, 0.0f);
                    if constexpr (PRE == 2) inc = (float)(half_t)(m_run[qt] + inc) - m_run[qt];   // m_run stays fp16-representable
                    const float alpha = (t == 0) ? 1.0f : __builtin_amdgcn_exp2f(-inc);    // O and l are still 0 on the first tile
                    m_run[qt] += inc;
                    if constexpr (!ONES) l_run[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
                    if constexpr (PRE == 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) negm[qt][r] = -m_run[qt];
                    } else {
                        // pad column d = DQK - 8 of this query's Q row: last k-step, upper lane half, first element
                        if (hi == 1) qf[qt][NKS - 1][0] = (half_t)(-m_run[qt]);
                    }
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[qt][kh][r] -= inc;
                }
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float p0 = __builtin_amdgcn_exp2f(s[qt][kh][r]);
                        const float p1 = __builtin_amdgcn_exp2f(s[qt][kh][r + 1]);
                        if constexpr (!ONES) psum += p0 + p1;
                        f32x2 pv = {p0, p1};
                        const half2_t ph = __builtin_convertvector(pv, half2_t);
                        const unsigned pw = *reinterpret_cast<const unsigned*>(&ph);
                        const int j = kh * 2 + (r >> 3);
                        const int e = (r & 7) >> 1;
                        if (e == 0) pf[qt][j].x = pw; else if (e == 1) pf[qt][j].y = pw; else if (e == 2) pf[qt][j].z = pw; else pf[qt][j].w = pw;
                    }
            } else {
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * c_scale;
            if (__any(tmax > m_run[qt] + DEFER)) {
                const float m_new = fmaxf(m_run[qt], tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
                m_run[qt] = m_new;
                if constexpr (!ONES) l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
            }
            const float nm = -m_run[qt];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(s[qt][kh][r], c_scale, nm));
                    const float p1 = __builtin_amdgcn_exp2f(fmaf(s[qt][kh][r + 1], c_scale, nm));
                    if constexpr (!ONES) psum += p0 + p1;
                    f32x2 pv = {p0, p1};
                    const half2_t ph = __builtin_convertvector(pv, half2_t);
                    const unsigned pw = *reinterpret_cast<const unsigned*>(&ph);
                    const int j = kh * 2 + (r >> 3);
                    const int e = (r & 7) >> 1;
                    if (e == 0) pf[qt][j].x = pw; else if (e == 1) pf[qt][j].y = pw; else if (e == 2) pf[qt][j].z = pw; else pf[qt][j].w = pw;
                }
            }
            if constexpr (!ONES) l_run[qt] += psum;
        }
        // ---- O^T += V^T . P^T : k-step j covers keys 16j..16j+15 in the order (e&3) + 8*(e>>2) + 4*hi
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const half8_t vf = *reinterpret_cast<const half8_t*>(Vs + (dt * 32 + ql) * VSTR2 + 16 * j + 8 * hi);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma32(vf, *reinterpret_cast<const half8_t*>(&pf[qt][j]), o[qt][dt]);
            }
        }
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        if (t + 1 < ntiles) store_tile((t + 1) & 1);   // other buffer: last read one iteration ago, before the barrier
        __syncthreads();
    }

    // ---- finalize: O /= l, write fp16.  lane holds head-dim columns dt*32 + 8*rg + 4*hi + {0..3}
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float lpart = l_run[qt];
        if constexpr (ONES) lpart = (hi == 1) ? o[qt][NDT - 1][15] : 0.0f;   // column OC = 31 of the last tile lives in (r = 15, hi = 1)
        const float l_tot = lpart + __shfl_xor(lpart, 32, 64);
        const float inv = 1.0f / l_tot;
        const int q = q0 + qt * 32 + ql;
        if (q < Nq) {
            half_t* orow = reinterpret_cast<half_t*>(p.out) + (size_t)b * p.o_bstride + (size_t)q * p.ldo + (size_t)h * d;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int c = dt * 32 + 8 * rg + 4 * hi;
                    if (c < d) {
                        half4_t ov;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) ov[jj] = (half_t)(o[qt][dt][rg * 4 + jj] * inv);
                        *reinterpret_cast<half4_t*>(orow + c) = ov;
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Split-fp16 attention (strict mode, gl_attn_args.q_lo != NULL): every matrix-core operand is hi + lo
// (hi = fp16(x), lo = fp16(x - hi): ~22 mantissa bits), every product takes three MFMA passes
//   S = Khi.Qhi + Khi.Qlo + Klo.Qhi,   P = Phi + Plo (split in registers),   O += Vhi.Phi + Vhi.Plo + Vlo.Phi,
// softmax statistics and the row sum in fp32 from the unsplit probabilities; out / out_lo receive fp16(O), fp16(O - fp16(O)).
// Same work decomposition and register layouts as attn_kernel (S^T = K.Q^T so a lane owns its query's scores, P fed from
// the score registers); 4 waves x 32 queries, ONE LDS buffer set (K hi / lo, V^T hi / lo: 89 KB at d = 160, dynamic LDS)
// with two barriers per 64-key tile, the FMA form of the online softmax.  Three times the matrix work of attn_kernel by
// construction: this is the parity mode, not the fast one.
template <int DQK, int NW, bool DBUF>
__global__ __launch_bounds__(64 * NW) void attn_split_kernel(gl_attn_args p) {
    // NW = 4 or 8 waves (8: 256 queries share every staged tile).  DBUF: the next tile's rows are loaded into registers before the
    // MFMAs of the current one and stored to the OTHER LDS buffer set after them (one barrier per tile, loads in flight under the
    // compute, as attn_kernel); without it one buffer set and two barriers per tile (the large head dims: 89 KB per set at d = 160).
    constexpr int NTHR = 64 * NW;
    constexpr int NKS = DQK / 16, NDT = (DQK + 31) / 32, KSTR = DQK + 8, KCH = DQK / 8;
    constexpr int K_ITEMS = KT * KCH, K_PER_T = (K_ITEMS + NTHR - 1) / NTHR;
    constexpr int V_ITEMS = NDT * 32 * (KT / 8), V_PER_T = (V_ITEMS + NTHR - 1) / NTHR;
    constexpr int QBLK = NW * 32;
    constexpr int KBUF = KT * KSTR, VBUF = NDT * 32 * VSTR2;
    constexpr int SET = 2 * KBUF + 2 * VBUF;                    // one buffer set: K hi, K lo, V^T hi, V^T lo
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
    half_t* sm = reinterpret_cast<half_t*>(smem_attn);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 31, hi = lane >> 5;
    const int nqb = (p.Nq + QBLK - 1) / QBLK;
    int logical;
    {
        const int total = gridDim.x, L = blockIdx.x;
        const int xcd = L & 7, qd = total >> 3, rm = total & 7;
        logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (L >> 3);
    }
    const int qb = logical % nqb, bh = logical / nqb, h = bh % p.H, b = bh / p.H;
    const int q0 = qb * QBLK + wave * 32;
    const int d = p.d, Nq = p.Nq, Nk = p.Nk;
    const size_t qoff = (size_t)b * p.q_bstride + (size_t)h * d, koff = (size_t)b * p.k_bstride + (size_t)h * d;
    const size_t voff = (size_t)(b * p.H + h) * d * p.ldvt;
    const half_t* Kg[2] = {reinterpret_cast<const half_t*>(p.k) + koff, reinterpret_cast<const half_t*>(p.k_lo) + koff};
    const half_t* Vg[2] = {reinterpret_cast<const half_t*>(p.vt) + voff, reinterpret_cast<const half_t*>(p.vt_lo) + voff};

    half8_t qf[2][NKS];                     // [hi, lo] Q fragments
    {
        int q = q0 + ql;
        if (q >= Nq) q = Nq - 1;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const half_t* qrow = reinterpret_cast<const half_t*>(part ? p.q_lo : p.q) + qoff + (size_t)q * p.ldq;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int c0 = (2 * ks + hi) * 8;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (c0 < d) v = ld16(qrow + c0);
                qf[part][ks] = *reinterpret_cast<half8_t*>(&v);
            }
        }
    }
    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float c_scale = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    // ---- staging: rows of K hi / lo [64][d] and V^T hi / lo [d][64] (zero padding, keys >= Nk masked: see attn_kernel)
    uint4 rk[2][K_PER_T], rv[2][V_PER_T];
    auto load_tile = [&](const int key0) __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
            for (int i = 0; i < K_PER_T; ++i) {
                const int idx = tid + NTHR * i;
                const int row = idx / KCH, c = idx - row * KCH;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (idx < K_ITEMS && c * 8 < d && key0 + row < Nk) v = ld16(Kg[part] + (size_t)(key0 + row) * p.ldk + c * 8);
                rk[part][i] = v;
            }
#pragma unroll
            for (int i = 0; i < V_PER_T; ++i) {
                const int idx = tid + NTHR * i;
                const int row = idx >> 3, c = idx & 7;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (idx < V_ITEMS && row < d) {
                    v = ld16(Vg[part] + (size_t)row * p.ldvt + key0 + c * 8);
                    const int kfirst = key0 + c * 8;
                    if (kfirst + 8 > Nk) {
                        const int keep = Nk - kfirst;
                        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (2 * q >= keep) w[q] = 0u;
                            else if (2 * q + 1 >= keep) w[q] &= 0xFFFFu;
                        }
                        v = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                }
                rv[part][i] = v;
            }
        }
    };
    auto store_tile = [&](const int buf) __attribute__((always_inline)) {
        half_t* base = sm + buf * SET;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
            for (int i = 0; i < K_PER_T; ++i) {
                const int idx = tid + NTHR * i;
                const int row = idx / KCH, c = idx - row * KCH;
                if (idx < K_ITEMS) st16(base + part * KBUF + row * KSTR + c * 8, rk[part][i]);
            }
#pragma unroll
            for (int i = 0; i < V_PER_T; ++i) {
                const int idx = tid + NTHR * i;
                const int row = idx >> 3, c = idx & 7;
                if (idx < V_ITEMS) {
                    uint2* dst = reinterpret_cast<uint2*>(base + 2 * KBUF + part * VBUF + row * VSTR2 + 16 * (c >> 1) + 4 * (c & 1));
                    dst[0] = make_uint2(rv[part][i].x, rv[part][i].y);
                    dst[2] = make_uint2(rv[part][i].z, rv[part][i].w);
                }
            }
        }
    };

    const int ntiles = (Nk + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KT;
        const int cur = DBUF ? (t & 1) : 0;
        const half_t* Ksm = sm + cur * SET;
        const half_t* Vsm = Ksm + 2 * KBUF;
        if (DBUF && t + 1 < ntiles) load_tile(key0 + KT);          // global -> registers, in flight under the MFMAs below
        const bool tail = (key0 + KT > Nk);
        // ---- S^T = K . Q^T in three passes (small terms first)
        f32x16 s[2];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kh][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const half8_t kfh = *reinterpret_cast<const half8_t*>(Ksm + (kh * 32 + ql) * KSTR + (2 * ks + hi) * 8);
                const half8_t kfl = *reinterpret_cast<const half8_t*>(Ksm + KBUF + (kh * 32 + ql) * KSTR + (2 * ks + hi) * 8);
                s[kh] = mfma32(kfl, qf[0][ks], s[kh]);
                s[kh] = mfma32(kfh, qf[1][ks], s[kh]);
                s[kh] = mfma32(kfh, qf[0][ks], s[kh]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (tail) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= Nk) s[kh][r] = -INFINITY;
                }
        }
        // ---- online softmax, fp32, exact running max (no deferral: P <= 1)
        float tmax = s[0][0];
#pragma unroll
        for (int i = 1; i < 32; ++i) tmax = fmaxf(tmax, s[i >> 4][i & 15]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * c_scale;
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // first tile: exp2(-inf) = 0 on zero accumulators
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        uint4 pfh[4], pfl[4];
        float psum = 0.0f;
        // probabilities are kept scaled by 2^13 (P <= 8192; O and the row sum carry the same factor, which cancels in O / l): every P below
        // fp16's smallest normal (6.1e-5) loses its mantissa bits -- and its lo half -- to the subnormal quantum; at 1024+ keys that tail
        // carried 1e-4 of the result (tiny UNet at 32x32 latents: 9.9e-5 unscaled, 6.2e-6 scaled by 2^8)
        const float nm = 13.0f - m_run;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = pin_value(__builtin_amdgcn_exp2f(fmaf(s[kh][r], c_scale, nm)));       // hi and lo from ONE value (common.h)
                const float p1 = pin_value(__builtin_amdgcn_exp2f(fmaf(s[kh][r + 1], c_scale, nm)));
                psum += p0 + p1;
                f32x2 pv = {p0, p1};
                const half2_t ph = __builtin_convertvector(pv, half2_t);
                f32x2 pr = {p0 - (float)ph[0], p1 - (float)ph[1]};
                const half2_t pl = __builtin_convertvector(pr, half2_t);
                const unsigned pw = *reinterpret_cast<const unsigned*>(&ph), lw = *reinterpret_cast<const unsigned*>(&pl);
                const int j = kh * 2 + (r >> 3);
                const int e = (r & 7) >> 1;
                if (e == 0) { pfh[j].x = pw; pfl[j].x = lw; } else if (e == 1) { pfh[j].y = pw; pfl[j].y = lw; }
                else if (e == 2) { pfh[j].z = pw; pfl[j].z = lw; } else { pfh[j].w = pw; pfl[j].w = lw; }
            }
        l_run += psum;
        // ---- O^T += V^T . P^T in three passes
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const half8_t vfh = *reinterpret_cast<const half8_t*>(Vsm + (dt * 32 + ql) * VSTR2 + 16 * j + 8 * hi);
                const half8_t vfl = *reinterpret_cast<const half8_t*>(Vsm + VBUF + (dt * 32 + ql) * VSTR2 + 16 * j + 8 * hi);
                o[dt] = mfma32(vfl, *reinterpret_cast<const half8_t*>(&pfh[j]), o[dt]);
                o[dt] = mfma32(vfh, *reinterpret_cast<const half8_t*>(&pfl[j]), o[dt]);
                o[dt] = mfma32(vfh, *reinterpret_cast<const half8_t*>(&pfh[j]), o[dt]);
            }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (DBUF) {
            if (t + 1 < ntiles) store_tile((t + 1) & 1);       // the other set: last read one iteration ago, before the barrier
            __syncthreads();
        } else {
            __syncthreads();       // every wave is done with the tile before the next one is staged
            if (t + 1 < ntiles) {
                load_tile(key0 + KT);
                store_tile(0);
                __syncthreads();
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + ql;
    if (q < Nq) {
        const size_t orow = (size_t)b * p.o_bstride + (size_t)q * p.ldo + (size_t)h * d;
        half_t* oh = reinterpret_cast<half_t*>(p.out) + orow;
        half_t* ol = p.out_lo ? reinterpret_cast<half_t*>(p.out_lo) + orow : nullptr;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = dt * 32 + 8 * rg + 4 * hi;
                if (c < d) {
                    half4_t ov, lv;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const float v = pin_value(o[dt][rg * 4 + jj] * inv);
                        ov[jj] = (half_t)v;
                        lv[jj] = (half_t)(v - (float)ov[jj]);
                    }
                    *reinterpret_cast<half4_t*>(oh + c) = ov;
                    if (ol) *reinterpret_cast<half4_t*>(ol + c) = lv;
                }
            }
    }
}

// 16 zero bytes in global memory: what a masked lane of the LDS-DMA staging reads
__device__ uint4 g_attn_zero16[4];

__device__ __forceinline__ void glds16_attn(const char* src, char* dst) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(src)),
        reinterpret_cast<__attribute__((address_space(3))) void*>(reinterpret_cast<uintptr_t>(dst)), 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// attn_split_pipe_kernel<D8> (round 6): the split-fp16 attention of the level-0 head dims (d = 8 * D8 in {32, 40, 48}), restructured.
// attn_split_kernel runs [Q.K^T MFMAs] [softmax VALU] [P.V MFMAs] back to back in every wave, and the two waves of a SIMD, locked to one
// barrier per tile, do the same thing at the same time: measured 757 us at (40, 4096, 4096) = exactly the SUM of its MFMA time (42 x 32
// cycles) and its VALU time per wave-tile, i.e. the matrix pipe and the VALU never overlap.  Here:
//   * software pipeline inside a wave: iteration t issues the MFMAs of P.V(t-1) and Q.K^T(t+1) interleaved with the softmax VALU of tile
//     t (scores of tile t were produced one iteration earlier), so every MFMA has independent VALU work behind it in program order;
//   * fewer MFMAs: Q.K^T as ONE product over the concatenated K dimension [khi | khi | klo] x [qhi | qlo | qhi] (3 d = 120 -> 8 steps of 16
//     instead of 3 x 48 -> 9); P.V on the STACKED rows [vhi (d) | vlo (d)] of V^T: Phi multiplies all ceil(2 d / 32) row tiles, Plo only
//     the tiles that hold vhi rows (it also meets the vlo rows that share such a tile: the 4th term Plo.Vlo of the exact product, 2^-22) --
//     5 instead of 6 MFMAs per 16 keys at d = 40; the output is acc[row] + acc[row + d], which is the same lane, 4 D8 registers apart;
//   * deferred rescale: the running max moves only when a tile's max exceeds it by more than 2 (P <= 2^15 in the 2^13-scaled fp16
//     fragments), as a wave-uniform branch; packed fp32 VALU (v_pk_fma / v_pk_add / v_pk_mul) for the exponent, row sum and rescale.
// 8 waves x 32 queries; K tiles [64][khi | klo] and stacked V^T tiles in two LDS slots each (K(t+1), V(t-1) are read while K(t+2), V(t)
// are written; one barrier per tile); results agree with attn_split_kernel to fp32 summation order.
template <int D8>
__global__ __launch_bounds__(512) void attn_split_pipe_kernel(gl_attn_args p) {
    constexpr int D = 8 * D8;
    constexpr int NKS = D8 + (D8 + 1) / 2;          // 16-wide steps over the concatenated Q.K^T K dimension (3 d columns, see qcat)
    constexpr int NT = (2 * D + 31) / 32;           // 32-row tiles of the stacked V^T
    constexpr int NTL = (D + 31) / 32;              // ... that hold vhi rows
    constexpr int KSTR = 2 * D + 8;                 // K row [khi | klo] + 16 bytes: (KSTR / 2) / 4 is odd -> conflict-free 16-byte fragment reads
    constexpr int VROWS = NT * 32;
    constexpr int VCH = VSTR2 / 8;                  // 16-byte chunks per staged V^T row (8 of keys + 1 of padding)
    constexpr int KCH = KSTR / 8;                   // ... per staged K row (2 D8 of [khi | klo] + 1 of padding)
    constexpr int NV_I = (2 * D * VCH + 63) / 64;   // LDS-DMA wave instructions (64 x 16 bytes) per V^T tile; a K tile takes exactly KCH
    constexpr int KBUF = KT * KSTR, VBUF = (VROWS * VCH > NV_I * 64 ? VROWS * VCH : NV_I * 64) * 8;
    constexpr int NTHR = 512;
    static_assert(D8 >= 4 && D8 <= 6 && ((KSTR / 2) / 4) % 2 == 1, "head dims 32 / 40 / 48");
    constexpr int NS = ATTN_PIPE_SLOTS;             // tile k of either operand lives in slot k % NS
    static_assert(NS == 2 || NS == 4, "slots");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pipe[];
    half_t (*sK)[KBUF] = reinterpret_cast<half_t (*)[KBUF]>(smem_pipe);
    half_t (*sV)[VBUF] = reinterpret_cast<half_t (*)[VBUF]>(smem_pipe + (size_t)NS * KBUF * sizeof(half_t));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 31, hi = lane >> 5;
    const int nqb = (p.Nq + 255) / 256;
    int logical;
    {
        const int total = gridDim.x, L = blockIdx.x;
        const int xcd = L & 7, qd = total >> 3, rm = total & 7;
        logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (L >> 3);
    }
    const int qb = logical % nqb, bh = logical / nqb, h = bh % p.H, b = bh / p.H;
    const int q0 = qb * 256 + wave * 32;
    const int Nq = p.Nq, Nk = p.Nk;
    const size_t qoff = (size_t)b * p.q_bstride + (size_t)h * D, koff = (size_t)b * p.k_bstride + (size_t)h * D;
    const size_t voff = (size_t)(b * p.H + h) * D * p.ldvt;
    const half_t* Kg[2] = {reinterpret_cast<const half_t*>(p.k) + koff, reinterpret_cast<const half_t*>(p.k_lo) + koff};
    const half_t* Vg[2] = {reinterpret_cast<const half_t*>(p.vt) + voff, reinterpret_cast<const half_t*>(p.vt_lo) + voff};

    // rows [2 D, VROWS) of both V slots are never staged: zero them once (they meet P in the MFMAs: 0, not stale LDS bits)
    if constexpr (VROWS > 2 * D) {
        constexpr int ZW = (VROWS - 2 * D) * VSTR2 / 8;
        for (int i = tid; i < NS * ZW; i += NTHR) st16(&sV[i / ZW][2 * D * VSTR2 + 8 * (i % ZW)], make_uint4(0u, 0u, 0u, 0u));
    }

    // The concatenated K dimension of Q.K^T, 16 columns (two 8-column chunks: lanes hi = 0 / 1) per MFMA step, paired so that the K-side
    // fragment address is ONE per-lane base + a compile-time offset:
    //   steps 0 .. D8-1        : hi = 0 -> khi chunk i x qhi chunk i,  hi = 1 -> klo chunk i x qhi chunk i   (K base A = row + hi * D)
    //   steps D8 .. NKS-1 (m)  : khi chunk 2m + hi x qlo chunk 2m + hi (zero Q chunk past D8)                  (K base B = row + hi * 8)
    half8_t qcat[NKS];
    {
        int q = q0 + ql;
        if (q >= Nq) q = Nq - 1;
        const half_t* qh = reinterpret_cast<const half_t*>(p.q) + qoff + (size_t)q * p.ldq;
        const half_t* qlo = reinterpret_cast<const half_t*>(p.q_lo) + qoff + (size_t)q * p.ldq;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (ks < D8) v = ld16(qh + ks * 8);
            else if (2 * (ks - D8) + hi < D8) v = ld16(qlo + (2 * (ks - D8) + hi) * 8);
            qcat[ks] = *reinterpret_cast<half8_t*>(&v);
        }
    }
    const int kbase_a = ql * KSTR + hi * D, kbase_b = ql * KSTR + hi * 8;
    // halves added to kbase_a / kbase_b for step ks (the zero Q chunk of an odd D8 meets khi chunk D8 - 1 again: finite)
    auto kfo = [&](const int ks) __attribute__((always_inline)) -> int {
        if (ks < D8) return kbase_a + ks * 8;
        const int m = ks - D8;
        return (2 * m + 1 < D8) ? kbase_b + 16 * m : ql * KSTR + 16 * m;        // last step of an odd D8: both halves read chunk D8 - 1
    };

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float c_scale = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    // ---- staging (global -> registers -> LDS): K rows [khi | klo], stacked V^T rows with the key order of the P fragments (attn_kernel)
    // Staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land at consecutive LDS addresses, so the LDS image is produced by choosing
    // each lane's SOURCE address): no staging registers, no ds_write, nothing to wait for before the stores -- the loop runs at the 256-register
    // limit and the 15 registers this frees pay for a deeper K-fragment ring.  A K tile is 64 rows x KCH chunks = KCH wave instructions, a stacked
    // V^T tile 2 D rows x VCH chunks = NV_I; the KCH + NV_I instructions of a tile are dealt round-robin to the 8 waves (the KIND of a wave's
    // instruction i is wave-uniform).  Per lane and instruction one loop-invariant byte offset from the kind's hi array at key0 = 0 (the lo arrays
    // are reached through their distance from the hi ones: the launcher checks that it fits); ZPAGE = the lane reads 16 zero bytes instead (pad
    // rows of the V^T image; K rows at or past Nk; pad chunks read a valid neighbour).
    constexpr int NI = KCH + NV_I, DPW = (NI + 7) / 8;
    constexpr unsigned ZPAGE = 0xFFFFFFFFu;
    const char* zpage = reinterpret_cast<const char*>(g_attn_zero16);
    // (offsets are taken from the LOWER of the hi / lo arrays, so they are non-negative whichever way round the caller allocated them)
    const char* kmin = reinterpret_cast<const char*>(Kg[0] < Kg[1] ? Kg[0] : Kg[1]);
    const char* vmin = reinterpret_cast<const char*>(Vg[0] < Vg[1] ? Vg[0] : Vg[1]);
    const unsigned kd[2] = {(unsigned)(reinterpret_cast<const char*>(Kg[0]) - kmin), (unsigned)(reinterpret_cast<const char*>(Kg[1]) - kmin)};
    const unsigned vd[2] = {(unsigned)(reinterpret_cast<const char*>(Vg[0]) - vmin), (unsigned)(reinterpret_cast<const char*>(Vg[1]) - vmin)};
    unsigned doff[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int I = wave + 8 * i;                     // wave-uniform
        if (I < KCH) {
            const int pch = 64 * I + lane, row = pch / KCH, c = pch - row * KCH;
            doff[i] = (unsigned)(row * p.ldk * 2) + (c < D8 ? kd[0] + (unsigned)(c * 16) : (c < 2 * D8 ? kd[1] + (unsigned)((c - D8) * 16) : kd[0]));
        } else {
            const int pch = 64 * (I - KCH) + lane, row = pch / VCH, c = pch - row * VCH;
            doff[i] = row >= 2 * D ? ZPAGE : (unsigned)((row >= D ? row - D : row) * p.ldvt * 2) + (row >= D ? vd[1] : vd[0]) + (c < 8 ? (unsigned)(c * 16) : 0u);
        }
    }
    auto stage_dma = [&](const int key0_k, const bool do_k, const int kslot, const int key0_v, const bool do_v, const int vslot) __attribute__((always_inline)) {
        const bool full_k = key0_k + KT <= Nk;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int I = __builtin_amdgcn_readfirstlane(wave) + 8 * i;
            if (I < KCH) {
                if (do_k) {
                    const char* src = kmin + (size_t)key0_k * p.ldk * 2 + doff[i];
                    if (!full_k && key0_k + (64 * I + lane) / KCH >= Nk) src = zpage;
                    glds16_attn(src, reinterpret_cast<char*>(&sK[kslot][0]) + I * 1024);
                }
            } else if (I < NI) {
                if (do_v) {
                    const char* src = doff[i] == ZPAGE ? zpage : vmin + (size_t)key0_v * 2 + doff[i];
                    glds16_attn(src, reinterpret_cast<char*>(&sV[vslot][0]) + (I - KCH) * 1024);
                }
            }
        }
    };
    // the LAST key tile's V^T (key columns past Nk may hold anything, NaN included, and a 16-byte chunk may straddle Nk) goes through registers with
    // the per-key masking of attn_kernel, into the same contiguous-key image; once per block, synchronous
    auto stage_v_masked = [&](const int key0_v, const int vslot) __attribute__((always_inline)) {
        for (int idx = tid; idx < 2 * D * 8; idx += NTHR) {
            const int row = idx >> 3, c = idx & 7;
            const half_t* src = (row >= D ? Vg[1] + (size_t)(row - D) * p.ldvt : Vg[0] + (size_t)row * p.ldvt) + key0_v + c * 8;
            uint4 v = ld16(src);
            const int kfirst = key0_v + c * 8;
            if (kfirst + 8 > Nk) {
                const int keep = Nk - kfirst;
                unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (2 * q >= keep) w[q] = 0u;
                    else if (2 * q + 1 >= keep) w[q] &= 0xFFFFu;
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            st16(&sV[vslot][row * VSTR2 + c * 8], v);
        }
    };
    // fragment loads (LDS -> registers); every consumer below prefetches one group ahead of its MFMAs
    auto load_vf = [&](const half_t* Vsm, const int j, half8_t (&vf)[NT]) __attribute__((always_inline)) {
        // the P fragment of step j holds, per lane half hi, keys 16 j + 4 hi + [0, 4) and 16 j + 8 + 4 hi + [0, 4) (attn_kernel): two 8-byte reads
        // of the contiguous-key row (the LDS-DMA image cannot carry attn_kernel's permuted key order)
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const half_t* rp = Vsm + (i * 32 + ql) * VSTR2 + 16 * j + 4 * hi;
            const uint2 a = *reinterpret_cast<const uint2*>(rp), b2 = *reinterpret_cast<const uint2*>(rp + 8);
            const uint4 v = make_uint4(a.x, a.y, b2.x, b2.y);
            vf[i] = *reinterpret_cast<const half8_t*>(&v);
        }
    };
    constexpr int KG = 2;                               // K fragments per prefetch group
    constexpr int KRING = ATTN_PIPE_KRING;              // groups resident in registers (KRING - 1 ahead of the MFMAs)
    constexpr int NKG = (NKS + KG - 1) / KG;            // groups per 32-key half
    auto load_kf = [&](const half_t* Ksm, const int kh, const int g, half8_t (&kf)[KG]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KG; ++i)
            if (g * KG + i < NKS) kf[i] = *reinterpret_cast<const half8_t*>(Ksm + kh * 32 * KSTR + kfo(g * KG + i));
    };
    // O^T += stacked V^T . P^T for the 16 keys of fragment j: Phi meets every row tile, Plo the tiles that hold vhi rows
    auto pv_mfma = [&](const half8_t (&vf)[NT], const uint4 (&fh)[4], const uint4 (&fl)[4], const int j) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = mfma32(vf[i], *reinterpret_cast<const half8_t*>(&fh[j]), acc[i]);
#pragma unroll
        for (int i = 0; i < NTL; ++i) acc[i] = mfma32(vf[i], *reinterpret_cast<const half8_t*>(&fl[j]), acc[i]);
    };
    // probabilities of fragment g (8 scores of this lane's query) of the tile whose scores sit in s: p = 2^13 exp2(s c - m), split hi + lo
    f32x2 ps2 = {0.0f, 0.0f};
    float nm = 0.0f;
    auto prob_group = [&](const f32x16 (&s)[2], const int g, uint4& fh, uint4& fl) __attribute__((always_inline)) {
        const int kh = g >> 1, r0 = (g & 1) * 8;
        unsigned hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x2 sv = {s[kh][r0 + 2 * e], s[kh][r0 + 2 * e + 1]};
            f32x2 ar = sv * c_scale + nm;
            f32x2 pv;
            pv[0] = __builtin_amdgcn_exp2f(ar[0]);
            pv[1] = __builtin_amdgcn_exp2f(ar[1]);
            asm("" : "+v"(pv));                         // hi and lo from ONE value (common.h pin_value)
            ps2 += pv;
            const half2_t ph = __builtin_convertvector(pv, half2_t);
            hw[e] = *reinterpret_cast<const unsigned*>(&ph);
            // lo = fp16(p - fp32(hi)), both halves by the mixed-precision FMA (fp16 source read in place, fp32 arithmetic, fp16 result written
            // into its half of the destination): 2 instructions per pair instead of 2 x v_cvt_f32_f16 + v_pk_add_f32 + v_cvt_pk_f16_f32.
            // p - hi is exact in fp32, so the result is the same bits.
            unsigned lo2;                               // (mixlo leaves the upper half of its destination alone; mixhi writes it next)
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo2) : "v"(hw[e]), "v"(pv[0]));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo2) : "v"(hw[e]), "v"(pv[1]));
            lw[e] = lo2;
            // the fragments are only CONSUMED one iteration later: without a use here the compiler sinks the conversions past the MFMAs,
            // to the end of the iteration, where nothing covers them
            asm volatile("" : "+v"(hw[e]), "+v"(lw[e]));
        }
        fh = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        fl = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    };

    const int ntiles = (Nk + KT - 1) / KT;
    // Schedule (NS = 4): iteration t READS K(t+1) and V(t-1) and STAGES K(t+3) and V(t+1) -- first read in iteration t+2 -- into the slots of K(t-1) / V(t-3),
    // last read in iteration t-2; a barrier (with vmcnt(0)) at the end of every ODD iteration and of the last one lies between every such write and its
    // readers (RAW) and between every slot's last readers and its next writer (WAR).  NS = 2: K(t+2) / V(t) staged in iteration t, a barrier every iteration.
    constexpr int AHEAD = NS == 4 ? 1 : 0;
    // prologue: K(0) .. K(1 + AHEAD) (and V(0) when it is staged a tile ahead); scores of tile 0
    stage_dma(0, true, 0, 0, AHEAD == 1 && KT <= Nk, 0);
    if (ntiles > 1) stage_dma(KT, true, 1 % NS, 0, false, 0);
    if (AHEAD == 1 && ntiles > 2) stage_dma(2 * KT, true, 2, 0, false, 0);
    if (AHEAD == 1 && KT > Nk) stage_v_masked(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 s[2];                            // scores of the current tile; overwritten IN PLACE by the next tile's, half by half, as its probabilities are done
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int g = 0; g < NKG; ++g) {
            half8_t kf[KG];
            load_kf(sK[0], kh, g, kf);
#pragma unroll
            for (int i = 0; i < KG; ++i)
                if (g * KG + i < NKS) s[kh] = mfma32(kf[i], qcat[g * KG + i], (g | i) == 0 ? zero16 : s[kh]);
        }
    }
    if constexpr (ATTN_PIPE_PRIO == 1) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
    if constexpr (NS == 4) __syncthreads();  // every wave is done with K(0) before iteration 1 restages its slot (no barrier ends iteration 0)
    uint4 pfh[4], pfl[4];                   // P fragments: of tile t-1 while its P.V MFMAs run, replaced IN PLACE by tile t's, fragment by fragment
#pragma unroll
    for (int j = 0; j < 4; ++j) { pfh[j] = make_uint4(0u, 0u, 0u, 0u); pfl[j] = make_uint4(0u, 0u, 0u, 0u); }

    // One iteration (tile t).  MFMA stream: P.V(t-1) [4 fragments] -> Q.K^T(t+1) keys 0-31 -> Q.K^T(t+1) keys 32-63; VALU stream beside it:
    // running max + probabilities of fragments 0, 1 (scores s[0]) | fragments 2, 3 (s[1]; s[0] is being overwritten) | row sum, rare rescale.
    // has_prev / has_next / tail are compile-time so that an iteration is ONE basic block apart from the staging loads and the rare rescale.
    auto body = [&](auto prev_c, auto next_c, auto tail_c, const int t) __attribute__((always_inline)) {
        constexpr bool has_prev = decltype(prev_c)::value, has_next = decltype(next_c)::value, tail = decltype(tail_c)::value;
        const int key0 = t * KT;
        const bool more_k = t + 2 + AHEAD < ntiles;
        const half_t* Ksm = sK[(t + 1) % NS];           // K(t+1)
        const half_t* Vsm = sV[(t + NS - 1) % NS];      // V(t-1)
        half8_t vf[NT];                                 // ONE buffer: the next fragment's loads are issued right behind the MFMAs that read it
        if constexpr (has_prev) load_vf(Vsm, 0, vf);
        // K(t+2) and V(t) straight into the slots last read one iteration ago (free since the barrier that ended it); landed before this iteration's barrier
        // the V^T tile staged here (t + AHEAD) goes by LDS-DMA unless it is the last, partial one (masked register path at the bottom)
        const int tv = t + AHEAD;
        const bool v_dma = tv < ntiles && (tv + 1) * KT <= Nk, v_masked = tv < ntiles && !v_dma;
        if constexpr (!(ATTN_PIPE_DBG & 4)) stage_dma(key0 + (2 + AHEAD) * KT, more_k, (t + 2 + AHEAD) % NS, tv * KT, v_dma, tv % NS);
        if constexpr (tail) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= Nk) s[kh][r] = -INFINITY;
                }
        }
        float tmax = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
#pragma unroll
        for (int i = 3; i + 1 < 32; i += 2) tmax = fmaxf(fmaxf(tmax, s[i >> 4][i & 15]), s[(i + 1) >> 4][(i + 1) & 15]);
        tmax = fmaxf(tmax, s[1][15]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * c_scale;
        // deferred rescale (wave-uniform): the running max follows only when some query's tile max exceeds it by more than 2 -- the
        // probabilities, kept scaled by 2^13 (attn_split_kernel), then stay below 2^15 in their fp16 fragments
        float alpha = 1.0f;
        const bool resc = __any(tmax > m_run + 2.0f) != 0;
        if (resc) {
            const float m_new = fmaxf(m_run, tmax);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // first tile: exp2(-inf) = 0 on zero accumulators
            m_run = m_new;
        }
        nm = 13.0f - m_run;
        ps2[0] = 0.0f; ps2[1] = 0.0f;
        __builtin_amdgcn_sched_barrier(0);
        // ---- segment A: P.V(t-1), fragments 0..3 (V fragments one step ahead) | probabilities of fragments 0, 1 of tile t
        // (a fragment of tile t replaces the fragment of tile t-1 IN PLACE right after the MFMAs that consumed it, in program order)
        constexpr int NG = 2 * NKG;                     // K fragment groups of the tile: keys 0-31 (NKG groups), then keys 32-63
        half8_t kf[KRING][KG];                          // ring: group g sits in kf[g % KRING], KRING - 1 groups in flight ahead of the MFMAs
        auto load_group = [&](const int g) __attribute__((always_inline)) { load_kf(Ksm, g / NKG, g % NKG, kf[g % KRING]); };
        if constexpr (has_prev) { pv_mfma(vf, pfh, pfl, 0); load_vf(Vsm, 1, vf); }
        prob_group(s, 0, pfh[0], pfl[0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (has_prev) { pv_mfma(vf, pfh, pfl, 1); load_vf(Vsm, 2, vf); }
        prob_group(s, 1, pfh[1], pfl[1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (has_prev) { pv_mfma(vf, pfh, pfl, 2); load_vf(Vsm, 3, vf); }
        if constexpr (has_next) load_group(0);
        prob_group(s, 2, pfh[2], pfl[2]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- segment B: last P.V fragment, then Q.K^T(t+1): keys 0-31 into s[0], keys 32-63 into s[1] | probabilities of fragment 3, row sum
        if constexpr (has_prev) pv_mfma(vf, pfh, pfl, 3);
        if constexpr (has_next) {
#pragma unroll
            for (int g = 1; g < KRING - 1; ++g)
                if (g < NG) load_group(g);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if constexpr (!(ATTN_PIPE_DBG & 2)) {
                    if (g + KRING - 1 < NG) load_group(g + KRING - 1);
                }
                const int kh = g / NKG, gi = g % NKG;
#pragma unroll
                for (int i = 0; i < KG; ++i)
                    if (gi * KG + i < NKS) s[kh] = mfma32(kf[(ATTN_PIPE_DBG & 2) ? 0 : g % KRING][i], qcat[gi * KG + i], (gi | i) == 0 ? zero16 : s[kh]);      // C = 0: inline constant
                if (g == 0) prob_group(s, 3, pfh[3], pfl[3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            prob_group(s, 3, pfh[3], pfl[3]);
        }
        l_run = l_run * alpha + (ps2[0] + ps2[1]);
        if (resc) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
        }
        // ---- stage K(t+2) and V(t) into the slots last read one iteration ago; one barrier per tile
        if constexpr (!(ATTN_PIPE_DBG & 1)) {
            if (v_masked) stage_v_masked(tv * KT, tv % NS);
            if (NS == 2 || (t & 1) || t + 1 >= ntiles) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    };
    {
        using T = std::true_type;
        using F = std::false_type;
        if (ntiles == 1) body(F{}, F{}, T{}, 0);
        else {
            body(F{}, T{}, F{}, 0);
            for (int t = 1; t + 1 < ntiles; ++t) body(T{}, T{}, F{}, t);
            body(T{}, F{}, T{}, ntiles - 1);
        }
    }
    // P.V of the last tile
    {
        const half_t* Vsm = sV[(ntiles - 1) % NS];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            half8_t vf[NT];
            load_vf(Vsm, j, vf);
            pv_mfma(vf, pfh, pfl, j);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + ql;
    if (q < Nq) {
        const size_t orow = (size_t)b * p.o_bstride + (size_t)q * p.ldo + (size_t)h * D;
        half_t* oh = reinterpret_cast<half_t*>(p.out) + orow;
        half_t* ol = p.out_lo ? reinterpret_cast<half_t*>(p.out_lo) + orow : nullptr;
        // output row c (channel) = acc row c (P . vhi, + Plo . vhi) + acc row D + c (Phi . vlo): flat accumulator registers f and f + 4 D8
#pragma unroll
        for (int f4 = 0; f4 < D8; ++f4) {
            const int c = 8 * f4 + 4 * hi;
            half4_t ov, lv;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int f = 4 * f4 + jj, g = f + 4 * D8;
                const float v = pin_value((acc[f >> 4][f & 15] + acc[g >> 4][g & 15]) * inv);
                ov[jj] = (half_t)v;
                lv[jj] = (half_t)(v - (float)ov[jj]);
            }
            *reinterpret_cast<half4_t*>(oh + c) = ov;
            if (ol) *reinterpret_cast<half4_t*>(ol + c) = lv;
        }
    }
}

template <int D8>
constexpr int attn_pipe_lds() {          // attn_split_pipe_kernel: ATTN_PIPE_SLOTS x (K tile [64][2 d + 8] + stacked V^T tile, whole LDS-DMA instructions)
    constexpr int D = 8 * D8, KB = KT * (2 * D + 8), VR = ((2 * D + 31) / 32) * 32, VCH = VSTR2 / 8, NVI = (2 * D * VCH + 63) / 64;
    return ATTN_PIPE_SLOTS * (KB + (VR * VCH > NVI * 64 ? VR * VCH : NVI * 64) * 8) * (int)sizeof(half_t);
}

template <int DQK, bool DBUF>
constexpr int attn_split_lds() { return (DBUF ? 2 : 1) * 2 * (KT * (DQK + 8) + ((DQK + 31) / 32) * 32 * VSTR2) * (int)sizeof(half_t); }

// 8-wave double-buffered blocks for the small head dims on long query ranges (d = 40 at N = 4096: 1.35 ms -> see DESIGN.md 3), 4-wave double-buffered
// up to d = 80, single buffer set above (2 x 89 KB would not fit at d = 160)
template <int DQK>
int launch_attn_split(const gl_attn_args& a, hipStream_t st) {
    if constexpr (DQK == 32 || DQK == 48) {
        // level-0 head dims on long query ranges: the software-pipelined kernel (key 53 = 2: the round-5 kernel, A/B)
        // (its LDS-DMA staging reaches the lo arrays through 32-bit offsets from the lower of each hi / lo pair)
        auto near32 = [](const void* x, const void* y, int64_t span) {
            const int64_t dlt = reinterpret_cast<const char*>(x) - reinterpret_cast<const char*>(y);
            return (dlt < 0 ? -dlt : dlt) + span < (int64_t)0xFFFF0000LL;
        };
        const bool reach = near32(a.k, a.k_lo, (int64_t)64 * a.ldk * 2 + 256) && near32(a.vt, a.vt_lo, (int64_t)a.d * a.ldvt * 2 + 256);
        if (a.Nq >= 512 && g_attn_split_var == 0 && (a.d == 32 || a.d == 40 || a.d == 48) && reach) {
            const dim3 grid(gl_cdiv(a.Nq, 256) * a.H * a.B);
            if (a.d == 32) attn_split_pipe_kernel<4><<<grid, dim3(512), attn_pipe_lds<4>(), st>>>(a);
            else if (a.d == 40) attn_split_pipe_kernel<5><<<grid, dim3(512), attn_pipe_lds<5>(), st>>>(a);
            else attn_split_pipe_kernel<6><<<grid, dim3(512), attn_pipe_lds<6>(), st>>>(a);
            GL_CHECK_LAUNCH();
            return 0;
        }
    }
    if constexpr (DQK <= 48) {
        if (a.Nq >= 512 && g_attn_split_var != 1) {
            attn_split_kernel<DQK, 8, true><<<dim3(gl_cdiv(a.Nq, 256) * a.H * a.B), dim3(512), attn_split_lds<DQK, true>(), st>>>(a);
            GL_CHECK_LAUNCH();
            return 0;
        }
    }
    if constexpr (DQK <= 80) attn_split_kernel<DQK, 4, true><<<dim3(gl_cdiv(a.Nq, 128) * a.H * a.B), dim3(256), attn_split_lds<DQK, true>(), st>>>(a);
    else attn_split_kernel<DQK, 4, false><<<dim3(gl_cdiv(a.Nq, 128) * a.H * a.B), dim3(256), attn_split_lds<DQK, false>(), st>>>(a);
    GL_CHECK_LAUNCH();
    return 0;
}

template <int DQK>
int set_attr_split() {
    hipError_t e = hipSuccess;
    if constexpr (DQK <= 48) e = hipFuncSetAttribute((const void*)attn_split_kernel<DQK, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_split_lds<DQK, true>());
    if constexpr (DQK <= 80) {
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_split_kernel<DQK, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_split_lds<DQK, true>());
    } else {
        e = hipFuncSetAttribute((const void*)attn_split_kernel<DQK, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_split_lds<DQK, false>());
    }
    return e == hipSuccess ? 0 : (int)e;
}

// V [B, Nk, *] -> V^T [B, H, d, ldvt] with zero fill of keys >= Nk.  One block per (64 keys, head, sample): 16-byte
// global loads (8 channels of one key), a [64][d + 2] LDS tile (row stride chosen so that the 8 key groups of one
// channel column fall on 8 different banks), 16-byte global stores (8 consecutive keys of one channel; 8 lanes
// cover a 128-byte line of a V^T row).
__global__ __launch_bounds__(256) void transpose_v_kernel(const half_t* __restrict__ v, int64_t v_bstride, int ldv,
                                                          half_t* __restrict__ vt, int ldvt, int H, int d, int Nk) {
    __shared__ __attribute__((aligned(16))) half_t tile[64 * 162];
    const int key0 = blockIdx.x * 64;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int tstr = d + 2;
    const int dch = d >> 3;                       // 16-byte chunks per key
    const half_t* src = v + (size_t)b * v_bstride + (size_t)h * d;
    for (int idx = threadIdx.x; idx < 64 * dch; idx += 256) {
        const int key = idx / dch;
        const int ch = idx - key * dch;
        uint4 raw = make_uint4(0u, 0u, 0u, 0u);
        if (key0 + key < Nk) raw = ld16(src + (size_t)(key0 + key) * ldv + ch * 8);
        unsigned* t32 = reinterpret_cast<unsigned*>(tile + key * tstr + ch * 8);      // 4-byte aligned (tstr even)
        t32[0] = raw.x; t32[1] = raw.y; t32[2] = raw.z; t32[3] = raw.w;
    }
    __syncthreads();
    half_t* dst = vt + (size_t)(b * H + h) * d * ldvt;
    for (int idx = threadIdx.x; idx < d * 8; idx += 256) {
        const int c = idx >> 3;
        const int kg = idx & 7;
        if (key0 + kg * 8 < ldvt) {               // ldvt is a multiple of 8: whole 8-key groups
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = tile[(kg * 8 + j) * tstr + c];
            st16(dst + (size_t)c * ldvt + key0 + kg * 8, *reinterpret_cast<uint4*>(&o));
        }
    }
}

template <int DQK, int QT, int NW = 4>
int launch_attn(const gl_attn_args& a, hipStream_t st) {
    dim3 grid(gl_cdiv(a.Nq, 32 * NW * QT) * a.H * a.B);
    const int flags = g_attn_setprio < 0 ? (DQK <= 48 ? 1 : 0) : g_attn_setprio;
    // the C-init form needs 16 more registers: measured +9 % at d = 80 (35.1 vs 38.4 us, N = 1024) but -12 % at d = 40, where
    // the kernel lives on 8 waves per SIMD (60 -> 76 registers drops it to 6)
    if (a.q_prescaled && DQK >= 80) attn_kernel<DQK, QT, NW, 1><<<grid, dim3(64 * NW), 0, st>>>(a, flags);
    else if (a.q_prescaled && g_attn_padmax && a.d + 8 == DQK) attn_kernel<DQK, QT, NW, 2><<<grid, dim3(64 * NW), 0, st>>>(a, flags);
    else attn_kernel<DQK, QT, NW, 0><<<grid, dim3(64 * NW), 0, st>>>(a, flags);
    GL_CHECK_LAUNCH();
    return 0;
}

// 4-wave (128-query) or 8-wave (256-query) blocks.  Round 5: a barrier-enforced two-group ping-pong form (waves w / w + 4 one phase
// apart: P.V(i-1) + Q.K^T(i) MFMAs in one group while the other runs the softmax of its tile; K / V^T staging split by group,
// bitwise equal to this kernel) measured 248 vs 230 us at d = 40, N = 4096 and 35.6 vs 31.6 us at d = 80 (profiles/r5_ab_attn_pingpong.txt);
// removed.  Round-1 variants that lost (64 queries per wave: 280 registers ->
// 1 wave/SIMD, 651 vs 479 us; a software-pipelined S(t+1) || softmax(t) kernel with a 3-slot LDS ring: -10 % at d = 40)
// are recorded in DESIGN.md and no longer compiled.
template <int DQK>
int launch_attn_auto(const gl_attn_args& a, hipStream_t st) {
    if constexpr (DQK <= 80) {
        // 8 waves (256 queries) per block halve the K / V^T tile traffic per query: 355 -> 326 us at d = 40,
        // N = 4096 (with the XCD-aware block order)
        // (round 4: also for the 77-key text cross-attention -- forward -0.03 ms; key 3 = 5 restores the Nk >= 512 condition)
        if ((g_attn_qt2 == 3 && a.Nq >= 256) || (g_attn_qt2 == 0 && a.Nq >= 512) || (g_attn_qt2 == 5 && a.Nq >= 512 && a.Nk >= 512))
            return launch_attn<DQK, 1, 8>(a, st);
    }
    return launch_attn<DQK, 1>(a, st);
}

}  // namespace

extern "C" int gl_attention(const gl_attn_args* a, void* stream) {
    if (!a || !a->q || !a->k || !a->vt || !a->out) return GL_ERR_BAD_ARG;
    if (a->d <= 0 || a->d > 160 || (a->d % 8) != 0 || a->Nq <= 0 || a->Nk <= 0) return GL_ERR_BAD_ARG;
    if ((a->ldq % 8) || (a->ldk % 8) || (a->ldvt % 8) || (a->ldo % 4)) return GL_ERR_BAD_ARG;
    if (a->ldvt < ((a->Nk + 63) / 64) * 64) return GL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int d = a->d;
    if (a->q_lo != nullptr || a->k_lo != nullptr || a->vt_lo != nullptr) {
        if (!a->q_lo || !a->k_lo || !a->vt_lo || (a->ldo % 4)) return GL_ERR_BAD_ARG;
        if (d <= 16) return launch_attn_split<16>(*a, st);
        if (d <= 32) return launch_attn_split<32>(*a, st);
        if (d <= 48) return launch_attn_split<48>(*a, st);
        if (d <= 64) return launch_attn_split<64>(*a, st);
        if (d <= 80) return launch_attn_split<80>(*a, st);
        if (d <= 128) return launch_attn_split<128>(*a, st);
        return launch_attn_split<160>(*a, st);
    }
    if (d <= 16) return launch_attn_auto<16>(*a, st);
    if (d <= 32) return launch_attn_auto<32>(*a, st);
    if (d <= 48) return launch_attn_auto<48>(*a, st);
    if (d <= 64) return launch_attn_auto<64>(*a, st);
    if (d <= 80) return launch_attn_auto<80>(*a, st);
    if (d <= 128) return launch_attn_auto<128>(*a, st);
    return launch_attn_auto<160>(*a, st);
}

template <int D8>
static int set_attr_pipe() {
    return hipFuncSetAttribute((const void*)attn_split_pipe_kernel<D8>, hipFuncAttributeMaxDynamicSharedMemorySize, attn_pipe_lds<D8>()) == hipSuccess ? 0 : GL_ERR_UNSUPPORTED;
}

extern "C" int gl_init_attn(void) {
    if (int e = set_attr_pipe<4>()) return e;
    if (int e = set_attr_pipe<5>()) return e;
    if (int e = set_attr_pipe<6>()) return e;
    // the split-fp16 kernels' dynamic LDS (above the 64 KB static limit for the double-buffered and the large-head-dim forms)
    int e;
    if ((e = set_attr_split<16>()) || (e = set_attr_split<32>()) || (e = set_attr_split<48>()) || (e = set_attr_split<64>()) || (e = set_attr_split<80>()) ||
        (e = set_attr_split<128>()) || (e = set_attr_split<160>()))
        return e;
    return 0;
}

extern "C" int gl_transpose_v(const void* v, int64_t v_bstride, int32_t ldv, void* vt, int32_t ldvt, int32_t B,
                              int32_t H, int32_t d, int32_t Nk, void* stream) {
    if (!v || !vt || d <= 0 || d > 160 || (d % 8) || Nk <= 0 || ldvt < Nk) return GL_ERR_BAD_ARG;
    if ((ldv % 8) || (ldvt % 8) || (v_bstride % 8)) return GL_ERR_BAD_ARG;        // 16-byte loads / stores
    dim3 grid(gl_cdiv(ldvt, 64), H, B);
    transpose_v_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const half_t*>(v), v_bstride, ldv, reinterpret_cast<half_t*>(vt), ldvt, H, d, Nk);
    GL_CHECK_LAUNCH();
    return 0;
}
