// Flash-style fused attention for gfx950: out = softmax(scale * Q K^T) V without the N x N tensor.
//
// Replaces SelfAttention.forward (attention.py:164-176) and CrossAttention.forward (:128-141), which
// materialise `sim` as (B*8) x N x N fp32 (2.1 GB at the 64x64 level for B=4).
//
// Work decomposition: grid = (ceil(Nq/128), H, B); 256 threads = 4 waves, each wave owns 32 queries.
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

namespace {

constexpr int KT = 64;          // keys per tile
constexpr int VSTR = KT + 4;    // V^T LDS row stride in halfs (136 B: odd number of 8-byte slots)

template <int DQK>
__global__ __launch_bounds__(256) void attn_kernel(gl_attn_args p) {
    constexpr int NKS = DQK / 16;            // MFMA k-steps over the head dim
    constexpr int NDT = (DQK + 31) / 32;     // 32-wide head-dim tiles of O
    constexpr int KSTR = DQK + 8;            // K LDS row stride in halfs (odd number of 16-byte slots)
    constexpr int KCH = DQK / 8;             // 16-byte chunks per K row
    constexpr int K_ITEMS = KT * KCH;        // chunks in a K tile
    constexpr int K_PER_T = (K_ITEMS + 255) / 256;
    constexpr int V_ITEMS = NDT * 32 * (KT / 8);
    constexpr int V_PER_T = (V_ITEMS + 255) / 256;

    __shared__ __attribute__((aligned(16))) half_t Ks[KT * KSTR];
    __shared__ __attribute__((aligned(16))) half_t Vs[NDT * 32 * VSTR];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ql = lane & 31;
    const int hi = lane >> 5;
    const int b = blockIdx.z;
    const int h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int d = p.d, Nq = p.Nq, Nk = p.Nk;

    const half_t* __restrict__ Qg = reinterpret_cast<const half_t*>(p.q) + (size_t)b * p.q_bstride + (size_t)h * d;
    const half_t* __restrict__ Kg = reinterpret_cast<const half_t*>(p.k) + (size_t)b * p.k_bstride + (size_t)h * d;
    const half_t* __restrict__ Vg = reinterpret_cast<const half_t*>(p.vt) + (size_t)(b * p.H + h) * d * p.ldvt;

    // Q fragments (B operand: row = query, k = head-dim chunk 2*ks + hi)
    half8_t qf[NKS];
    {
        int q = q0 + ql;
        if (q >= Nq) q = Nq - 1;
        const half_t* qrow = Qg + (size_t)q * p.ldq;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int c0 = (2 * ks + hi) * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (c0 < d) v = ld16(qrow + c0);
            qf[ks] = *reinterpret_cast<half8_t*>(&v);
        }
    }

    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
    float m_run = -INFINITY;
    float l_run = 0.0f;
    const float c_scale = p.scale * 1.4426950408889634f;

    uint4 rk[K_PER_T], rv[V_PER_T];

    auto load_tile = [&](int key0) {
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i) {
            const int idx = tid + 256 * i;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (idx < K_ITEMS) {
                const int row = idx / KCH;
                const int c = idx - row * KCH;
                const int key = key0 + row;
                if (key < Nk && c * 8 < d) v = ld16(Kg + (size_t)key * p.ldk + c * 8);
            }
            rk[i] = v;
        }
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) {
            const int idx = tid + 256 * i;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (idx < V_ITEMS) {
                const int row = idx >> 3;      // head-dim column
                const int c = idx & 7;         // 8-key chunk
                if (row < d) v = ld16(Vg + (size_t)row * p.ldvt + key0 + c * 8);
            }
            rv[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < K_PER_T; ++i) {
            const int idx = tid + 256 * i;
            if (idx < K_ITEMS) {
                const int row = idx / KCH;
                const int c = idx - row * KCH;
                st16(Ks + row * KSTR + c * 8, rk[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < V_PER_T; ++i) {
            const int idx = tid + 256 * i;
            if (idx < V_ITEMS) {
                const int row = idx >> 3;
                const int c = idx & 7;
                uint2* dst = reinterpret_cast<uint2*>(Vs + row * VSTR + c * 8);   // 8-byte aligned only
                dst[0] = make_uint2(rv[i].x, rv[i].y);
                dst[1] = make_uint2(rv[i].z, rv[i].w);
            }
        }
    };

    const int ntiles = (Nk + KT - 1) / KT;
    load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KT;
        __syncthreads();            // all waves finished reading the previous tile
        store_tile();
        __syncthreads();
        if (t + 1 < ntiles) load_tile(key0 + KT);

        // ---- S^T = K . Q^T : two 32-key halves
        f32x16 s[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kh][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (kh * 32 + ql) * KSTR + (2 * ks + hi) * 8);
                s[kh] = mfma32(kf, qf[ks], s[kh]);
            }
        }
        // ---- online softmax (this lane: query ql, keys kh*32 + (r&3) + 8*(r>>2) + 4*hi)
        float tmax = -INFINITY;
        const bool tail = (key0 + KT > Nk);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = s[kh][r] * c_scale;
                if (tail) {
                    const int key = key0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= Nk) v = -INFINITY;
                }
                s[kh][r] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.0f;
        half8_t pf[4];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[kh][r] - m_new);
                psum += pv;
                pf[kh * 2 + (r >> 3)][r & 7] = (half_t)pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;

        // ---- O^T += V^T . P^T : k-step j covers keys 16j..16j+15 in the order (e&3) + 8*(e>>2) + 4*hi
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const half_t* vrow = Vs + (dt * 32 + ql) * VSTR + 16 * j + 4 * hi;
                const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
                const uint2 hi8 = *reinterpret_cast<const uint2*>(vrow + 8);
                uint4 v4 = make_uint4(lo.x, lo.y, hi8.x, hi8.y);
                const half8_t vf = *reinterpret_cast<half8_t*>(&v4);
                o[dt] = mfma32(vf, pf[j], o[dt]);
            }
        }
    }

    // ---- finalize: O /= l, write fp16.  lane holds head-dim columns dt*32 + 8*rg + 4*hi + {0..3}
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + ql;
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
                    for (int jj = 0; jj < 4; ++jj) ov[jj] = (half_t)(o[dt][rg * 4 + jj] * inv);
                    *reinterpret_cast<half4_t*>(orow + c) = ov;
                }
            }
    }
}

// V [B, Nk, *] -> V^T [B, H, d, ldvt] with zero fill of keys >= Nk.
__global__ __launch_bounds__(256) void transpose_v_kernel(const half_t* __restrict__ v, int64_t v_bstride, int ldv,
                                                          half_t* __restrict__ vt, int ldvt, int H, int d, int Nk) {
    __shared__ half_t tile[64 * 162];
    const int key0 = blockIdx.x * 64;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int tstr = d + 2;
    const half_t* src = v + (size_t)b * v_bstride + (size_t)h * d;
    for (int idx = threadIdx.x; idx < 64 * d; idx += 256) {
        const int key = idx / d;
        const int c = idx - key * d;
        half_t val = (half_t)0.0f;
        if (key0 + key < Nk) val = src[(size_t)(key0 + key) * ldv + c];
        tile[key * tstr + c] = val;
    }
    __syncthreads();
    half_t* dst = vt + (size_t)(b * H + h) * d * ldvt;
    for (int idx = threadIdx.x; idx < 64 * d; idx += 256) {
        const int c = idx >> 6;
        const int key = idx & 63;
        if (key0 + key < ldvt) dst[(size_t)c * ldvt + key0 + key] = tile[key * tstr + c];
    }
}

template <int DQK>
int launch_attn(const gl_attn_args& a, hipStream_t st) {
    dim3 grid(gl_cdiv(a.Nq, 128), a.H, a.B);
    attn_kernel<DQK><<<grid, dim3(256), 0, st>>>(a);
    GL_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int gl_attention(const gl_attn_args* a, void* stream) {
    if (!a || !a->q || !a->k || !a->vt || !a->out) return GL_ERR_BAD_ARG;
    if (a->d <= 0 || a->d > 160 || (a->d % 8) != 0 || a->Nq <= 0 || a->Nk <= 0) return GL_ERR_BAD_ARG;
    if ((a->ldq % 8) || (a->ldk % 8) || (a->ldvt % 8) || (a->ldo % 4)) return GL_ERR_BAD_ARG;
    if (a->ldvt < ((a->Nk + 63) / 64) * 64) return GL_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int d = a->d;
    if (d <= 16) return launch_attn<16>(*a, st);
    if (d <= 32) return launch_attn<32>(*a, st);
    if (d <= 48) return launch_attn<48>(*a, st);
    if (d <= 64) return launch_attn<64>(*a, st);
    if (d <= 80) return launch_attn<80>(*a, st);
    if (d <= 128) return launch_attn<128>(*a, st);
    return launch_attn<160>(*a, st);
}

extern "C" int gl_transpose_v(const void* v, int64_t v_bstride, int32_t ldv, void* vt, int32_t ldvt, int32_t B,
                              int32_t H, int32_t d, int32_t Nk, void* stream) {
    if (!v || !vt || d <= 0 || d > 160 || Nk <= 0 || ldvt < Nk) return GL_ERR_BAD_ARG;
    dim3 grid(gl_cdiv(ldvt, 64), H, B);
    transpose_v_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>(
        reinterpret_cast<const half_t*>(v), v_bstride, ldv, reinterpret_cast<half_t*>(vt), ldvt, H, d, Nk);
    GL_CHECK_LAUNCH();
    return 0;
}
