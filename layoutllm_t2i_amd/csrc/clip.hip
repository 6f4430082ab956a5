// Small kernels of the CLIP towers used by the reward stage of the train_rl.py rollout (reference
// models/policy.py:106-113 calls transformers.CLIPModel.get_text_features / get_image_features).  The heavy work of the
// towers -- projections, MLPs, LayerNorms, the vision tower's bidirectional attention -- runs on gl_gemm / gl_layernorm /
// gl_attention; what is here are the tower-specific data movements and the text tower's short CAUSAL attention:
//   gl_clip_patchify       pixel_values fp32 [B, 3, S, S] -> fp16 patch rows [B * np, Kpad] (the patch conv as a GEMM operand)
//   gl_clip_assemble       [class_embedding ; patch embeddings] + position_embedding -> fp32 residual stream [B, 1 + np, C]
//   gl_clip_embed_tokens   token_embedding[ids] + position_embedding -> fp32 stream [B, T, C]
//   gl_clip_gather_rows    stream rows (class token / EOS token) -> fp32 [B, C]
//   gl_attention_small     attention for short sequences (T <= 128, d <= 64) with an optional causal mask, fp32 arithmetic
// HBM-bound / latency-bound kernels: 16-byte accesses where the layout allows, nothing fancy.
#include "common.h"
#include "gligen_hip.h"

namespace {

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ px, int B, int S, int P, int np_side, int Kpad,
                                                       half_t* __restrict__ out) {
    // one block per patch row; K index = (c * P + i) * P + j, zero-padded to Kpad
    const int row = blockIdx.x;
    const int np = np_side * np_side;
    const int b = row / np, pi = row - b * np;
    const int py = pi / np_side, pxx = pi - py * np_side;
    const int K = 3 * P * P;
    for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
        float v = 0.0f;
        if (k < K) {
            const int c = k / (P * P);
            const int r = k - c * P * P;
            const int i = r / P, j = r - i * P;
            v = px[(((size_t)b * 3 + c) * S + (py * P + i)) * S + (pxx * P + j)];
        }
        out[(size_t)row * Kpad + k] = (half_t)v;
    }
}

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void assemble_kernel(const half_t* __restrict__ pe, int ldpe, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, int B, int T, int C, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float* __restrict__ x) {
    // row (b, t): t == 0 -> class embedding, else patch embedding row b * (T - 1) + t - 1; + pos[t]; optional fp32 LayerNorm
    __shared__ float rowbuf[4096];
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int b = row / T, t = row - b * T;
    float s = 0.0f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float v = ((t == 0) ? cls[c] : (float)pe[((size_t)b * (T - 1) + (t - 1)) * ldpe + c]) + pos[(size_t)t * C + c];
        rowbuf[c] = v;
        s += v;
    }
    if (gamma == nullptr) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) x[(size_t)row * C + c] = rowbuf[c];
        return;
    }
    const float mean = block_sum256(s, red) / (float)C;
    float q = 0.0f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { const float d = rowbuf[c] - mean; q += d * d; }
    const float rstd = rsqrtf(block_sum256(q, red) / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[(size_t)row * C + c] = (rowbuf[c] - mean) * rstd * gamma[c] + beta[c];
}

__global__ __launch_bounds__(256) void embed_tokens_kernel(const int* __restrict__ ids, const float* __restrict__ tok,
                                                           const float* __restrict__ pos, int B, int T, int C, int vocab,
                                                           float* __restrict__ x) {
    const int row = blockIdx.x;
    const int t = row % T;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[(size_t)row * C + c] = tok[(size_t)id * C + c] + pos[(size_t)t * C + c];
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ rows, int C,
                                                          float* __restrict__ out) {
    const int b = blockIdx.x;
    const size_t r = (size_t)rows[b];
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[(size_t)b * C + c] = x[r * ldx + c];
}

// One block per (batch, head): K and V of the head staged in LDS as fp32 rows of DS = d rounded up to 4 floats (zero-filled),
// one thread per query row, fp32 scores.  Every lane reads the SAME K / V row (an LDS broadcast, no bank conflicts), as
// 16-byte ds_read_b128 with four independent partial sums per dot product: the first version read one float per FMA in one
// dependent chain and ran at LDS latency (650 us per launch at T = 77, d = 64; this form: see DESIGN.md).
// T <= 128, d <= 64: 2 * 128 * 64 * 4 = 64 KiB of LDS at most.  Exactness (fp32 softmax, fp32 accumulation, two passes so no
// rescaling) matters more than speed here.
constexpr int AS_MAXT = 128, AS_MAXD = 64;
__global__ __launch_bounds__(128) void attention_small_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k,
                                                              const half_t* __restrict__ v, int ld, int T, int H, int d, float scale,
                                                              int causal, half_t* __restrict__ out, int ldo) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int DS = (d + 3) & ~3;
    const int n4 = DS >> 2;
    float* Ks = sm;                 // [T][DS]
    float* Vs = sm + T * DS;        // [T][DS]
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const size_t base = (size_t)b * T * ld + (size_t)h * d;
    for (int i = threadIdx.x; i < T * DS; i += blockDim.x) {
        const int t = i / DS, c = i - t * DS;
        const bool in = c < d;
        Ks[i] = in ? (float)k[base + (size_t)t * ld + c] : 0.0f;
        Vs[i] = in ? (float)v[base + (size_t)t * ld + c] : 0.0f;
    }
    __syncthreads();
    const int tq = threadIdx.x;
    if (tq >= T) return;
    float qr[AS_MAXD];
#pragma unroll
    for (int c = 0; c < AS_MAXD; ++c) qr[c] = (c < d) ? (float)q[base + (size_t)tq * ld + c] * scale : 0.0f;
    const int nk = causal ? tq + 1 : T;
    auto dot = [&](int j) -> float {
        const float4* kr = reinterpret_cast<const float4*>(Ks + j * DS);
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
        for (int c4 = 0; c4 < AS_MAXD / 4; ++c4)
            if (c4 < n4) {
                const float4 kv = kr[c4];
                s0 = fmaf(qr[4 * c4], kv.x, s0);
                s1 = fmaf(qr[4 * c4 + 1], kv.y, s1);
                s2 = fmaf(qr[4 * c4 + 2], kv.z, s2);
                s3 = fmaf(qr[4 * c4 + 3], kv.w, s3);
            }
        return (s0 + s1) + (s2 + s3);
    };
    // pass 1: row max; pass 2: exp / sum / weighted V (scores are recomputed: 2 x T x d FMAs per thread, nothing to store)
    float m = -INFINITY;
    for (int j = 0; j < nk; ++j) m = fmaxf(m, dot(j));
    float acc[AS_MAXD];
#pragma unroll
    for (int c = 0; c < AS_MAXD; ++c) acc[c] = 0.0f;
    float l = 0.0f;
    for (int j = 0; j < nk; ++j) {
        const float p = __expf(dot(j) - m);
        l += p;
        const float4* vr = reinterpret_cast<const float4*>(Vs + j * DS);
#pragma unroll
        for (int c4 = 0; c4 < AS_MAXD / 4; ++c4)
            if (c4 < n4) {
                const float4 vv = vr[c4];
                acc[4 * c4] = fmaf(p, vv.x, acc[4 * c4]);
                acc[4 * c4 + 1] = fmaf(p, vv.y, acc[4 * c4 + 1]);
                acc[4 * c4 + 2] = fmaf(p, vv.z, acc[4 * c4 + 2]);
                acc[4 * c4 + 3] = fmaf(p, vv.w, acc[4 * c4 + 3]);
            }
    }
    const float inv = 1.0f / l;
    half_t* o = out + (size_t)(b * T + tq) * ldo + (size_t)h * d;
#pragma unroll
    for (int c = 0; c < AS_MAXD; ++c)
        if (c < d) o[c] = (half_t)(acc[c] * inv);
}

}  // namespace

extern "C" int gl_clip_patchify(const float* pixel_values, int32_t B, int32_t S, int32_t patch, int32_t Kpad, void* out, void* stream) {
    if (!pixel_values || !out || B <= 0 || S <= 0 || patch <= 0 || (S % patch) || Kpad < 3 * patch * patch || (Kpad % 64)) return GL_ERR_BAD_ARG;
    const int nps = S / patch;
    patchify_kernel<<<dim3(B * nps * nps), dim3(256), 0, (hipStream_t)stream>>>(pixel_values, B, S, patch, nps, Kpad, reinterpret_cast<half_t*>(out));
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_clip_assemble(const void* patch_emb, int32_t ldpe, const float* class_emb, const float* pos_emb, int32_t B, int32_t T,
                                int32_t C, const float* ln_gamma, const float* ln_beta, float ln_eps, float* x, void* stream) {
    if (!patch_emb || !class_emb || !pos_emb || !x || B <= 0 || T <= 1 || C <= 0 || C > 4096) return GL_ERR_BAD_ARG;
    if ((ln_gamma == nullptr) != (ln_beta == nullptr)) return GL_ERR_BAD_ARG;
    assemble_kernel<<<dim3(B * T), dim3(256), 0, (hipStream_t)stream>>>(reinterpret_cast<const half_t*>(patch_emb), ldpe, class_emb, pos_emb, B, T, C,
                                                                        ln_gamma, ln_beta, ln_eps, x);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_clip_embed_tokens(const int32_t* ids, const float* tok_emb, const float* pos_emb, int32_t B, int32_t T, int32_t C,
                                    int32_t vocab, float* x, void* stream) {
    if (!ids || !tok_emb || !pos_emb || !x || B <= 0 || T <= 0 || C <= 0 || vocab <= 0) return GL_ERR_BAD_ARG;
    embed_tokens_kernel<<<dim3(B * T), dim3(256), 0, (hipStream_t)stream>>>(ids, tok_emb, pos_emb, B, T, C, vocab, x);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_clip_gather_rows(const float* x, int32_t ldx, const int32_t* rows, int32_t B, int32_t C, float* out, void* stream) {
    if (!x || !rows || !out || B <= 0 || C <= 0) return GL_ERR_BAD_ARG;
    gather_rows_kernel<<<dim3(B), dim3(256), 0, (hipStream_t)stream>>>(x, ldx, rows, C, out);
    GL_CHECK_LAUNCH();
    return 0;
}

extern "C" int gl_attention_small(const void* q, const void* k, const void* v, int32_t ld, int32_t B, int32_t T, int32_t H, int32_t d,
                                  float scale, int32_t causal, void* out, int32_t ldo, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || T <= 0 || T > AS_MAXT || d <= 0 || d > AS_MAXD) return GL_ERR_BAD_ARG;
    const size_t lds = (size_t)2 * T * ((d + 3) & ~3) * sizeof(float);
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)attention_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                2 * AS_MAXT * AS_MAXD * (int)sizeof(float)) != hipSuccess) return GL_ERR_UNSUPPORTED;
    attention_small_kernel<<<dim3(B * H), dim3(128), lds, (hipStream_t)stream>>>(
        reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k), reinterpret_cast<const half_t*>(v), ld, T, H, d, scale, causal,
        reinterpret_cast<half_t*>(out), ldo);
    GL_CHECK_LAUNCH();
    return 0;
}
