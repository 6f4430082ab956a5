// Reward scoring arithmetic of the RL rollout: models/policy.py:106-124 (CLIP text-image and image-image
// cosine similarities of L2-normalised features) and tools/aesthetic.py:15-31 (AestheticMLP: five Linear layers,
// 768 -> 1024 -> 128 -> 64 -> 16 -> 1; the Dropouts are identities in eval mode, there are no activations) on the
// re-normalised predicted-image embedding (policy.py:120-122, tools/aesthetic.py `normalized`).
//
// One block per rollout sample, everything in fp32 (the reference runs this part in fp32), fixed summation order
// (deterministic).  It is a few MFLOP: the point is that the rollout's scoring tail needs no host round trip and no
// framework ops between the decoded images' CLIP features and the scalar rewards.  The CLIP towers that PRODUCE the
// features, and the IoU / DocSim layout rewards (CPU python), stay with the caller.
#include "common.h"
#include "gligen_hip.h"

namespace {

constexpr int RW_MAX_D = 1024;

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// y[n] = b[n] + sum_k W[n][k] x[k]: one wave per output row, lanes stride over k (coalesced), wave-shuffle reduction
__device__ __forceinline__ void linear_layer(const float* __restrict__ W, const float* __restrict__ b, const float* x, float* y, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = wave; n < N; n += 4) {
        const float* w = W + (size_t)n * K;
        float s = 0.0f;
        for (int k = lane; k < K; k += 64) s = fmaf(w[k], x[k], s);
        s = wave_sum(s);
        if (lane == 0) y[n] = s + b[n];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void reward_score_kernel(gl_reward_args a) {
    __shared__ float xt[RW_MAX_D], xp[RW_MAX_D], xg[RW_MAX_D];
    __shared__ float h1[1024], h2[128], h3[64], h4[16];
    __shared__ float red[4];
    const int b = blockIdx.x, D = a.D;
    float st = 0.0f, sp = 0.0f, sg = 0.0f;
    for (int k = threadIdx.x; k < D; k += 256) {
        const float t = a.txt[(size_t)b * D + k], p = a.img_pred[(size_t)b * D + k], g = a.img_gt[(size_t)b * D + k];
        xt[k] = t; xp[k] = p; xg[k] = g;
        st = fmaf(t, t, st); sp = fmaf(p, p, sp); sg = fmaf(g, g, sg);
    }
    // F.normalize(x, dim=-1): x / max(||x||_2, 1e-12)   (policy.py:113-115)
    const float nt = fmaxf(sqrtf(block_sum(st, red)), 1e-12f);
    const float np_ = fmaxf(sqrtf(block_sum(sp, red)), 1e-12f);
    const float ng = fmaxf(sqrtf(block_sum(sg, red)), 1e-12f);
    float dti = 0.0f, dii = 0.0f, s2 = 0.0f;
    for (int k = threadIdx.x; k < D; k += 256) {
        const float t = xt[k] / nt, p = xp[k] / np_, g = xg[k] / ng;
        xp[k] = p;
        dti = fmaf(t, p, dti);
        dii = fmaf(g, p, dii);
        s2 = fmaf(p, p, s2);
    }
    const float sims_ti = block_sum(dti, red);      // (txt * img_pred).sum(-1)   policy.py:116
    const float sims_ii = block_sum(dii, red);      // (img_gt * img_pred).sum(-1) policy.py:117
    // aesthetic.normalized(): divide by the L2 norm once more, a zero norm counts as 1 (tools/aesthetic.py:52-57)
    float l2 = sqrtf(block_sum(s2, red));
    if (l2 == 0.0f) l2 = 1.0f;
    for (int k = threadIdx.x; k < D; k += 256) xp[k] = xp[k] / l2;
    __syncthreads();
    linear_layer(a.w1, a.b1, xp, h1, 1024, D);
    linear_layer(a.w2, a.b2, h1, h2, 128, 1024);
    linear_layer(a.w3, a.b3, h2, h3, 64, 128);
    linear_layer(a.w4, a.b4, h3, h4, 16, 64);
    if (threadIdx.x < 64) {
        float s = (threadIdx.x < 16) ? a.w5[threadIdx.x] * h4[threadIdx.x] : 0.0f;
        s = wave_sum(s);
        if (threadIdx.x == 0) {
            const float aes = s + a.b5[0];
            a.sims_ti[b] = sims_ti;
            a.sims_ii[b] = sims_ii;
            a.aesthetic[b] = aes;
            // reward = clip_reward + aes * 0.1 (+ miou * 10 + laysim * 10 added by the caller, policy.py:135)
            if (a.partial_reward) a.partial_reward[b] = (sims_ti + sims_ii) + aes * 0.1f;
        }
    }
}

}  // namespace

extern "C" int gl_reward_score(const gl_reward_args* a, void* stream) {
    if (!a || !a->txt || !a->img_pred || !a->img_gt || !a->sims_ti || !a->sims_ii || !a->aesthetic) return GL_ERR_BAD_ARG;
    if (!a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->w3 || !a->b3 || !a->w4 || !a->b4 || !a->w5 || !a->b5) return GL_ERR_BAD_ARG;
    if (a->B <= 0 || a->D <= 0 || a->D > RW_MAX_D) return GL_ERR_BAD_ARG;
    reward_score_kernel<<<dim3(a->B), dim3(256), 0, (hipStream_t)stream>>>(*a);
    GL_CHECK_LAUNCH();
    return 0;
}
extern "C" int gl_sizeof_reward_args(void) { return (int)sizeof(gl_reward_args); }
