/*
 * gligen_hip.h -- C ABI of the MI355X (gfx950) kernels for the layout-conditioned denoising hot path
 * of LayoutLLM-T2I (modified GLIGEN UNet + PLMS sampler).
 *
 * The reference has no native layer: every op on the path is a torch call inside an
 * nn.Module.forward.  Each entry point below therefore names the reference *Python* code it replaces
 * (file:line relative to the reference root).  All pointers are raw DEVICE pointers (owned by the
 * caller, e.g. torch tensors); `stream` is a hipStream_t.  Every function returns 0 on success or a
 * hipError_t / negative gl error code; nothing throws across the ABI.
 *
 * Activation layout is token-major ("NHWC"): a feature map is [B, H*W, C] fp16 with C contiguous, so
 * 1x1 convs and Linear layers are plain GEMMs and the 3x3 conv is an implicit GEMM.  Weights are fp16
 * [N, K] row-major (torch Linear layout); 3x3 conv weights are repacked to [Cout, Cin/64, 3, 3, 64].
 * Accumulation and all normalisation/softmax statistics are fp32.
 */
#ifndef GLIGEN_HIP_H
#define GLIGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GL_ABI_VERSION 15

/* error codes (negative; positive values are hipError_t) */
#define GL_ERR_BAD_ARG (-1)
#define GL_ERR_UNSUPPORTED (-2)

/* ---- epilogues of gl_gemm / gl_conv3x3 ------------------------------------------------------- */
enum gl_epilogue {
    GL_EPI_BIAS = 0,      /* out = acc + bias                                                        */
    GL_EPI_SILU = 1,      /* out = silu(acc + bias)             PositionNet MLP, time_embed           */
    GL_EPI_GEGLU = 2,     /* out[:, j] = (acc_x + b) * gelu_erf(acc_gate + b); weights row-interleaved */
                          /*   in blocks of 32 (x rows, gate rows); N = 8C packed, out width N/2       */
    GL_EPI_RES = 3,       /* out = acc + bias + res                                                    */
    GL_EPI_GATE_RES = 4,  /* out = res + gate[0] * (acc + bias)   gated fuser residual                 */
    GL_EPI_ROWBIAS = 5    /* out = acc + bias + rowbias[m / rows_per_sample, n]   (+ time embedding)   */
};

enum gl_out_mode {
    GL_OUT_F16_ROWMAJOR = 0, /* out[m * ldc + n] fp16                                   */
    GL_OUT_F32_NCHW = 1,     /* out[(b * N + n) * HW + p] fp32, m = b * HW + p           */
    GL_OUT_F32_ROWMAJOR = 2  /* out[m * ldc + n] fp32: the RESIDUAL STREAM (ResBlock / transformer-block sums,
                                openaimodel.py:231, attention.py:395-402,446) is kept in fp32 so that ~100 chained
                                residual adds are not rounded to fp16 each; out2 (optional) gets an fp16 copy for
                                consumers that feed it to the matrix cores (down / up convs)                        */
    ,
    GL_OUT_F16_HILO = 3      /* split-fp16 operand for a following 1x1 product: out[m * ldc + n] = hi = fp16(v) and
                                out[m * ldc + N + n] = lo = fp16(v - hi)  (ldc >= 2N).  The consumer runs gl_gemm with
                                K = 2N against the weight stored twice, [W | W]: x.W = hi.W + lo.W restores ~22 mantissa
                                bits of the activation operand (proj_out of attention.py:444-446)            */
};

/*
 * gl_gemm: out[M, N] = A[M, K] . W[N, K]^T (+ epilogue).   K % 64 == 0.
 * Replaces nn.Linear / 1x1 nn.Conv2d calls: attention.py:43 (GEGLU proj), :61 (ff out), :124-126,
 * :160-162 (q/k/v), :143,:178 (to_out), :228 (fuser.linear), :440,:445 (proj_in/out);
 * openaimodel.py:190-194 (skip 1x1), :220 (emb_layers), :428-429 (time_embed);
 * text_grounding_net.py:41 (PositionNet MLP).
 * Two-source A (a2 != NULL): k < ksplit reads a[m*lda + k], else a2[m*lda2 + (k - ksplit)] -- this folds
 * th.cat([h, hs.pop()], 1) (openaimodel.py:456) into the skip-connection GEMM.  ksplit % 64 == 0.
 */
typedef struct gl_gemm_args {
    const void* a;      int32_t lda;
    const void* a2;     int32_t lda2;  int32_t ksplit;
    const void* w;                      /* fp16 [N, K] */
    const float* bias;                  /* fp32 [N] or NULL */
    int32_t M, N, K;
    int32_t epi;                        /* enum gl_epilogue */
    int32_t out_mode;                   /* enum gl_out_mode */
    void* out;          int32_t ldc;    /* fp16 row stride (elements); GEGLU: stride of the N/2-wide output */
    const void* res;    int32_t ldres;  /* fp16 residual [M, N] */
    const float* gate;                  /* device scalar for GL_EPI_GATE_RES */
    const void* rowbias; int32_t ld_rowbias; int32_t rows_per_sample;  /* fp16 [B, N] */
    int32_t hw;                         /* GL_OUT_F32_NCHW: pixels per sample */
    /* optional split-K scratch: when the (M, N) grid would leave most of the 256 CUs idle (8x8 / 16x16
     * levels: 40-160 tiles) and K is long, K is cut into slices that write fp32 partial tiles here and a
     * second kernel reduces them and applies the epilogue.  NULL disables splitting. */
    void* workspace;    int64_t workspace_bytes;
    int32_t res_f32;                    /* != 0: res is fp32 [M, ldres] (residual stream), else fp16 */
    void* out2;         int32_t ldc2;   /* optional fp16 copy [M, N] of a GL_OUT_F32_ROWMAJOR output; NULL = none */
    /* optional transposed tail (fused QKV projection, attention.py:160-162): columns [vt_col0, N) are NOT written to
     * out but as gl_attention's V^T operand, vt[((b * vt_H + h) * vt_d + c) * vt_ld + key] with b = m / vt_rows,
     * key = m % vt_rows, (h, c) = divmod(n - vt_col0, vt_d).  fp16 row-major out, epi BIAS only, vt_col0 % 64 == 0. */
    void* vt;           int32_t vt_col0, vt_rows, vt_d, vt_ld, vt_H;
    /* weight reuse along K (split-fp16 operands): with kwrap != 0 column k of the product reads w[n * ldw + k] for k < kwrap
     * and w[n * ldw + k - kwrap] beyond (ONE step back).  K = 2 * kwrap: A = [hi | lo] is multiplied against [W | W] without storing W
     * twice.  K = 3 * kwrap (ldw >= 2 * kwrap, weight rows [Whi | Wlo]): columns [0, kwrap) twice, then [kwrap, 2 * kwrap) -- with
     * A = [xhi | xlo] and the second source a2 = xhi at ksplit = 2 * kwrap that is the three-pass product xhi.Whi + xlo.Whi + xhi.Wlo.
     * kwrap % 64 == 0, kwrap < K <= 3 * kwrap.  ldw == 0: the row stride is K (kwrap == 0) or kwrap.  gl_gemm only (ignored by gl_conv3x3). */
    int32_t ldw, kwrap;
    int32_t rowbias_f32;                /* != 0: rowbias is fp32 [B, N] (ld_rowbias in floats) instead of fp16: the ResBlock's emb_layers output is
                                           added to the conv result unrounded (openaimodel.py:220-226) */
    /* ABI 15 (strict mode): with out_mode GL_OUT_F16_HILO the transposed tail takes BOTH halves -- vt receives fp16(v) and vt_lo (same layout)
     * fp16(v - fp16(v)) of the V columns [vt_col0, N): the two V^T operands of the split-fp16 attention (gl_attn_args.vt / vt_lo) straight
     * from the fused QKV projection's epilogue.  Required (non-NULL) exactly when vt != NULL and out_mode == GL_OUT_F16_HILO.  Implemented by the 8-wave
     * kernel's epilogue only: GL_ERR_UNSUPPORTED for launches it does not take (M < 256, K % 64, a tail that does not start on a wave's column range). */
    void* vt_lo;
} gl_gemm_args;

/*
 * gl_ff_fused: the whole FeedForward of a BasicTransformerBlock / GatedSelfAttentionDense in ONE launch (attention.py:38-62,
 * called at :231-232 and :401):   out = res (+|gate *) ( GEGLU(x . W1^T + b1) . W2^T + b2 )
 * for narrow channel widths (C in {64, 128, 192, 256, 320}; gl_ff_fused_supported(C)) -- the level whose [M, 4C]
 * GEGLU intermediate (84 MB at 64x64x320) would otherwise make a round trip through HBM between two gl_gemm launches.
 * w1 / b1 are gl_gemm's GL_EPI_GEGLU operands (rows interleaved in blocks of 32: x rows, gate rows), w2 is [C, 4C].
 * gate == NULL: out = y + res (GL_EPI_RES);  gate != NULL: out = res + gate[0] * y (GL_EPI_GATE_RES).
 * Agrees with the two-launch form to fp32 accumulation rounding (same fp16 intermediate, other summation order).
 */
typedef struct gl_ff_args {
    const void* x;      int32_t ldx;    /* fp16 [M, C] (already normalised rows) */
    const void* w1;     const float* b1;/* fp16 [8C, C] packed, fp32 [8C] packed   */
    const void* w2;     const float* b2;/* fp16 [C, 4C], fp32 [C]                  */
    const void* res;    int32_t ldres;  int32_t res_f32;   /* residual rows: fp32 stream (res_f32 != 0) or fp16 */
    const float* gate;                  /* device scalar or NULL */
    void* out;          int32_t ldc;    int32_t out_mode;  /* GL_OUT_F16_ROWMAJOR or GL_OUT_F32_ROWMAJOR */
    int32_t M, C;
} gl_ff_args;
int gl_ff_fused(const gl_ff_args* args, void* stream);
int gl_ff_fused_supported(int32_t C);
/* 1 when the fused form is the faster choice for [M, C] on the current device (enough 128-row blocks to fill whole rounds
 * of CUs); the engine uses it, and so must anything that wants to reproduce the engine's results bit for bit */
int gl_ff_fused_applicable(int32_t C, int32_t M);
int gl_sizeof_ff_args(void);

/*
 * gl_conv3x3: 3x3, pad 1 convolution as implicit GEMM over NHWC fp16 input [B, Hin, Win, Cin]
 * (Cin % 64 == 0; weights [Cout, Cin/64, 3, 3, 64], i.e. K = (channel block, tap, channel)).  M = B*Hout*Wout, N = Cout, K = 9*Cin.
 *   stride 1            : ResBlock convs openaimodel.py:158,184; conv_in :299; out conv :388
 *   stride 2            : Downsample.op openaimodel.py:105-107,:114
 *   upsample2x != 0     : F.interpolate(nearest, x2) + conv, openaimodel.py:82-84 (input is Hout/2 x Wout/2)
 * Epilogue fields are those of gl_gemm (the `g` member; g.a/g.K/g.M are ignored and derived here).
 */
typedef struct gl_conv_args {
    const void* in;
    int32_t B, Hin, Win, Cin;
    int32_t Hout, Wout;
    int32_t stride;        /* 1 or 2 */
    int32_t upsample2x;    /* 0 or 1 (stride must be 1) */
    gl_gemm_args g;        /* w, bias, N, epi, out_mode, out, ldc, res, ldres, rowbias..., hw */
    /* split-fp16 operands (ABI 14; strict mode).  in_split = 2: the input pixels are [hi | lo] rows of 2 Cin channels
     * (hi = fp16(x), lo = fp16(x - hi), e.g. gl_groupnorm_ex's out / out_lo with ldo = 2 Cin) and the K walk visits both halves against
     * the same weight: x.W = hi.W + lo.W (~22 mantissa bits of the activation); in_split = 3 (needs w_split): a third pass hi.Wlo.
     * w_split != 0: the weight rows are [Whi | Wlo], fp16 [Cout, 2, Cin/64, 3, 3, 64] (Wlo = fp16(W - Whi)); in_split 0 / 2 read Whi only. */
    int32_t in_split, w_split;
} gl_conv_args;

/*
 * gl_attention: out[b, q, h*d : (h+1)*d] = softmax_k(scale * Q.K^T) . V   (flash-style, no S x S tensor).
 * Replaces SelfAttention.forward attention.py:164-176 and CrossAttention.forward :128-141 (mask=None).
 * Q [B, Nq, *] fp16 rows of stride ldq, head h at column offset h*d; K likewise (ldk).
 * V is passed TRANSPOSED per head: vt[((b*H + h)*d + c) * ldvt + key], ldvt % 8 == 0 and >= Nk rounded up to 64 (the
 * pad keys are never used: the kernel masks them, so their contents are arbitrary).  d in {8..160}, d % 8 == 0.
 * Batch strides are in elements.
 */
typedef struct gl_attn_args {
    const void* q;  int64_t q_bstride;  int32_t ldq;
    const void* k;  int64_t k_bstride;  int32_t ldk;
    const void* vt; int32_t ldvt;
    void* out;      int64_t o_bstride;  int32_t ldo;
    int32_t B, H, d, Nq, Nk;
    float scale;
    int32_t q_prescaled;   /* != 0: scale * log2(e) is already folded into Q (the packer folds it into the q projection
                              weights); `scale` is then ignored and the running max is subtracted inside the MFMA */
    /* split-fp16 attention (ABI 14; strict mode): with q_lo != NULL every operand is hi + lo -- q_lo / k_lo / vt_lo hold the fp16 residuals
     * in the layouts (strides, batch strides) of q / k / vt, the logits are q.k = qhi.khi + qlo.khi + qhi.klo, the probabilities are split
     * P = Phi + Plo in registers, O = Phi.Vhi + Plo.Vhi + Phi.Vlo, and out / out_lo (same ldo, out_lo may be NULL) receive fp16(O) and
     * fp16(O - fp16(O)).  All three of q_lo, k_lo, vt_lo must be given. */
    const void* q_lo; const void* k_lo; const void* vt_lo; void* out_lo;
} gl_attn_args;

/* gl_transpose_v: V [B, Nk, *] (row stride ldv, head h at column h*d) -> vt [B, H, d, ldvt], zero-fills keys
 * [Nk, ldvt).  Layout glue for gl_attention's P.V MFMA operand.  d, ldv, ldvt, v_bstride multiples of 8 (16-byte
 * accesses), v and vt 16-byte aligned. */
int gl_transpose_v(const void* v, int64_t v_bstride, int32_t ldv, void* vt, int32_t ldvt,
                   int32_t B, int32_t H, int32_t d, int32_t Nk, void* stream);

/*
 * GroupNorm(32 groups) over NHWC fp16, fp32 statistics (GroupNorm32, util.py:226-228; Normalize,
 * attention.py:78-79), fused SiLU (openaimodel.py:155-157,180-181) and fused channel concat of two
 * sources (openaimodel.py:456): channels [0, C1) come from x1 [B, HW, C1], [C1, C1+C2) from x2.
 *   gl_groupnorm_stats : partial[b][chunk][32][2] = (mean, M2) per group and pixel chunk, fp32, gathered about a
 *                        per-channel shift and merged with the exact pairwise (Welford / Chan) formula
 *   gl_groupnorm_apply : y = (x - mean) * rstd * gamma + beta (, SiLU) -> fp16 [B, HW, C1+C2]
 */
int gl_groupnorm_stats(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t B, int32_t HW,
                       float* partial, int32_t nchunk, void* stream);
int gl_groupnorm_apply(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t B, int32_t HW,
                       const float* partial, int32_t nchunk, const float* gamma, const float* beta,
                       float eps, int32_t silu, void* out, void* stream);
/* gl_groupnorm: the whole operator.  Small maps with C % 256 == 0 (the 16x16 / 8x8 levels) run as ONE launch that keeps a
 * group's slab in registers (two-pass statistics); everything else as gl_groupnorm_stats + gl_groupnorm_apply (partial /
 * nchunk are only used then).  gl_groupnorm_launches tells which (1 or 2). */
int gl_groupnorm(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t B, int32_t HW, const float* gamma,
                 const float* beta, float eps, int32_t silu, void* out, float* partial, int32_t nchunk, void* stream);
int gl_groupnorm_launches(int32_t C, int32_t HW);

/* gl_groupnorm_ex: gl_groupnorm with the inputs optionally in fp32 (the residual stream: GroupNorm32 of openaimodel.py:155,
 * Normalize of attention.py:440 read the block input itself, not an fp16 copy of it) and two optional extra outputs that
 * feed split-fp16 1x1 products:
 *   out_lo : fp16 residual of the normalised rows, out_lo[m * ldo + c] = fp16(y - fp16(y))       (proj_in, attention.py:440)
 *   raw    : the INPUT concat itself as [hi | lo], raw[m * ldraw + c] = fp16(x), raw[m * ldraw + C + c] = fp16(x - fp16(x))
 *            (ResBlock skip_connection reads x, openaimodel.py:190-194,231), ldraw >= 2 (C1 + C2)
 * ldo = row stride of out / out_lo in elements (0: C1 + C2).  partial / nchunk as for gl_groupnorm. */
typedef struct gl_gn_args {
    const void* x1;     int32_t C1;
    const void* x2;     int32_t C2;       /* second source of the channel concat, or NULL */
    int32_t x_f32;                        /* != 0: x1 / x2 are fp32, else fp16 */
    int32_t B, HW;
    const float* gamma; const float* beta; float eps; int32_t silu;
    void* out;          int32_t ldo;
    void* out_lo;
    void* raw;          int32_t ldraw;
    float* partial;     int32_t nchunk;
} gl_gn_args;
int gl_groupnorm_ex(const gl_gn_args* args, void* stream);
int gl_groupnorm_launches_ex(int32_t C, int32_t HW, int32_t x_f32);

/*
 * gl_layernorm: row LayerNorm eps 1e-5 over C (attention.py:216-217,292-294,369-371), fp32 statistics (two-pass).
 * x_f32 is a bit set: bit 0 = the input rows are fp32 (the residual stream) instead of fp16; bit 1 = the OUTPUT rows are fp32
 * instead of fp16 (y, ldy then count floats: the conditioning text encoder's last_hidden_state, encoders/modules.py:167).
 * Input row r = (b, i) with i < rows_in; output row index
 * = b * rows_out + row_off + i -- this writes straight into the [x ; objs] concatenation of GatedSelfAttentionDense
 * (attention.py:230).  stats (optional): (mean, rstd) per input row, fp32 [B * rows_in, 2].  C % 8 == 0, C <= 2048.
 * x2 (optional, fp16, rows2 rows per sample, row stride ldx2): a second source whose rows follow the rows_in rows of x in
 * every sample's output block -- [x ; objs] normalised in one launch.
 */
int gl_layernorm(const void* x, int32_t ldx, int32_t x_f32, void* y, int32_t ldy, const float* gamma, const float* beta,
                 int32_t B, int32_t rows_in, int32_t rows_out, int32_t row_off, int32_t C, float eps, float* stats,
                 const void* x2, int32_t ldx2, int32_t rows2, void* stream);
/* x_f32 bit 2 (value 4): the second source x2 is fp32 (ldx2 in floats) -- the fuser's fuser.linear(objs) rows enter LayerNorm unrounded.
 * gl_layernorm_stats: only the per-row (mean, rstd) of fp32 rows [rows, C] (the same two-pass arithmetic as gl_layernorm), for consumers that
 * re-evaluate the normalisation themselves in fp32 (gl_rela_pool_ln3, gl_rela_merge): rela_fuse's LayerNorm3 output is then never stored. */
int gl_layernorm_stats(const float* x, int32_t ldx, int32_t rows, int32_t C, float eps, float* stats, void* stream);
/* x_f32 bit 3 (value 8; ABI 14): the fp16 output rows are written as [hi | lo] -- y[r * ldy + c] = fp16(v), y[r * ldy + C + c] = fp16(v - fp16(v)),
 * ldy >= 2 C -- the split-fp16 operand of the projection that follows (strict mode).  fp32 input rows only (bit 0; with a second source: bit 2). */
/* gl_split_f32: fp32 rows [rows, C] (row stride ldx floats) -> fp16 [rows, 2 C] rows [hi | lo] (row stride ldy): the split-fp16 form of a
 * residual-stream tensor for a matrix-core consumer that has no producer to write it (down / up convs, the conditioning tensors). C % 8 == 0. */
int gl_split_f32(const float* x, int32_t ldx, int64_t rows, int32_t C, void* y, int32_t ldy, void* stream);

/*
 * RelationCrossAttention (attention.py:315-359) in closed form:
 *   out = hid + (1/max_objs) * sum_i 1[p in rect_i] f_i ,  x_new = (out + x) / 2   (attention.py:398)
 * rects[b][i] = {top, bottom, left, right} int32 pixel bounds (already python-slice-normalised on the
 * host, attention.py:325-346), nvalid[b] = boxes before the first padded/degenerate one, poison[b] != 0
 * reproduces the reference's NaN for an empty (right < left) slice.
 *   gl_rela_pool  : feat[b, i, :] = mean_{p in rect_i} hid[b, p, :]   (0 rows for i >= nvalid[b]); with ln_out != NULL also
 *                   ln_out[b, i, :] = LayerNorm(feat[b, i, :]; ln_gamma, ln_beta, eps 1e-5)  (norm1, attention.py:348)
 *   gl_rela_merge : y = 0.5 * (x + hid + (1/max_objs) * sum_i 1[p in rect_i] f[b, i, :])
 *                   x / y fp32 when x_f32 != 0 (residual stream) else fp16.  hid = LayerNorm3(x) is either read (fp16,
 *                   ln_stats == NULL) or re-evaluated in fp32 from ln_stats = gl_layernorm's (mean, rstd) rows and
 *                   gamma / beta, so that it enters the stream unrounded.  With ln2_out != NULL (fp32 stream + ln_stats
 *                   form only, max_objs <= 32) the same launch also writes ln2_out = LayerNorm(y; ln2_gamma, ln2_beta,
 *                   eps 1e-5) in fp16 -- the norm2 in front of attn2 (attention.py:400) -- bit-identical to gl_layernorm(y).
 * slots (ABI 13): rows per sample of feat / ln_out / f, 0 < slots <= max_objs (0 = max_objs).  The reference runs the relation chain
 * (attention.py:348-351) over all max_objs = 30 rows of every sample although only the first nvalid[b] enter the result; a caller that knows
 * max_b nvalid[b] <= slots computes just `slots` rows per sample -- the rows are independent of one another, so the used ones are unchanged.
 * rects stays [B, max_objs, 4] and 1/max_objs stays the reference's divisor.
 */
int gl_rela_pool(const void* hid, int32_t B, int32_t H, int32_t W, int32_t C, const int32_t* rects,
                 const int32_t* nvalid, const int32_t* poison, int32_t max_objs, int32_t slots, void* feat, const float* ln_gamma,
                 const float* ln_beta, void* ln_out, void* stream);
/* gl_rela_pool_ln3 (round 4): gl_rela_pool on the fp32 stream: x fp32 [B, H*W, C] with the per-row (mean, rstd) of rela_fuse's LayerNorm3
 * (gl_layernorm_stats) and its gamma / beta; the pooled row is gamma * mean_rect((x - mean_r) * rstd_r) + beta in fp32 -- hid = LN3(x)
 * (attention.py:317) is neither rounded to fp16 nor stored. */
int gl_rela_pool_ln3(const float* x, const float* ln3_stats, const float* ln3_gamma, const float* ln3_beta, int32_t B, int32_t H,
                     int32_t W, int32_t C, const int32_t* rects, const int32_t* nvalid, const int32_t* poison, int32_t max_objs,
                     int32_t slots, void* feat, const float* ln_gamma, const float* ln_beta, void* ln_out, void* stream);
int gl_rela_merge(const void* x, int32_t x_f32, const void* hid, const float* ln_stats, const float* gamma,
                  const float* beta, const void* f, int32_t B, int32_t H, int32_t W, int32_t C,
                  const int32_t* rects, const int32_t* nvalid, const int32_t* poison, int32_t max_objs, int32_t slots,
                  void* y, const float* ln2_gamma, const float* ln2_beta, void* ln2_out, void* stream);

/*
 * gl_posnet_input: builds the PositionNet MLP input (text_grounding_net.py:30-41, util.py:12-26):
 *   out[b, i, :] = [ emb * m + (1 - m) * null_pos  |  fourier(box) * m + (1 - m) * null_xyxy ]   fp16
 * boxes [B, n, 4] fp32, masks [B, n] fp32, emb [B, n, in_dim] fp32, nulls fp32.
 */
int gl_posnet_input(const float* boxes, const float* masks, const float* emb, const float* null_pos,
                    const float* null_xyxy, int32_t rows, int32_t in_dim, int32_t num_freqs, void* out,
                    void* stream);

/* gl_timestep_embedding: [cos(t w_k) | sin(t w_k)], w_k = exp(-ln(1e4) k / half) (util.py:161-181) -> fp16 [B, dim] */
int gl_timestep_embedding(const float* t, int32_t B, int32_t dim, void* out, void* stream);
/* fp32-output forms (ABI 14; strict mode: the rows then go through gl_split_f32 and enter the first Linear as [hi | lo]) */
int gl_timestep_embedding_f32(const float* t, int32_t B, int32_t dim, float* out, void* stream);
int gl_posnet_input_f32(const float* boxes, const float* masks, const float* emb, const float* null_pos, const float* null_xyxy,
                        int32_t rows, int32_t in_dim, int32_t num_freqs, float* out, void* stream);

/* gl_silu_f16: y = silu(x) elementwise fp16 (emb_layers SiLU, openaimodel.py:173). n % 8 == 0. */
int gl_silu_f16(const void* x, void* y, int64_t n, void* stream);

/*
 * PLMS step (plms.py:110-163) on fp32 NCHW latents, sigma = 0.
 *   gl_cfg_combine : e = e_u + s * (e_c - e_u)  from the UNet output of the 2B batch [cond ; uncond] (plms.py:123)
 *   gl_plms_update : e' = c0*e + c1*e1 + c2*e2 + c3*e3 (then / div), pred_x0 = (x - s1m*e') / sqrt_at,
 *                    x_prev = sqrt_aprev * pred_x0 + dir_coef * e'                       (plms.py:126-161)
 *                    evaluated in the reference's operation order with FP contraction off.
 *   gl_pack_latent : x fp32 NCHW [B, C, hw] -> fp16 NHWC [reps*B, hw, Cpad] (zero channel padding), the
 *                    first-conv input for both CFG halves.  split != 0 (Cpad >= 3C): channels [0, C) = hi = fp16(x), [C, 2C) = fp16(x - hi),
 *                    [2C, 3C) = hi again -- with first-conv weights packed [Whi | Whi | Wlo] over those channels the conv computes
 *                    xhi.Whi + xlo.Whi + xhi.Wlo in the padding it carries anyway (the engine's weight table always holds that packing;
 *                    gl_set_option 38 = 0 feeds hi alone).
 */
int gl_cfg_combine(const float* eps2b, float guidance, int64_t n, float* e_out, void* stream);
int gl_plms_update(const float* x, const float* e, const float* e1, const float* e2, const float* e3,
                   float c0, float c1, float c2, float c3, float div, float sqrt_at, float s1m,
                   float sqrt_aprev, float dir_coef, int64_t n, float* x_prev, void* stream);
int gl_pack_latent(const float* x, int32_t B, int32_t C, int32_t hw, int32_t Cpad, int32_t reps, int32_t split, void* out,
                   void* stream);

/*
 * VAE decode stage (AutoencoderKL.decode autoencoder.py:40-44, Decoder.forward model.py:535-568).
 * It reuses gl_conv3x3 / gl_gemm / gl_groupnorm_* ; only two extra kernels exist:
 *   gl_latent_affine_pack : z fp32 NCHW -> fp16 NHWC (channel-padded) of post_quant_conv(z / scale_factor)
 *   gl_softmax_rows       : in-place row softmax(scale * x) over fp16 [rows, n]: the d = 512 single-head
 *                           AttnBlock (model.py:150-202) is two GEMMs around it.
 */
int gl_latent_affine_pack(const float* z, const float* w, const float* bias, float pre, int32_t B, int32_t C,
                          int32_t hw, int32_t Cpad, void* out, void* stream);
int gl_softmax_rows(void* x, int32_t rows, int32_t n, int32_t ld, float scale, void* stream);

/* =====================================================================================================
 * Forward-level API ("what a C-ABI replacement must export underneath"): one handle per device per
 * thread owns the block plan, the packed weights' layout, the activation pool, the hoisted conditioning and the
 * captured hipGraphs, so that a host in ANY language can run UNetModel.forward (openaimodel.py:413-459) and a PLMS
 * step (plms.py:110-163) through plain C calls.  All data pointers are DEVICE pointers owned by the caller.
 * ===================================================================================================== */
typedef struct gl_unet_config {          /* UNetModel.__init__ arguments (openaimodel.py:234-262; coco2014.yaml:9-30) */
    int32_t in_channels, model_channels, out_channels, num_res_blocks;
    int32_t n_levels;            int32_t channel_mult[8];
    int32_t n_attn_res;          int32_t attention_resolutions[8];     /* downsample factors that carry a transformer */
    int32_t num_heads, context_dim;
    int32_t pos_in_dim, pos_out_dim, fourier_freqs;                    /* PositionNet (text_grounding_net.py:7-24) */
    int32_t max_objs;                                                  /* 30 (interface.py:158,425) */
    int32_t split_weights;       /* ABI 14.  != 0: EVERY matrix of the weight table is stored [Whi | Wlo] (Whi = fp16(W), Wlo = fp16(W - Whi); Linear
                                    rows [N, 2K], 3x3 convs [Cout, 2, Cin/64, 3, 3, 64]): the layout the strict mode (gl_set_handle_option 50) needs for
                                    its third pass x.Wlo -- with it the engine reproduces the reference's fp32 weights to ~22 bits.  The default
                                    mode of such a handle reads the Whi halves and computes what a handle without the flag computes; the table is
                                    twice the size (5 GB for the GLIGEN UNet).  0 = the compact table (only the 1x1 convs keep [Whi | Wlo]). */
} gl_unet_config;

typedef struct gl_engine gl_engine;      /* opaque */

typedef struct gl_weight_info {          /* one packed tensor of the flat weight buffer */
    char name[160];                      /* e.g. "input_blocks.1.1.transformer_blocks.0.attn1.qkv.w" (weights.py naming) */
    int64_t offset, nbytes;              /* 256-byte aligned slot inside the flat buffer */
    int32_t dtype;                       /* 0 = fp16, 1 = fp32 */
    int32_t ndim;   int64_t shape[4];
} gl_weight_info;

/* gl_create builds the plan and the weight table (no GPU needed); gl_destroy frees pool, graphs and the handle. */
int gl_create(const gl_unet_config* cfg, gl_engine** out);
int gl_destroy(gl_engine* e);
/* The packed-weight layout is defined HERE (the host packer fills it): count, i-th entry, total bytes. */
int gl_num_weights(const gl_engine* e);
int gl_weight_at(const gl_engine* e, int32_t i, gl_weight_info* info);
int64_t gl_weights_bytes(const gl_engine* e);
/* gl_load_weights: `packed` = device buffer laid out per the table (stays owned by the caller and must outlive the
 * handle); has_sd_conv != 0 when the "sd_first_conv.*" slots are filled (GLIGEN/SD_input_conv_weight_bias.pth). */
int gl_load_weights(gl_engine* e, const void* packed, int64_t bytes, int32_t has_sd_conv, void* stream);
/* gl_set_conditioning: everything that depends only on the image's conditioning, hoisted out of the 102 forwards
 * (PositionNet tokens text_grounding_net.py:26-43, fuser.linear attention.py:228, attn2 / rela_fuse K,V projections
 * attention.py:124-125,348-349, integer box rectangles attention.py:321-346).  fp32 inputs: context [Bn, Lc, ctx],
 * relations [Bn, R, ctx], boxes [Bn, max_objs, 4], masks [Bn, max_objs], pos_emb [Bn, max_objs, pos_in_dim]; null
 * grounding = zeros (text_layout_tokinzer_input.py:47-62).  hw = latent side. */
int gl_set_conditioning(gl_engine* e, const float* context, const float* relations, const float* boxes,
                        const float* masks, const float* pos_emb, int32_t Bn, int32_t Lc, int32_t R, int32_t hw,
                        void* stream);
/* gl_unet_forward: eps[Bn, out_ch, hw, hw] fp32 = UNet(x, t | conditioning).  x fp32 NCHW [Bn / reps, in_ch, hw, hw]:
 * with reps = 2 both CFG halves of a [cond ; uncond] conditioning batch share the latent.  t_dev: fp32 [Bn] device
 * timesteps, or NULL to use t_host for every sample.  fuser_scale = what set_alpha_scale wrote (interface.py:34-38;
 * 0 skips the gated self-attention exactly); sd_conv != 0 = restore_first_conv_from_SD is in effect
 * (openaimodel.py:393-405).  Replays a hipGraph captured on first use of each (shape, fuser on/off, conv) variant
 * unless use_graph == 0. */
int gl_unet_forward(gl_engine* e, const float* x, const float* t_dev, float t_host, int32_t reps, float fuser_scale,
                    int32_t sd_conv, float* eps, int32_t use_graph, void* stream);
/* gl_plms_step: one denoiser evaluation + classifier-free guidance + the PLMS / DDIM-sigma-0 update
 * (plms.py:110-163): eps = UNet(x_eval, t); e_out = eps_u + guidance (eps_c - eps_u) (reps == 2) or eps;
 * e' = (sum_j coef[j] * e_terms[j]) / div over n_terms <= 4 terms (Adams-Bashforth forms plms.py:144-159; terms
 * may alias e_out); x_out = sqrt_aprev * (x_base - s1m e') / sqrt_at + dir_coef * e', evaluated in the
 * reference's operation order (bit-identical to torch fp32 given the same eps). */
typedef struct gl_plms_step_args {
    const float* x_eval;  const float* x_base;  float* x_out;       /* fp32 [B, C, hw, hw] */
    float* e_out;                                                    /* guided eps of this evaluation */
    const float* e_terms[4];  float coef[4];  int32_t n_terms;  float div;
    float t;  int32_t reps;  float guidance;  float fuser_scale;  int32_t sd_conv;
    float sqrt_at, s1m, sqrt_aprev, dir_coef;
    int32_t use_graph;
} gl_plms_step_args;
int gl_plms_step(gl_engine* e, const gl_plms_step_args* a, void* stream);
/* introspection for tests / tools */
int64_t gl_pool_bytes(const gl_engine* e);
int gl_num_launches(const gl_engine* e);       /* kernel launches of the last eagerly executed / captured forward */
int gl_sizeof_unet_config(void);
int gl_sizeof_weight_info(void);
int gl_sizeof_plms_step_args(void);

/*
 * gl_reward_score: the GPU part of Reward_Model.forward for the train_rl.py rollout (models/policy.py:106-135,
 * tools/aesthetic.py:15-31,52-57), given CLIP features [B, D] fp32 (D = 768 for ViT-L/14) of the captions, the generated
 * images and the ground-truth images:
 *   sims_ti = <norm(txt), norm(pred)>, sims_ii = <norm(gt), norm(pred)>            (F.normalize, eps 1e-12)
 *   aesthetic = AestheticMLP(normalized(norm(pred)))   five Linear layers D -> 1024 -> 128 -> 64 -> 16 -> 1, fp32 weights
 *   partial_reward (optional) = sims_ti + sims_ii + 0.1 * aesthetic; the caller adds 10 * mIoU + 10 * DocSim (CPU python)
 */
typedef struct gl_reward_args {
    const float* txt; const float* img_pred; const float* img_gt;   /* [B, D] */
    int32_t B, D;
    const float* w1; const float* b1;   /* [1024, D], [1024] */
    const float* w2; const float* b2;   /* [128, 1024] */
    const float* w3; const float* b3;   /* [64, 128]   */
    const float* w4; const float* b4;   /* [16, 64]    */
    const float* w5; const float* b5;   /* [1, 16], [1] */
    float* sims_ti; float* sims_ii; float* aesthetic; float* partial_reward;   /* [B] each */
} gl_reward_args;
int gl_reward_score(const gl_reward_args* a, void* stream);
int gl_sizeof_reward_args(void);

int gl_gemm(const gl_gemm_args* a, void* stream);
int gl_conv3x3(const gl_conv_args* a, void* stream);
int gl_attention(const gl_attn_args* a, void* stream);

/*
 * VAE decode stage behind a forward-level handle: AutoencoderKL.decode (GLIGEN/ldm/models/autoencoder.py:40-44)
 * = Decoder.forward (GLIGEN/ldm/modules/diffusionmodules/model.py:535-568) on z / scale_factor after post_quant_conv.
 * Same contract as the UNet handle: gl_vae_create builds the plan and the LIBRARY-DEFINED flat weight layout (no GPU needed;
 * gl_vae_num_weights / gl_vae_weight_at / gl_vae_weights_bytes; packed forms: 3x3 convs fp16 [Cout, Cin/64, 3, 3, 64] with the
 * latent's channels zero-padded to 64, 1x1 convs fp16 [N, K] with C^-0.5 folded into mid.attn_1.q, norm affine / biases /
 * post_quant_conv fp32), gl_vae_load_weights references a DEVICE buffer in that layout (the caller keeps it alive),
 * gl_vae_decode runs z fp32 [B, z_channels, side, side] -> out fp32 [B, out_ch, side * 2^(n_mult-1), ...] (device pointers),
 * as one hipGraph per (B, side) when use_graph != 0.  One handle per device per thread.
 */
typedef struct gl_vae_config {
    int32_t ch;                  /* ddconfig.ch (128) */
    int32_t ch_mult[8];          /* ddconfig.ch_mult (1, 2, 4, 4) */
    int32_t n_mult;              /* len(ch_mult) */
    int32_t num_res_blocks;      /* ddconfig.num_res_blocks (2) */
    int32_t z_channels;          /* 4 */
    int32_t out_ch;              /* 3 */
    int32_t embed_dim;           /* 4 (post_quant_conv input channels) */
    float scale_factor;          /* 0.18215 */
} gl_vae_config;
typedef struct gl_vae gl_vae;
int gl_vae_create(const gl_vae_config* cfg, gl_vae** out);
int gl_vae_destroy(gl_vae* v);
int gl_vae_num_weights(const gl_vae* v);
int gl_vae_weight_at(const gl_vae* v, int32_t i, gl_weight_info* info);
int64_t gl_vae_weights_bytes(const gl_vae* v);
int gl_vae_load_weights(gl_vae* v, const void* packed, int64_t bytes, void* stream);
int gl_vae_decode(gl_vae* v, const float* z, int32_t B, int32_t side, float* out, int32_t use_graph, void* stream);
int gl_vae_num_launches(const gl_vae* v);
int64_t gl_vae_pool_bytes(const gl_vae* v);
int gl_sizeof_vae_config(void);

/*
 * CLIP towers of the reward stage: transformers.CLIPModel.get_image_features / get_text_features as the
 * reference's Reward_Model calls them (models/policy.py:106-113).  Projections / MLPs / LayerNorms / the vision tower's
 * attention run on gl_gemm / gl_layernorm / gl_attention; these are the tower-specific pieces.
 *   gl_clip_patchify      CLIPVisionEmbeddings.patch_embedding as a GEMM operand: pixel_values fp32 [B, 3, S, S] -> fp16
 *                         [B * (S/patch)^2, Kpad], K index (c * patch + i) * patch + j, zero-padded to Kpad (% 64 == 0)
 *   gl_clip_assemble      x[b, 0] = class_embedding, x[b, 1 + p] = patch_emb[b * (T-1) + p]; x += position_embedding; then, when
 *                         ln_gamma != NULL, x = LayerNorm(x) in fp32 (the vision tower's pre_layrnorm, whose OUTPUT is the start of
 *                         the fp32 residual stream); fp32 [B, T, C], C <= 4096
 *   gl_clip_embed_tokens  CLIPTextEmbeddings: x[b, t] = token_embedding[ids[b, t]] + position_embedding[t]; fp32 [B, T, C]
 *   gl_clip_gather_rows   out[b] = x[rows[b]]   (class-token rows / EOS rows = input_ids.argmax(-1)), fp32
 *   gl_attention_small    softmax(scale * q k^T [+ causal mask]) v for T <= 128, d <= 64; q / k / v fp16 rows of stride ld with head
 *                         h at column h * d (e.g. the three column blocks of a fused q|k|v projection), fp32 arithmetic;
 *                         the text tower's causal attention (create_causal_mask, CLIPTextModel.forward)
 */
int gl_clip_patchify(const float* pixel_values, int32_t B, int32_t S, int32_t patch, int32_t Kpad, void* out, void* stream);
int gl_clip_assemble(const void* patch_emb, int32_t ldpe, const float* class_emb, const float* pos_emb, int32_t B, int32_t T,
                     int32_t C, const float* ln_gamma, const float* ln_beta, float ln_eps, float* x, void* stream);
int gl_clip_embed_tokens(const int32_t* ids, const float* tok_emb, const float* pos_emb, int32_t B, int32_t T, int32_t C,
                         int32_t vocab, float* x, void* stream);
int gl_clip_gather_rows(const float* x, int32_t ldx, const int32_t* rows, int32_t B, int32_t C, float* out, void* stream);
int gl_attention_small(const void* q, const void* k, const void* v, int32_t ld, int32_t B, int32_t T, int32_t H, int32_t d,
                       float scale, int32_t causal, void* out, int32_t ldo, void* stream);

/* ---- image preprocessing of the reward stage (models/policy.py:108-111: self.processor(images=...), the HuggingFace CLIP
 * feature extractor on PIL images; transformers 4.19.2 image_utils + Pillow) -- on the GPU, so rollout images stay in HBM
 * between the VAE decoder and the CLIP vision tower.
 *   gl_image_to_u8      decoded image fp32 [B, 3, H, W] in [-1, 1] -> uint8 [B, H, W, 3] with the arithmetic of
 *                       GLIGEN/interface.py:543-547 (clamp, * 0.5 + 0.5, * 255, truncation): the pixels of the PIL image there
 *   gl_resample_h_u8    Pillow's horizontal BICUBIC pass (src/libImaging/Resample.c ImagingResampleHorizontal_8bpc), bit-exact:
 *                       out[b, y, xx, c] = clip8((2^21 + sum_{k < bounds[xx][1]} in[b, y, bounds[xx][0] + k, c] * coeffs[xx][k]) >> 22);
 *                       bounds int32 [Wout, 2] and coeffs int32 [Wout, ksize] are DEVICE arrays built by the host
 *                       (layoutllm_t2i_amd/preprocess.py: Pillow's precompute_coeffs + normalize_coeffs_8bpc in float64)
 *   gl_resample_v_norm  Pillow's vertical pass on that 8-bit intermediate, restricted to the centre-crop window
 *                       [top, top + crop_h) x [left, left + crop_w) of the resized image, followed by the feature extractor's
 *                       float32(u8) / 255 -> (x - mean) / std -> channels first: fp32 [B, 3, crop_h, crop_w] (out_nchw) and / or the
 *                       cropped uint8 pixels [B, crop_h, crop_w, 3] (out_u8_hwc, for tests); mean3 / std3 are HOST pointers (3 floats). */
int gl_image_to_u8(const float* img_nchw, int32_t B, int32_t H, int32_t W, uint8_t* out_hwc, void* stream);
int gl_resample_h_u8(const uint8_t* in, int32_t B, int32_t H, int32_t W, const int32_t* bounds, const int32_t* coeffs, int32_t ksize,
                     int32_t Wout, uint8_t* out, void* stream);
int gl_resample_v_norm(const uint8_t* in, int32_t B, int32_t H, int32_t W, const int32_t* bounds, const int32_t* coeffs, int32_t ksize,
                       int32_t Hout, int32_t top, int32_t left, int32_t crop_h, int32_t crop_w, const float* mean3, const float* std3,
                       float* out_nchw, uint8_t* out_u8_hwc, void* stream);

/* introspection: ABI version and struct sizes (checked by the host loader) */
int gl_abi_version(void);
int gl_sizeof_gemm_args(void);
int gl_sizeof_conv_args(void);
int gl_sizeof_attn_args(void);
int gl_sizeof_gn_args(void);
/* tuning knobs for A/B measurements (results do not depend on them beyond fp32 summation order in split-K):
 * key 2 = GEMM tile shape policy (0 auto, 1 force 128x128, 2 prefer 128x160); key 3 = attention block shape (0 auto,
 * 3 always 8 waves, 4 always 4 waves); keys 4-7 = small-tile / split-K / 256-row-tile thresholds; key 8 = short-K GEGLU
 * GEMMs on the BK 32 / 4-blocks-per-CU variant (1 default, 0 off); key 10 = s_setprio around the attention MFMA
 * clusters (-1 auto, 0 off, 1 on); key 13 = intra-block K-split GEMM/conv variants (0 off, 1 auto = default, 2 always);
 * key 16 = GroupNorm apply pixels per block; key 17 = single-launch GroupNorm kernels (1 default: small maps + group bundles up to 80 KB per block, 2 = small maps only, 0 off); key 20 = (tests) execute the gated-SA fuser even at fuser_scale 0;
 * key 21 = V^T written by the QKV GEMM epilogue (1, default) or by gl_transpose_v (0); key 23 = output-tile order (0 N-tiles
 * fastest, 1 = default: M-tiles fastest when the weight matrix is the larger operand, so each XCD's L2 streams only its
 * slice of the weights, 2 always M-fastest); key 24 = skinny-GEMM kernel (M <= 1024 rows, register operands, four waves split
 * K) while its operand re-reads stay below this many MiB (64 default, 0 = LDS-staged kernels only); key 25 = gl_rela_merge
 * also writes the following LayerNorm (1, default) or a separate gl_layernorm launch does (0); key 27 = fused FeedForward where applicable (1, default) or never (0); key 29 =
 * attention keeps the running max in the padding column of Q / K where the head dim leaves one (d % 16 == 8; 1 default, 0 off);
 * key 30 = 8-wave deep-pipelined 256-row GEMM / conv kernel (0 off, 1 default: problems with at least key-31 (200) such tiles,
 * 2 wherever it applies, with split-K); key 32 = (measurement) its timestamping instantiation, see gl_debug_read; keys 33-35, 37 = its K order
 * / split-K slice length / minimum K / use on short-K multi-round grids that fill >= 80 % of their rounds (1 default).
 * Keys that DO change results (precision): key 41 = the engine's 1x1 convs (ResBlock skip_connection, proj_in, proj_out)
 * take split-fp16 activations ([hi | lo] against the same weight) and every GroupNorm that reads a residual-stream tensor reads it
 * in fp32 (1 default; 0 = round-3 behaviour: fp16 copies); key 42 = the ResBlock's first conv writes its output in fp32 for the
 * following GroupNorm (1 default, 0 = fp16).
 * key 44 = a [cond ; uncond] forward (reps == 2, one timestep for the batch) computes everything that precedes the first
 * conditioning-dependent op -- conv_in, the first ResBlock, proj_in .. attn1 of the first transformer block -- ONCE on the shared
 * latents and duplicates it for the uncond half (1 default; 0 = both halves computed; results equal up to the tile / split-K choice
 * of the half-sized launches).
 * key 38 = the first conv takes the latent as [hi | lo | hi] channels against [Whi | Whi | Wlo] weights (1 default; 0 = fp16(x) against Whi alone).
 * key 43 = the relation chain of rela_fuse (attention.py:348-351) runs on max_b nvalid[b] rows per sample (rounded up to 8) instead of all
 * max_objs = 30 (1 default; read when gl_set_conditioning runs, which then copies the nvalid counts back to the host once); the used rows
 * are unchanged up to the tile / split-K choice of GEMMs with fewer rows.
 * key 45 (with key 41) = the three kinds of 1x1 conv, whose weight rows are stored [Whi | Wlo] (Whi = fp16(W), Wlo = fp16(W - Whi)), add the
 * third pass xhi.Wlo to xhi.Whi + xlo.Whi for launches of more than this many rows (default 1024, the minimum; 0 = the fp16 weight alone): the rounding of those
 * weights is 57 % of the error of storing the UNet's weights in fp16.
 * key 46 = half-height (128-row) tiles of the 8-wave GEMM / conv kernel where the 256-row grid would cover at most half the chip (32x32
 * maps and below at 2B = 8): bit 0 = convs whose 256-row plan leaves <= 16 K-tiles per split-K slice, bit 1 = plain GEMMs, bit 2 = every
 * conv (A/B), bit 3 = multi-round plain GEMMs whose 256-row grid ends in a mostly empty round while the 128-row grid fills its rounds;
 * default 11, 0 = 256-row tiles only.  Results differ from the 256-row plan only through the number of split-K slices.
 * key 47 = plain GEMMs use the 8-wave kernel from this many blocks (tiles x K slices) on (default 100; the split-K decision keeps key 31).
 * key 50 = STRICT mode (0 default; handles created with gl_unet_config.split_weights only, GL_ERR_BAD_ARG otherwise): every matrix product of
 * the forward takes split-fp16 operands -- activations as [hi | lo] (LayerNorm / GroupNorm / GEGLU / attention / projection epilogues write
 * both halves, gl_split_f32 for stream tensors entering a down / up conv), 3x3 convs with gl_conv_args.in_split, attention with
 * gl_attn_args.q_lo / k_lo / vt_lo (three passes for Q.K^T and P.V, P split in registers) -- so that the output is within north_star's
 * rtol 1e-3 / atol 1e-4 of the fp32 reference for > 99 % of the elements (measured 0.1-0.2 % outside, rel-L2 < 1e-4, at 2-3x the
 * default mode's time).  The relation chain of rela_fuse (1/30 weight, 3e-6 of the result) and the fused first conv keep their forms.
 * key 51 = strict mode's third pass x.Wlo (1 default; 0 = activations split only: exact for fp16-representable weights except the
 * folded softmax scale of the q projections).  The strict mode's conditioning hoists are computed lazily: by gl_set_conditioning when key 50
 * is set at that time, else by the first strict forward after it (and again after key 51 changed) -- a split_weights handle that stays in
 * default mode never runs them.
 * key 52 = three-pass split-fp16 products xhi.Whi + xlo.Whi + xhi.Wlo (gl_gemm with K = 3 * kwrap whose second source is the first one again;
 * gl_conv3x3 with in_split = 3) run the DEDICATED three-pass main loop of the 8-wave kernel (1 default; csrc/gemm8.hip S3: a ring stage holds
 * one 32-wide k slice of {xhi, xlo, Whi, Wlo} and feeds three MFMA groups, so no operand is staged twice); 0 = the K-walk over
 * [xhi | xlo | xhi] x [Whi | Whi | Wlo].  Same products, other fp32 summation order. */
int gl_set_option(int key, int value);
/* gl_set_option writes the PROCESS defaults (op-level calls and every handle without an override see them).  A handle can
 * override individual keys for itself: while one of ITS entry points (gl_set_conditioning / gl_unet_forward / gl_plms_step,
 * gl_vae_decode) runs, lookups on that thread see the defaults with the handle's overrides applied, so two hosts in one
 * process tune independently.  A change drops only that handle's captured graphs.  Same keys and value normalisation as
 * gl_set_option; -1 for an unknown key. */
int gl_set_handle_option(gl_engine* e, int key, int value);
int gl_clear_handle_options(gl_engine* e);
int gl_vae_set_option(gl_vae* v, int key, int value);
/* measurement hook (tools/g8_probe.py): what = 8 copies the per-block cycle stamps [entry, prologue done, main loop done,
 * epilogue done] (4 x uint64 per block, up to 4096 blocks) that the timestamping instantiation of the 8-wave GEMM / conv
 * kernel writes while gl_set_option(32, 1) is in effect (synchronous device-to-host copy); what = 9 copies one uint64: the
 * number of gl_gemm / gl_conv3x3 calls this process has served with the 8-wave kernel (the tests of that kernel assert it moved). */
int gl_debug_read(int what, void* dst, int64_t bytes);
/* one-time per-process setup (raises dynamic-LDS limits of the tiled kernels); idempotent */
int gl_init(void);

#ifdef __cplusplus
}
#endif
#endif /* GLIGEN_HIP_H */
