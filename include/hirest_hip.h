/*
 * hirest_hip.h — C ABI of libhirest_hip.so, the MI355X (gfx950) implementation of HiREST's
 * frame/text encoding + cross-modal scoring hot path.
 *
 * The reference has no FFI of its own: its boundary is the duck-typed Python surface
 *   EVA_CLIP.encode_image / encode_text / forward      (EVA_clip/eva_model.py:317-334)
 *   build_eva_model_and_transforms                      (EVA_clip/eva_clip.py:155-172)
 *   mean-pool + L2 + text @ video.T                     (inference_video_retrieval.py:283-285,323-334)
 *   sort-by-score ranking, R@k                          (evaluate.py:58-69)
 * and every "kernel" underneath is a stock ATen op.  This header is the seam a
 * maintainer binds instead (ctypes stub in INTEGRATION.md): plain device pointers and
 * sizes, no torch types.  Conventions for every entry point:
 *   - all pointers are DEVICE pointers unless the name says host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing
 *     synchronises, nothing allocates device memory, there is no hidden global state
 *     (besides the test / timing switches hirest_*_select_kernel, hirest_*_debug_mode — process-wide atomics — and the optional profiler);
 *   - return value: 0 on success, a negative HIREST_E_* for argument errors, or a
 *     positive hipError_t from the launch;
 *   - bf16 tensors are raw uint16 bit patterns (round-to-nearest-even from fp32);
 *   - matrices are row-major; "ld*" are leading dimensions in ELEMENTS.
 */
#ifndef HIREST_HIP_H
#define HIREST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIREST_ABI_VERSION 4   /* 3: hirest_gemm_args gained struct_size (first member) and flags; 4 (round 6): hirest_split2_both_bf16 and
                                 * HIREST_GEMM_KBLOCKED are gone, hirest_gemm_bf16 rejects unknown flag bits, hirest_attention_set_mapping
                                 * takes 0..2 (default 2), the joint model gained its split-operand (bf16x3) precision */

#define HIREST_E_BADARG   (-1)
#define HIREST_E_SHAPE    (-2)   /* unsupported shape (see each function) */
#define HIREST_E_WORKSPACE (-3)  /* workspace too small */

typedef uint16_t hirest_bf16;

int hirest_abi_version(void);
/* human-readable build string: arch, compiler, kernel variants */
const char* hirest_build_info(void);

/* ------------------------------------------------------------------------------------
 * GEMM with fused epilogues:  acc[m][n] = sum_k A[m][k] * W[n][k]   (bf16 in, fp32 accumulate
 * on MFMA).  W is a torch nn.Linear weight as stored ([out_features, in_features]).
 * Replaces F.linear / nn.Linear / `x @ proj` on the path (vit_model.py:56-62,124-127,148;
 * eva_model.py:137-142,249; nn.MultiheadAttention in/out projections).
 * Requirements: K % 64 == 0, N % 4 == 0, lda/ldw multiples of 8, pointers 16-B aligned.
 * ------------------------------------------------------------------------------------ */
enum hirest_epilogue {
    HIREST_EPI_BIAS_BF16 = 0,        /* out bf16 [M,ldo]  = acc + bias                        */
    HIREST_EPI_BIAS_GELU_BF16 = 1,   /* out bf16          = gelu_erf(acc + bias)  (nn.GELU()) */
    HIREST_EPI_BIAS_QGELU_BF16 = 2,  /* out bf16          = quick_gelu(acc + bias) (model.py:175) */
    HIREST_EPI_BIAS_RESID_F32 = 3,   /* out f32 (in/out)  = out + acc + bias     (x += f(x)) */
    HIREST_EPI_BIAS_F32 = 4,         /* out f32           = acc + bias                        */
    HIREST_EPI_PATCH_POS_F32 = 5,    /* patch-embed: row m=b*P+p -> out row b*(P+1)+1+p;
                                        out f32 = acc + bias + pos[(1+p)*ldo' ...]; see below */
    /* LayerNorm folded into the neighbouring GEMMs (large problems only: M*N >= 2^21, M >= 512, N >= 256;
     * HIREST_E_SHAPE otherwise).  LN(x) W^T + b = rstd * (x W'^T - mean * s) + b' with W' = W * gamma (per input
     * column), s[n] = sum_k W'[n][k], b' = b + W beta, so the GEMM reads the un-normalised stream and the LayerNorm
     * pass (vit_model.py:177-178) disappears: */
    HIREST_EPI_BIAS_RESID_LNSTATS_F32 = 6, /* producer: as BIAS_RESID_F32, plus aux0 = bf16 copy [M,N] of the new rows
                                        and aux1 = f32 [M, ceil(N/64), 2] per-row (sum, sum of squares) of each 64-column
                                        group of the bf16-ROUNDED values (N % 8 == 0); hirest_ln_stats_finalize turns them into
                                        (mean, rstd) rows */
    HIREST_EPI_LNFOLD_BF16 = 7,      /* consumer: A = that bf16 copy, W = W', bias = b', aux0 = f32 [M,2] (mean, rstd; the buffer
                                        must be readable up to an EVEN number of rows, i.e. M + 1 rows when M is odd),
                                        aux1 = s [N];  out bf16 = rstd * (acc - mean * s) + b' */
    HIREST_EPI_LNFOLD_GELU_BF16 = 8, /* same, then gelu_erf */
    HIREST_EPI_BIAS_GELU_SPLIT2 = 9, /* HIREST_GEMM_X3 only: out bf16 [M, 2N] = nn.GELU()(acc + bias) (erf form in fp32 by the bf16 towers' degree-8
                                        exp2 polynomial, csrc/common.h gelu_erf2: max abs error 1.1e-6 against exact erff, which the separate
                                        hirest_split2_bf16(act = 1) pass uses — the two routes are NOT bit-equal) in the split operand format
                                        of hirest_split2_bf16 (per 64 columns: hi of 32 outputs | their lo): fc1 of the bf16x3 tower feeding fc2
                                        without an fp32 round trip.  N % 32 == 0, ldo >= 2N. */
    HIREST_EPI_BIAS_RESID2_LNSTATS = 10 /* producer like HIREST_EPI_BIAS_RESID_LNSTATS_F32 with the residual stream kept as two bf16 arrays:
                                        aux0 = hi [M, N] = bf16(x) (in / out; the next GEMM's A operand), out = lo [M, ldo] = bf16(x - hi)
                                        (in / out): x' = hi + lo + acc + bias in fp32, hi' = bf16(x'), lo' = bf16(x' - hi'); aux1 = row
                                        partials of hi' as before.  16 significand bits per residual value, 8 instead of 10 bytes of
                                        epilogue traffic per element (the bf16 vision tower's blocks, calls of >= 64 frames). */
};

typedef struct hirest_gemm_args {
    uint64_t struct_size;                 /* = sizeof(hirest_gemm_args) of the header the caller was built against; a
                                             binding with another layout is rejected (HIREST_E_BADARG) instead of being
                                             read past its end */
    const hirest_bf16* A;  int64_t lda;   /* [M,K] */
    const hirest_bf16* W;  int64_t ldw;   /* [N,K] */
    const float* bias;                    /* [N] or NULL */
    void* out;             int64_t ldo;   /* [M,N] bf16 or f32 by epilogue */
    int32_t M, N, K;
    int32_t epilogue;                     /* enum hirest_epilogue */
    /* HIREST_EPI_PATCH_POS_F32 only: */
    const float* pos;                     /* [(P+1), N] position embedding (row 0 = cls slot) */
    int32_t patches_per_frame;            /* P (256 for 224^2 / 14) */
    /* LN-fold epilogues only (see enum): */
    void* aux0;
    void* aux1;
    int32_t flags;                        /* HIREST_GEMM_REVERSE: the persistent kernels walk their tile list backwards.  Same
                                             results; a kernel that starts where its producer stopped finds the producer's
                                             last ~256 MB in the Infinity Cache (the tower runs fc2 backwards: it reads fc1's
                                             output, and the next qkv reads fc2's) */
} hirest_gemm_args;
enum { HIREST_GEMM_REVERSE = 1,
       HIREST_GEMM_X3 = 2,     /* "bf16x3": A and W are fp32 matrices split by hirest_split2_bf16 ([rows, 2K] bf16, each block of 64 columns
                                * = hi parts of 32 consecutive k | their lo parts), K = 2 x the real depth; the kernel adds W_hi A_lo + W_lo A_hi
                                * + W_hi A_hi per 32 k in fp32 (products carry ~16 mantissa bits at 3 bf16 MFMAs each).  Epilogues
                                * HIREST_EPI_BIAS_F32 / HIREST_EPI_BIAS_RESID_F32 (+ GELU_SPLIT2); the persistent ping-pong kernel, or — for
                                * problems of fewer than 256 tiles of 256 x 256 — an 8-wave 128 x 128 kernel (gemm_t128x3). */
       HIREST_GEMM_X3_T128 = 4 /* with HIREST_GEMM_X3: the 128 x 128 kernel at every size and for HIREST_EPI_BIAS_GELU_SPLIT2 as well — the joint
                                * model's products (1 500 ... 10 000 rows, 768 ... 3 072 wide) are a few hundred such tiles, which fill the 256
                                * CUs better than 256 x 256 tiles do (hirest_joint_encoder_x3_forward).  With HIREST_EPI_BIAS_RESID_F32, fewer
                                * than 32 row panels and aux0 != NULL (aux1 NULL), aux0 is fp32 scratch of >= 4 M N floats and the K range of a
                                * tile may be cut into up to 4 slices that run on different CUs (partial tiles added in slice order). */
     };

int hirest_gemm_bf16(const hirest_gemm_args* args, void* stream);
/* Kernel selection for tests / A-B timing: 0 = automatic (default), 1 = force the 128x128 kernel,
 * 2 / 3 = the 256x256 ping-pong kernel with a 4- / 5-slot LDS ring, 4 = t256p (32-deep slabs), 5 = t256q
 * (64-deep steps), 6 / 7 = the persistent 256x256 kernel with 8 / 4 waves, 8 = the persistent ping-pong kernel (default for
 * K >= 4096), 9 = that kernel with two phases of 32 MFMAs per 64-deep step instead of four of 16 (pq256; also selects the two-phase form of the
 * HIREST_GEMM_X3 kernel).  10..17 (the 4-wave kernel of round 2) and 18..20 (the two-workgroup kernel gemm_d2 of round 3) are retired and rejected.
 * Results are identical for every valid selection (same k order per output element). */
int hirest_gemm_select_kernel(int32_t which);
/* TIMING EXPERIMENTS ONLY (results become wrong): bit0 = skip the main-loop LDS-DMA, bit1 = skip the
 * main-loop barrier and waits of the t256p kernel, bit2 = the persistent kernel streams tile (0,0)'s operands for
 * every tile (L2-resident operands).  Bits 3-5 only reorder the persistent kernel's tile walk (results stay correct):
 * bit3 force the grouped order, bit4 force panel-major, bit5 pair ragged edge tiles into equal-duration units.
 * Persistent kernel, results wrong: bit6 no wait for the LDS-DMA, bit7 DMA of the A operand only, bit8 half the fragment
 * reads.  Any of these selects a separate (slower-scheduled) instantiation.  bit9 (512, results unchanged, normal kernels):
 * ignore HIREST_GEMM_REVERSE.  0 restores normal operation. */
int hirest_gemm_debug_mode(int32_t bits);
/* Name of the kernel instantiation hirest_gemm_bf16 would launch for `args` under the current selection state, as rocprofv3
 * prints it (e.g. "gemm_p256<8, 64, false>", "gemm_pp256<6>").  Host logic only; out_len >= 48.  Lets a committed profile be
 * checked against what the library dispatches today. */
int hirest_gemm_dispatch_name(const hirest_gemm_args* args, char* out, int32_t out_len);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the last dim (biased variance, fp32 statistics), fp32 in -> bf16 or f32 out.
 * Replaces nn.LayerNorm / LayerNorm subclasses (vit_model.py:159,165,285; eva_model.py:19-25;
 * model.py:166-172).  Row i of the input is x + row_index[i]*ldx (row_index NULL -> i).
 * D % 4 == 0, D <= 8192.
 * ------------------------------------------------------------------------------------ */
int hirest_layernorm(const float* x, int64_t ldx, const int32_t* row_index,
                     const float* gamma, const float* beta, float eps,
                     void* out, int64_t ldo, int32_t out_is_f32,
                     int32_t rows, int32_t D, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-head self-attention core on a packed QKV activation (the output of the QKV GEMM):
 *   qkv bf16 [B*N, 3*H*dh] with columns ordered (which in {q,k,v}, head, dh)
 *   out bf16 [B*N, H*dh]   = softmax(scale * q k^T [+ causal mask]) v,  heads concatenated
 * Replaces vit_model.py:127-147 (no mask) and nn.MultiheadAttention's core with the additive
 * -inf causal mask of eva_model.py:224-230.  fp32 softmax, bf16 P.V on MFMA.
 * Supported: N <= 272, dh in {64, 88} (dh 88 is zero-padded to 96 on chip).
 * ------------------------------------------------------------------------------------ */
int hirest_attention_bf16(const hirest_bf16* qkv, hirest_bf16* out,
                          int32_t B, int32_t N, int32_t H, int32_t dh,
                          float scale, int32_t causal, void* stream);
/* The same, but only the output rows of queries [0, q_rows) of every sequence are guaranteed to be written (the
 * persistent kernel computes the leading ceil(q_rows / 16) query tiles and still streams all keys / values; the other
 * kernels compute everything).  Rows that are written are bit-identical to hirest_attention_bf16's.  The last
 * block of the vision tower uses q_rows = 1: vit_model.py:340-351 reads only x[:, 0] after it. */
int hirest_attention_bf16_rows(const hirest_bf16* qkv, hirest_bf16* out,
                               int32_t B, int32_t N, int32_t H, int32_t dh,
                               float scale, int32_t causal, int32_t q_rows, void* stream);
/* 1 = register-staged kernel with a transposed V image, 2 = LDS-DMA staging + hardware transpose reads, one workgroup
 * per (frame, head), 3 = 2's arithmetic in one persistent workgroup per frame (nine waves; used for 80 < N <= 272 tokens
 * and >= 64 frames; other shapes fall back to 2), 4 = 3 with twelve waves, 7 (default) = 3 with a tenth wave that issues all
 * LDS-DMA (same bits as 3), 5 = 3 with the softmax on fewer VALU instructions (row sums taken from the P.V product through a
 * column of ones in V's pad; last-bit differences from 3), 6 = 5 + the producer wave.  For tests / A-B timing. */
int hirest_attention_select_kernel(int32_t which);
/* Persistent kernel, which (frame, head) pairs a workgroup walks: 0 = one frame per workgroup, its heads in order; 1 = one head per
 * workgroup over frames b0, b0 + 16, ..., the 16 heads of two frames running on the 32 CUs of one XCD at the same time (used when
 * 32 % H == 0 and the batch is large enough; anything else takes mapping 0); 2 (default) = automatic: mapping 1 below 256 frames, where
 * one workgroup per frame leaves CUs idle (64 frames: 0.131 -> 0.045 ms per launch), mapping 0 from 256 frames on.  Per (frame, head)
 * the arithmetic is the same: results are bit-identical. */
int hirest_attention_set_mapping(int32_t by_head);
/* Persistent kernel with a producer wave: ~64 * units cycles between two K pieces of the next step's K image (0 = one burst).
 * Results are unchanged. */
int hirest_attention_set_pace(int32_t units);
/* Persistent kernel: the first workgroup of CU c starts ~64 * units * (c mod 16) cycles late, so that the CUs' per-step memory bursts
 * do not coincide.  0 = off.  Results are unchanged. */
int hirest_attention_set_stagger(int32_t units);
/* Persistent kernel: park waves 4-7 (the second wave of each SIMD) for ~64 * units cycles after the per-head barrier, so that
 * their MFMA phases fall under the first wave's softmax (VALU) phase and vice versa.  0 = off.  Results are unchanged. */
int hirest_attention_set_skew(int32_t units);
/* TIMING EXPERIMENTS ONLY (results become wrong), persistent kernel: bit0 skip the S^T MFMAs, bit1 skip the softmax
 * exponentials, bit2 skip P.V, bit3 skip the K/V LDS-DMA, bit4 skip the output stores, bit5 skip the Q loads, bit6 / bit7 skip
 * the per-head barriers B / A.  0 restores normal operation. */
int hirest_attention_debug_mode(int32_t bits);
/* Debug instantiation with bit8 of the mode set: workgroup 0 stamps the shader clock at 11 points of every step
 * ([step < 64][wave < 12][slot < 12] int64; slots: 10 top of the step, 0 after barrier A, 1 / 5 tile start, 2 / 6 S^T issued, 3 / 7 softmax
 * done, 4 / 8 V ready, 9 tiles done).  Copies the first n stamps to host memory (tools/attn_trace.py prints the phase table). */
int hirest_attention_debug_trace_read(int64_t* dst, int32_t n);

/* ------------------------------------------------------------------------------------
 * Patch extraction (im2col for Conv2d with kernel == stride == P, vit_model.py:198,205):
 * frames [B,3,S,S] -> patches bf16 [B*(S/P)^2, Kpad], column order (c, ph, pw) = the conv
 * weight's flatten order; columns >= 3*P*P are zero.  in_dtype: 0 = f32 NCHW (already
 * normalised), 1 = bf16 NCHW, 2 = uint8 NHWC raw RGB with fused (x/255-mean)/std
 * (eva_clip.py:16-17,125-153 ToTensor+Normalize).
 * ------------------------------------------------------------------------------------ */
int hirest_patchify(const void* frames, int32_t in_dtype, int32_t B, int32_t S, int32_t P,
                    const float* mean3, const float* std3,
                    hirest_bf16* patches, int32_t Kpad, void* stream);

/* Weight preparation of the folded LayerNorm (HIREST_EPI_LNFOLD_*), once per checkpoint:
 * Wf = bf16(W * gamma) [N, K], colsum_out[n] = sum_k Wf[n][k] (of the rounded values), bias_out = bias + W beta (bias NULL = 0). */
int hirest_fold_layernorm(const float* W, const float* gamma, const float* beta, const float* bias, hirest_bf16* Wf,
                          float* bias_out, float* colsum_out, int32_t N, int32_t K, void* stream);
/* x[b*(P+1)] = cls + pos[0]  for every frame (vit_model.py:330-333, the CLS row). */
int hirest_write_cls_rows(float* x, int64_t ldx, const float* cls, const float* pos0,
                          int32_t B, int32_t tokens_per_frame, int32_t D, void* stream);

/* Statistics side of the folded LayerNorm (HIREST_EPI_LNFOLD_*; vit_model.py:177-178):
 *   hirest_rowstats_bf16       x f32 [rows, D] -> xb bf16 [rows, D] (the GEMM's A operand) and stats f32 [rows, 2] =
 *                              (mean, 1/sqrt(var + eps)) of the ROUNDED row, biased variance.  D % 4 == 0, D <= 1536.
 *   hirest_ln_stats_finalize   partials f32 [rows, groups, 2] written by HIREST_EPI_BIAS_RESID_LNSTATS_F32 -> stats. */
/* `guard` (device float, may be NULL): both kernels raise it to max over their rows of |mean| * rstd — how many sigmas the
 * worst row sits away from zero.  The fold hands the GEMM the UN-normalised bf16 row, so a row offset of r sigma costs
 * about r times the rounding noise of the LayerNorm pass; callers zero it before a tower call and compare afterwards. */
int hirest_rowstats_bf16(const float* x, int64_t ldx, hirest_bf16* xb, float* stats, float eps, int32_t rows, int32_t D,
                         float* guard, void* stream);
/* The same with the residual's low part as well: xlo [rows, D] = bf16(x - xb) (the two-array residual stream of
 * HIREST_EPI_BIAS_RESID2_LNSTATS); hirest_combine_hi_lo_f32 turns rows of it back into fp32 (out = hi + lo). */
int hirest_rowstats_split_bf16(const float* x, int64_t ldx, hirest_bf16* xb, hirest_bf16* xlo, float* stats, float eps, int32_t rows,
                               int32_t D, float* guard, void* stream);
int hirest_combine_hi_lo_f32(const hirest_bf16* hi, const hirest_bf16* lo, int64_t ld_in, float* out, int64_t ldo, int32_t rows,
                             int32_t D, void* stream);
int hirest_ln_stats_finalize(const float* partials, int32_t groups, float* stats, float eps, int32_t rows, int32_t D,
                             float* guard, void* stream);

/* Text prologue: x[b,t,:] = tok_emb[tok[b,t]] + pos[t]  (eva_model.py:233-235); also writes
 * eot_row[b] = b*L + argmax_t tok[b,t] (first maximum) for the EOT gather (eva_model.py:243). */
int hirest_embed_tokens(const int64_t* tokens, const float* tok_emb, const float* pos,
                        float* x, int32_t* eot_row, int32_t B, int32_t L, int32_t D,
                        int32_t vocab, void* stream);

/* fp32 -> bf16 conversion of a dense buffer (weights upload, feature casts). n % 4 == 0. */
int hirest_f32_to_bf16(const float* in, hirest_bf16* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * Pooling + scoring (inference_video_retrieval.py:283-285, 323-334; evaluate.py:58-60)
 * ------------------------------------------------------------------------------------ */
/* [V,F,E] f32 -> [V,E] f32: (optional per-frame L2) -> mean over F -> L2.  F==1 is plain L2. */
int hirest_pool_l2norm(const float* frame_embeds, float* out, int32_t V, int32_t F, int32_t E,
                       int32_t normalize_frames_first, void* stream);
/* ragged form: out[v] = L2-normalised mean of rows seg_off[v] .. seg_off[v+1] of a packed [rows, E] matrix (seg_off: V + 1 device
 * ints; an empty segment gives zeros).  sentence-transformers' Pooling(mean over the attention mask) + Normalize. */
int hirest_pool_l2norm_varlen(const float* rows, const int32_t* seg_off, float* out, int32_t V, int32_t E, void* stream);

/* scores[q][v] = <T[q], Vn[v]> in fp32 FMA arithmetic. */
int hirest_similarity_f32(const float* text_n, const float* video_n, float* scores,
                          int32_t Q, int32_t V, int32_t E, void* stream);

/* Per-query top-k by (score desc, tie_rank desc); tie_rank NULL -> ties by higher index first
 * = the reference's sorted(zip(scores, names))[::-1] when tie_rank[v] = rank of names[v]. */
int hirest_topk_f32(const float* scores, const int32_t* tie_rank, int32_t Q, int32_t V, int32_t k,
                    int32_t* out_index, float* out_score, void* stream);
/* The same selection for long rows (beam search over beams*vocab scores): per-chunk candidates in parallel, then a merge.
 * workspace: hirest_topk_workspace_bytes; identical results. */
int64_t hirest_topk_workspace_bytes(int32_t Q, int32_t V, int32_t k);
int hirest_topk_f32_ws(const float* scores, const int32_t* tie_rank, int32_t Q, int32_t V, int32_t k,
                       int32_t* out_index, float* out_score, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Whole-tower runners: one call = one forward of a transformer tower over a batch, all
 * kernels enqueued on `stream`.  Weights are referenced, never copied.
 * ------------------------------------------------------------------------------------ */
typedef struct hirest_block_weights {      /* pre-LN transformer block */
    const float* ln1_g; const float* ln1_b;
    const hirest_bf16* qkv_w; const float* qkv_b;     /* [3D,D], [3D] (EVA: q_bias,0,v_bias) */
    const hirest_bf16* proj_w; const float* proj_b;   /* [D,D] */
    const float* ln2_g; const float* ln2_b;
    const hirest_bf16* fc1_w; const float* fc1_b;     /* [Dm,D] */
    const hirest_bf16* fc2_w; const float* fc2_b;     /* [D,Dm] */
    /* Optional (all six or none; vision tower, gelu_erf only): LayerNorm-folded operands of the two GEMMs that follow a
     * LayerNorm — W' = W * gamma (bf16), b' = b + W beta, s = row sums of the bf16 W' (see HIREST_EPI_LNFOLD_*).  When
     * present, tower calls of >= 64 frames skip the LayerNorm passes of the block. */
    const hirest_bf16* qkv_wf; const float* qkv_bf; const float* qkv_s;
    const hirest_bf16* fc1_wf; const float* fc1_bf; const float* fc1_s;
} hirest_block_weights;

typedef struct hirest_vision_tower {       /* EVA ViT (vit_model.py:248-351) */
    int32_t image_size, patch, width, heads, head_dim, mlp_dim, layers, embed_dim;
    int32_t kpad;                          /* padded 3*P*P (multiple of 64) */
    int32_t act;                           /* 0 gelu_erf, 1 quick_gelu */
    float ln_eps;
    const hirest_bf16* patch_w;            /* [width, kpad] */
    const float* patch_b;                  /* [width] */
    const float* cls;                      /* [width] */
    const float* pos;                      /* [tokens, width] */
    const hirest_block_weights* blocks;    /* HOST array [layers] */
    const float* norm_g; const float* norm_b;
    const hirest_bf16* head_w;             /* [embed_dim, width] */
    const float* head_b;                   /* [embed_dim] or NULL */
    const float* image_mean; const float* image_std; /* device [3], used when in_dtype==2 */
    /* OpenAI-CLIP ViT variant as vendored by the reference (EVA_clip/model.py:216-273): */
    const float* ln_pre_g; const float* ln_pre_b;    /* LayerNorm right after cls/pos (NULL = none) */
    int32_t out_all_tokens;  /* 0: norm -> CLS row -> head = [B,E];  1: ln_post + proj on every token = [B,T,E]
                                (the caller drops token 0: model.py:269-273 returns the patch tokens) */
} hirest_vision_tower;

size_t hirest_vision_workspace_bytes(const hirest_vision_tower* t, int32_t B);
/* frames: see hirest_patchify in_dtype; out: f32 [B, embed_dim] (not normalised), like
 * EVA_CLIP.encode_image (eva_model.py:317).
 * flags: HIREST_TOWER_NO_LNFOLD runs the LayerNorm passes even where the folded form would apply (same workspace).
 * A folded call leaves, at byte offset hirest_vision_guard_offset(t, B) of the workspace, one float = the largest
 * |mean| / sigma any token row of any layer had (see hirest_rowstats_bf16); the offset is (size_t)-1 for calls that do
 * not fold.  The host layer re-runs a call with HIREST_TOWER_NO_LNFOLD when that value exceeds its threshold. */
enum { HIREST_TOWER_NO_LNFOLD = 1,
       /* run the last block on every token (A/B timing and the bit-identity test of the pruned form) */
       HIREST_TOWER_NO_PRUNE = 2,
       /* folded calls: keep the residual stream in the fp32 array x between the blocks (rounds 1-3) instead of as two bf16 arrays hi + lo
        * (16 significand bits, HIREST_EPI_BIAS_RESID2_LNSTATS: a fifth less epilogue traffic on proj / fc2) — A/B timing and the numerics
        * test of the two-array form */
       HIREST_TOWER_F32_RESIDUAL = 4 };
int hirest_vision_forward(const hirest_vision_tower* t, const void* frames, int32_t in_dtype,
                          int32_t B, float* out, void* workspace, size_t workspace_bytes,
                          int32_t flags, void* stream);
size_t hirest_vision_guard_offset(const hirest_vision_tower* t, int32_t B);

typedef struct hirest_text_tower {         /* CLIP text transformer (eva_model.py:177-250) */
    int32_t context, vocab, width, heads, layers, embed_dim;
    int32_t act;
    float ln_eps;
    const float* tok_emb;                  /* [vocab, width] f32 */
    const float* pos;                      /* [context, width] */
    const hirest_block_weights* blocks;    /* HOST array [layers] */
    const float* lnf_g; const float* lnf_b;
    const hirest_bf16* proj_w;             /* text_projection transposed: [embed_dim, width] */
} hirest_text_tower;

size_t hirest_text_workspace_bytes(const hirest_text_tower* t, int32_t B);
int hirest_text_forward(const hirest_text_tower* t, const int64_t* tokens, int32_t B, float* out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Reference-precision towers (csrc/tower_f32.hip): the same forwards with every product in exact fp32 (hirest_gemm_f32,
 * hirest_attention_f32_qkv incl. 88-wide heads, fp32 LayerNorm and activations).  What the host layer runs for
 * precision='fp32' — the reference's own default (EVA_clip/eva_clip.py:90; modeling.py:120 `.float()`): retrieval ranks then
 * equal the fp32 reference's wherever fp32 itself decides them.  ~16x the cost of the bf16 towers; fields as in
 * hirest_block_weights / hirest_vision_tower / hirest_text_tower with fp32 weights [out, in] as nn.Linear stores them.
 * ------------------------------------------------------------------------------------ */
typedef struct hirest_block_weights_f32 {
    const float* ln1_g; const float* ln1_b;
    const float* qkv_w; const float* qkv_b;
    const float* proj_w; const float* proj_b;
    const float* ln2_g; const float* ln2_b;
    const float* fc1_w; const float* fc1_b;
    const float* fc2_w; const float* fc2_b;
} hirest_block_weights_f32;

typedef struct hirest_vision_tower_f32 {
    int32_t image_size, patch, width, heads, head_dim, mlp_dim, layers, embed_dim;
    int32_t kpad;                          /* padded 3*P*P (multiple of 16) */
    int32_t act;                           /* 0 gelu_erf, 1 quick_gelu */
    float ln_eps;
    const float* patch_w;                  /* [width, kpad], zero padded */
    const float* patch_b; const float* cls; const float* pos;
    const hirest_block_weights_f32* blocks;    /* HOST array [layers] */
    const float* norm_g; const float* norm_b;
    const float* head_w; const float* head_b;  /* [embed_dim, width], [embed_dim] or NULL */
    const float* image_mean; const float* image_std;
    const float* ln_pre_g; const float* ln_pre_b;
    int32_t out_all_tokens;
} hirest_vision_tower_f32;

typedef struct hirest_text_tower_f32 {
    int32_t context, vocab, width, heads, layers, embed_dim;
    int32_t act;
    float ln_eps;
    const float* tok_emb; const float* pos;
    const hirest_block_weights_f32* blocks;    /* HOST array [layers] */
    const float* lnf_g; const float* lnf_b;
    const float* proj_w;                       /* [embed_dim, width] = text_projection^T */
} hirest_text_tower_f32;

size_t hirest_vision_workspace_bytes_f32(const hirest_vision_tower_f32* t, int32_t B);
int hirest_vision_forward_f32(const hirest_vision_tower_f32* t, const void* frames, int32_t in_dtype, int32_t B, float* out,
                              void* workspace, size_t workspace_bytes, void* stream);
/* Front end of hirest_vision_forward_f32 alone: patch embedding (+ bias + position rows), CLS rows, ln_pre when present, into the fp32
 * residual stream x [B*T, width].  `rows`: scratch of B*T*kpad floats (the im2col rows). */
int hirest_vision_embed_f32(const hirest_vision_tower_f32* t, const void* frames, int32_t in_dtype, int32_t B, float* x, float* rows,
                            void* stream);
size_t hirest_text_workspace_bytes_f32(const hirest_text_tower_f32* t, int32_t B);
int hirest_text_forward_f32(const hirest_text_tower_f32* t, const int64_t* tokens, int32_t B, float* out,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * "bf16x3" vision tower (precision='bf16x3'): the fp32 forward above with the four weight GEMMs of every block computed from bf16
 * hi + lo splits of both fp32 operands (HIREST_GEMM_X3: w_hi a_lo + w_lo a_hi + w_hi a_hi, fp32 accumulation) — ~16-bit products at
 * 3 bf16 MFMAs each, i.e. 3/16 of the exact-fp32 matrix cost; LayerNorm, attention, GELU, the residual stream, patch embedding and
 * head stay fp32 (tower_f32's kernels).  Meant to reproduce the fp32 reference's retrieval RANKS (EVA_clip/eva_clip.py:90) at
 * several times the exact-fp32 tower's throughput.  EVA towers only (GELU, no ln_pre, CLS head); width, mlp_dim % 32 == 0.
 *   hirest_split2_bf16        [rows, D] fp32 -> [rows, 2D] bf16 in the X3 operand format (each 64-column block: hi of 32 k | lo of
 *                             the same k); act 1 applies nn.GELU()'s erf form in fp32 first.  Also splits the weights (once).
 *   hirest_layernorm_split2   LayerNorm over the last dim (hirest_layernorm's arithmetic) stored in that format.
 * ------------------------------------------------------------------------------------ */
typedef struct hirest_block_weights_x3 {
    const hirest_bf16* qkv_w2;  /* [3*width, 2*width]   split of hirest_block_weights_f32.qkv_w  */
    const hirest_bf16* proj_w2; /* [width,   2*width]  */
    const hirest_bf16* fc1_w2;  /* [mlp_dim, 2*width]  */
    const hirest_bf16* fc2_w2;  /* [width,   2*mlp_dim] */
} hirest_block_weights_x3;

typedef struct hirest_vision_tower_x3 {
    const hirest_vision_tower_f32* base;       /* dimensions, fp32 patch / cls / pos / LayerNorm / bias / head parameters (HOST struct) */
    const hirest_block_weights_x3* blocks;     /* HOST array [layers] */
} hirest_vision_tower_x3;

int hirest_split2_bf16(const float* x, int64_t ldx, hirest_bf16* out, int64_t ldo, int64_t rows, int32_t D, int32_t act, void* stream);
/* The split of x^T without a transposed copy: x fp32 [rows, cols] -> out bf16 [cols, 2 rows] (row c = column c of x in the operand format
 * above) — the B operand of dX = dY W for a Linear's weight W [out_features, in_features] (module_visual.py's dense layers under
 * run.py:238-295).  rows % 32 == 0. */
int hirest_split2_transposed_bf16(const float* x, int64_t ldx, hirest_bf16* out, int64_t ldo, int32_t rows, int32_t cols, void* stream);
/* Any number of matrices split in one call (launches of HIREST_SPLIT_GROUP_MAX items), each as by hirest_split2_bf16 (transposed = 0, act 0;
 * cols % 32 == 0) or hirest_split2_transposed_bf16 (transposed = 1; rows that are no multiple of 32 are zero filled up to one, ldo >= 2 x that):
 * a training step splits its encoder weights both ways once per step.  `items` is a HOST array, copied into the kernel arguments. */
#define HIREST_SPLIT_GROUP_MAX 16
typedef struct hirest_split_item {
    const float* x; hirest_bf16* out;
    int64_t ldx, ldo;
    int32_t rows, cols, transposed, reserved;
} hirest_split_item;
int hirest_split2_grouped_bf16(const hirest_split_item* items, int32_t count, void* stream);
int hirest_layernorm_split2(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, hirest_bf16* out, int64_t ldo,
                            int32_t rows, int32_t D, void* stream);
/* softmax(q k^T * scale) v for fp32 q / k / v rows (row strides ldq / ldkv, head h at column h * dh; out [B * Tq, H * dh]) with both products
 * formed from bf16 hi + lo splits (three bf16 MFMAs per product, fp32 accumulation), softmax in fp32: the attention of the bf16x3 tower
 * (vit_model.py:127-147; no mask).  dh % 4 == 0, dh <= 96; pointers 16-B aligned. */
int hirest_attention_x3_qkv(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out, int32_t B, int32_t Tq,
                            int32_t Tk, int32_t H, int32_t dh, float scale, void* stream);
/* the same with the output stored as the split operand [B * Tq, 2 * H * dh] bf16 of the GEMM that follows ((H * dh) % 32 == 0) */
int hirest_attention_x3_qkv_split2(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, hirest_bf16* out2, int32_t B,
                                   int32_t Tq, int32_t Tk, int32_t H, int32_t dh, float scale, void* stream);
/* Waves (32 queries each) per workgroup of the split-operand attention: 0 = automatic, 3 / 4 / 8 / 9 force.  Every workgroup stages all
 * K / V tiles of its (frame, head), so fewer, larger workgroups read K / V fewer times.  Results are unchanged (per-query arithmetic). */
int hirest_attention_x3_select_waves(int32_t waves);
/* Timing tool: workgroup 0 of every following launch stamps the shader clock at six points of every 32-key tile into device_buffer
 * ([tile < 16][wave < 16][8] int64: 0 top, 1 after the barrier, 6 the next tile's loads landed, 7 split + LDS stores done, 2 next fetch issued,
 * 3 scores issued, 4 softmax done, 5 P.V issued); NULL switches it off (default).  tools/attn_x3_trace.py. */
int hirest_attention_x3_debug_trace(int64_t* device_buffer);
/* A/B and tests: bit 0 = the tower's attention is the exact-fp32 hirest_attention_f32_qkv instead of hirest_attention_x3_qkv; bit 1 = fc1 writes fp32
 * and GELU + split run as a separate pass (hirest_split2_bf16) instead of in its epilogue (HIREST_EPI_BIAS_GELU_SPLIT2).  Default 0. */
int hirest_vision_x3_select_attention(int32_t which);
size_t hirest_vision_workspace_bytes_x3(const hirest_vision_tower_x3* t, int32_t B);
int hirest_vision_forward_x3(const hirest_vision_tower_x3* t, const void* frames, int32_t in_dtype, int32_t B, float* out,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Joint model (MomentModel) path — fp32 end to end, because its outputs are frame INDICES
 * (modeling.py:155-474; clip4caption/modules/module_visual.py:104-264,396-424).
 * ------------------------------------------------------------------------------------ */
/* out = act(A @ W^T + bias) (+ resid) (+ periodic[m % period]) in exact fp32 (k-ordered fmaf chain on
 * v_mfma_f32_32x32x2_f32).  act: 0 none, 1 gelu(erf), 2 tanh, 3 quick-gelu.  K % 16 == 0, N % 4 == 0.
 * Replaces the nn.Linear layers of the fusion, VisualEmbeddings (periodic = position embeddings),
 * VisualSelfOutput / VisualOutput (resid = the residual that precedes the post-LayerNorm). */
int hirest_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                    const float* resid, int64_t ldr, const float* periodic, int32_t period,
                    float* out, int64_t ldo, int32_t M, int32_t N, int32_t K, int32_t act, void* stream);
/* hirest_gemm_f32 with scratch memory: problems of few 64x64 tiles (256 < M, at most 512 tiles, K >= 1024 — a 768-wide layer over
 * 1500 rows is 288 tiles for 256 CUs) run one block per (tile, K quarter) into the workspace and a second kernel adds the four
 * partial sums in the kernel's own order, so the result has the same bits.  hirest_gemm_f32_workspace_bytes returns what the
 * problem wants (0: the plain form is taken); a missing / short workspace falls back to hirest_gemm_f32. */
size_t hirest_gemm_f32_workspace_bytes(int32_t M, int32_t N, int32_t K);
int hirest_gemm_f32_ws(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
                       const float* periodic, int32_t period, float* out, int64_t ldo, int32_t M, int32_t N, int32_t K, int32_t act,
                       void* workspace, size_t workspace_bytes, void* stream);
/* out = act(A B^T + bias) (+ resid) with either operand optionally stored k-major (a_kmajor: A(m, k) at A[k * lda + m], M % 4 == 0;
 * w_kmajor: W(n, k) at W[k * ldw + n]) — the products of a linear layer's backward pass (dX = dY W: w_kmajor; dW = dY^T X: both)
 * without transposed copies.  Same kernel, same summation order and bits as hirest_gemm_f32 on the transposed, zero-padded copies;
 * K is free for a k-major operand (K % 4 == 0 otherwise).  The workspace (optional) enables the split form for few-tile problems. */
size_t hirest_gemm_f32_layouts_workspace_bytes(int32_t M, int32_t N, int32_t K);
int hirest_gemm_f32_layouts(const float* A, int64_t lda, int32_t a_kmajor, const float* W, int64_t ldw, int32_t w_kmajor,
                            const float* bias, const float* resid, int64_t ldr, float* out, int64_t ldo, int32_t M, int32_t N,
                            int32_t K, int32_t act, void* workspace, size_t workspace_bytes, void* stream);
/* 0 = automatic (M <= 256 rows and K % 32 == 0 [and N < 8192 above 32 rows]: 16-column tiles of v_mfma_f32_16x16x4_f32, four waves
 * share K, operands by LDS-DMA; other M <= 256: the split-K "skinny" kernel, 32x32 tiles; otherwise 64x64 tiles), 1 = always the
 * 64x64 kernel, 2 = automatic without the 16-column kernel.  All three are exact fp32 MFMA and add the same products in the same order: bit-identical results (tests / A-B timing). */
int hirest_gemm_f32_select_kernel(int32_t which);
/* The 64x64 kernel's operand path for row-major operands with K % 32 == 0: 0 = automatic (LDS-DMA ring, 4 slots and two blocks per CU, from 64
 * tiles per CU on: the fp32 towers' layers), 1 = off (two slabs ahead through registers, the only form for k-major operands and ragged K),
 * 2 = always the ring.  Same bits in every mode (tests / A-B timing). */
int hirest_gemm_f32_ring_mode(int32_t mode);
/* out = act(LayerNorm(X; gamma, beta, eps) @ W^T + bias) (+ resid) for M <= 256 rows (N < 8192; the LM-head form N >= 8192, K = 768
 * takes M <= 32), K % 256 == 0, K <= 1024: the rows are
 * normalised inside the GEMM with hirest_layernorm's own arithmetic (same bits as the two calls), and written to ln_out as well
 * when it is not NULL.  With ids != NULL, X[r] = table[ids[r]] + pos_row (table rows of K floats): a decoding step's token +
 * position embedding (module_decoder.py embeddings + LayerNorm).  Replaces hirest_layernorm + hirest_gemm_f32 (and
 * hirest_embedding_pos_fwd_f32) in the step-captioning decoder, where a launch costs as much as the arithmetic. */
int hirest_gemm_f32_ln(const float* X, int64_t ldx, const int32_t* ids, const float* table, const float* pos_row,
                       const float* gamma, const float* beta, float eps, float* ln_out, int64_t ldl, const float* W, int64_t ldw,
                       const float* bias, const float* resid, int64_t ldr, float* out, int64_t ldo, int32_t M, int32_t N, int32_t K,
                       int32_t act, void* stream);
/* The LM-head form of hirest_gemm_f32_ln (N >= 8192, K = 768, rows given directly, no ln_out, no activation / residual) that also
 * writes colmax[M, ceil(N / 16)]: per row, the maximum of each 16-column tile of `out` (what hirest_caption_beam_step's tail uses
 * instead of scanning the rows for their maximum). */
int hirest_gemm_f32_ln_colmax(const float* X, int64_t ldx, const float* gamma, const float* beta, float eps, const float* W, int64_t ldw,
                              const float* bias, float* out, int64_t ldo, float* colmax, int32_t M, int32_t N, int32_t K, void* stream);
/* out = A @ W^T + bias for the beam rows of a MERGED search (K = 768; any M, meant for 33 .. 256): row groups of 32 - 80 rows, each wave's A
 * fragments in registers, W streamed once per row group through LDS-DMA rings (the groups of a column stream share an XCD's L2); the
 * shared fp32 summation order, so bit-identical to hirest_gemm_f32.  colmax (may be NULL) [M, ceil(N / 16)]: per row, the maximum of
 * each 16-column tile of `out`.  The LM head of clip4caption's decoder (module_decoder.py:247-277) at the reference's default
 * --eval_batch_size 32 (args.py:27). */
/* A/B switch of hirest_gemm_f32_ln's row-group form (above 32 rows): 0 = one block per CU with a 12-slab ring and 1 - 3 row tiles per wave,
 * 1 / 2 = one row tile per wave, 4-slab rings, one / two blocks per CU (2 is the default).  Same bits. */
int hirest_gemm_f32_rows_ln_mode(int32_t mode);
/* 1 when hirest_gemm_f32 dispatches an M-row, N >= 8192, K = 768 product to that kernel (its row groups pad M less than 64-row tiles do), 0 when the
 * 64x64 kernel takes it. */
int hirest_gemm_f32_rows_preferred(int32_t M);
int hirest_gemm_f32_rows_colmax(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* out, int64_t ldo,
                                float* colmax, int32_t M, int32_t N, int32_t K, void* stream);
/* softmax(fl(fl(q.k*scale) + add_const)) v over packed fp32 qkv [B*T, 3*H*dh]; no key masking (the
 * reference passes an all-zeros mask, i.e. add_const = -10000 on every score: SURVEY hazard H3).  dh: any multiple of 4 up to 96
 * (instantiations for 32, 64 and 96 columns; a head gives the same bits in each one that holds it); blocks of one wave (sequences of
 * at most 64 queries), three or four waves of 32 queries, chosen per call — a query's result does not depend on the choice. */
int hirest_attention_f32(const float* qkv, float* out, int32_t B, int32_t T, int32_t H, int32_t dh,
                         float scale, float add_const, void* stream);
/* General form: separate q [B*Tq, ldq] and k/v [B*Tk, ldkv] (cross-attention), plus `causal_penalty` added to
 * every score with key > query (the decoder's self-attention adds -10000 there, NOT -inf: module_decoder.py:394-397). */
int hirest_attention_f32_qkv(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out,
                             int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t dh, float scale, float add_const,
                             float causal_penalty, void* stream);
/* Packed ragged self-attention: sequence b occupies rows seq_off[b] .. seq_off[b+1] of qkv / out (seq_off: B + 1 device ints,
 * max_len >= the longest sequence).  No padding rows and no key mask exist; a library that pads to the longest sequence and adds a
 * -inf-like mask on the pad keys computes the same numbers (the ASR sentence encoder, hirest_amd/sentence_encoder.py). */
int hirest_attention_f32_varlen(const float* qkv, float* out, const int32_t* seq_off, int32_t B, int32_t max_len, int32_t H,
                                int32_t dh, float scale, float add_const, void* stream);
/* out[r][:] = log_softmax(x[r][:]) + row_add[r]  (train.py:563-564 + the beam score add of beam.py:76); row_add may be NULL */
int hirest_log_softmax_f32(const float* x, int64_t ldx, const float* row_add, float* out, int64_t ldo, int32_t rows,
                           int32_t V, void* stream);
/* tin[b,t,:] = tanh(time(b,t)*w1 + b1), time = (linspace(0,1,n_valid[b])[t]-0.5)*2, 0 past n_valid (modeling.py:176-195) */
int hirest_joint_time_features(const int32_t* n_valid, const float* w1, const float* b1, float* tin,
                               int32_t B, int32_t T, int32_t E, void* stream);
/* grid[b][t] = that time(b,t) itself, [B, T] fp32: the per-row weight of temporal_embed.0.weight's gradient (a column sum of
 * d pre-activation * time), built on the device so that the training step never reads n_valid back (modeling.py:176-193) */
int hirest_joint_time_grid_f32(const int32_t* n_valid, int32_t B, int32_t T, float* grid, void* stream);
/* base = v * (text_proj/||text_proj||)[:,None,:] + asr + temporal   (modeling.py:163-195, loop invariant) */
int hirest_joint_base(const float* v, const float* text_proj, const float* asr, const float* temporal, float* base,
                      int32_t B, int32_t T, int32_t E, void* stream);
/* f = base (+ boundary_embed[boundary_mask]) + mask_embed[moment_mask]   (modeling.py:171-173,197-198) */
int hirest_joint_mask_add(const float* base, const int32_t* moment_mask, const int32_t* boundary_mask,
                          const float* mask_embed, const float* boundary_embed, float* f, int64_t rows, int32_t E,
                          void* stream);
/* up to three Linear(D,1) heads: logits[h*rows + r] = <x[r], w_h> + bias3[h] */
int hirest_linear_heads(const float* x, int64_t rows, int32_t D, int32_t nheads, const float* w0, const float* w1,
                        const float* w2, const float* bias3, float* logits, void* stream);
/* ------------------------------------------------------------------------------------
 * Step-captioning decoder, one beam-search step per call (clip4caption/modules/module_decoder.py:279-406 as driven by
 * clip4caption/train.py:511-599): embeds every beam's newest token, runs the post-LN decoder layers with each beam's kept
 * self-attention K / V (gathered from its parent beam's row of the previous step), cross-attends to the encoded frames, applies the
 * LM head and returns log_softmax + row_add.  Same numbers as re-running the whole prefix (the reference's -10000 "causal" penalty
 * is exactly 0 after exp in fp32).  All pointers are device pointers except the HOST arrays marked so.  head width must be 64.
 * ------------------------------------------------------------------------------------ */
typedef struct hirest_caption_layer {
    const float* qkv_w; const float* qkv_b;                         /* self-attention query | key | value, fused [3D, D], [3D] */
    const float* so_w; const float* so_b; const float* so_ln_g; const float* so_ln_b;     /* slf_attn.output dense + LayerNorm */
    const float* cq_w; const float* cq_b;                           /* enc_attn.att.query */
    const float* co_w; const float* co_b; const float* co_ln_g; const float* co_ln_b;     /* enc_attn.output dense + LayerNorm */
    const float* ff1_w; const float* ff1_b;                         /* intermediate.dense [I, D] (gelu) */
    const float* ff2_w; const float* ff2_b; const float* ff_ln_g; const float* ff_ln_b;   /* output.dense [D, I] + LayerNorm */
} hirest_caption_layer;
typedef struct hirest_caption_decoder {
    int32_t layers, heads, hidden, inter, vocab_padded, max_pos;
    const float* word_emb; const float* pos_emb; const float* emb_ln_g; const float* emb_ln_b;
    const hirest_caption_layer* layer;                              /* HOST array [layers] */
    const float* tr_w; const float* tr_b; const float* tr_ln_g; const float* tr_ln_b;     /* cls.predictions.transform */
    const float* lm_w; const float* lm_b;                           /* [vocab_padded, D] (tied to word_emb), [vocab_padded] */
    const hirest_bf16* lm_w2;                                       /* optional (ABI 4): lm_w in the split operand format [vocab_padded, 2 D]
                                                                     * (hirest_split2_bf16).  When set, steps of >= 64 rows run the LM head on
                                                                     * split operands (three bf16 MFMAs per product: 84.8 us of a 329-us word were
                                                                     * on the 1/16-rate fp32 MFMA at 160 rows); below that the product is bound by
                                                                     * the weight stream and stays exact fp32.  NULL: exact fp32 at every size. */
} hirest_caption_decoder;
size_t hirest_caption_step_workspace_bytes(const hirest_caption_decoder* d, int32_t R);
/* R beams (rows).  position = index of the newest token (0 for [CLS]).  kv_in / kv_out: HOST arrays of 2 * layers device buffers
 * (K0, V0, K1, V1, ...): kv_in[.] holds [R, position, D] from the previous step (unused at position 0), kv_out[.] receives
 * [R, position + 1, D]; they must not alias.  parent_rows[r] = the row of the previous step beam r continues (NULL at position 0).
 * enc_kv: HOST array [layers] of [R, F, 2 D] cross-attention key | value rows.  logp: [R, vocab_padded]. */
int hirest_caption_decode_step(const hirest_caption_decoder* d, int32_t R, int32_t position, const int32_t* last_ids,
                               const int32_t* parent_rows, const float* const* kv_in, float* const* kv_out,
                               const float* const* enc_kv, int32_t F, const float* row_add, float* logp,
                               void* workspace, size_t workspace_bytes, void* stream);

/* hirest_attention_f32_qkv for ONE query per row (a decoding step), head width 64, one wave per (row, head), same bits.  Keys of
 * row r: j < t_hist at k_hist / v_hist + ((parent ? parent[r] : r) * t_hist + j) * ld_hist, then (k_new != NULL) one more at
 * k_new / v_new + r * ld_new.  k_out / v_out (optional): [R, T, H * 64] receive the keys / values used, T = t_hist (+ 1) — the beam's
 * history re-gathered by parent row with the newest position appended.  The score of key j gets add_const (+ causal_penalty for
 * j > 0, as hirest_attention_f32_qkv does for query 0). */
int hirest_attention_f32_decode(const float* q, int64_t ldq, const float* k_hist, const float* v_hist, int64_t ld_hist,
                                const int32_t* parent, int32_t t_hist, const float* k_new, const float* v_new, int64_t ld_new,
                                float* k_out, float* v_out, float* out, int32_t R, int32_t H, float scale, float add_const,
                                float causal_penalty, void* stream);
/* 0 (default) = the decoder step runs its LayerNorms and the token embedding as prologues of the GEMMs that consume them
 * (hirest_gemm_f32_ln) and its attentions as hirest_attention_f32_decode (history read in place), 1 = LayerNorm, embedding, K / V
 * gather and hirest_attention_f32_qkv as separate kernels.  Same bits either way (tests / A-B timing). */
int hirest_caption_select(int32_t mode);
/* The same step up to the LM head: raw logits [R, vocab_padded] instead of log-probabilities (input of hirest_caption_beam_tail). */
int hirest_caption_decode_logits(const hirest_caption_decoder* d, int32_t R, int32_t position, const int32_t* last_ids,
                                 const int32_t* parent_rows, const float* const* kv_in, float* const* kv_out,
                                 const float* const* enc_kv, int32_t F, float* logits, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* The rest of a beam-search step in two kernels: log_softmax(logits) + row_add (train.py:563-564, beam.py:76; same bits as
 * hirest_log_softmax_f32), the top `beam` of every sample's beam x vocab scores (same strict order as hirest_topk_f32: higher
 * score, then higher flat index) and hirest_beam_advance's bookkeeping — without materialising the log-probabilities.  logits:
 * [B * beam, vocab] with row stride ldx, 16-byte aligned, vocab % 4 == 0, beam <= 16.  done_host (optional): pinned host int32 [B]
 * that receives, per sample, ((step + 1) << 1) | done — one 4-byte store each, so a host that finds the stamp of step s in all B words
 * has that step's flags without recording an event.  The other arguments are hirest_beam_advance's. */
size_t hirest_caption_beam_tail_workspace_bytes(int32_t B, int32_t beam, int32_t vocab);
int hirest_caption_beam_tail(const float* logits, int64_t ldx, const float* row_add, int32_t B, int32_t beam, int32_t vocab,
                             int32_t step, int32_t max_steps, int32_t eos_id, float* scores, int32_t* tokens, int32_t* backptr,
                             int32_t* n_steps, int32_t* done, int32_t* next_ids, int32_t* next_parents, float* next_add,
                             int32_t* done_host, void* workspace, size_t workspace_bytes, void* stream);

/* One word of the beam search in one call = hirest_caption_decode_logits + hirest_caption_beam_tail on R = B * beam rows at
 * `position` (= the tail's step).  ids / parent_rows / row_add ([R]) are the step's inputs and receive the next step's (as
 * hirest_beam_advance's next_ids / next_parents / next_add); logits: [R, vocab_padded] scratch.  workspace:
 * hirest_caption_step_workspace_bytes; tail_workspace: hirest_caption_beam_tail_workspace_bytes. */
int hirest_caption_beam_step(const hirest_caption_decoder* d, int32_t B, int32_t beam, int32_t position, int32_t* ids, int32_t* parent_rows,
                             const float* const* kv_in, float* const* kv_out, const float* const* enc_kv, int32_t F, float* row_add,
                             float* logits, int32_t max_steps, int32_t eos_id, float* scores, int32_t* tokens, int32_t* backptr,
                             int32_t* n_steps, int32_t* done, int32_t* done_host, void* workspace, size_t workspace_bytes,
                             void* tail_workspace, size_t tail_workspace_bytes, void* stream);

/* Beam bookkeeping of one decoding step on the device (clip4caption/modules/beam.py:70-92): val / idx = the sorted top-`beam` of
 * every sample's beam x vocab scores ([B, beam]; idx = source_beam * vocab + word).  Per sample b that is not done: scores <- val,
 * tokens[b][step][k] / backptr[b][step][k] record (word, source beam), n_steps[b] = step + 1, done[b] = 1 once the best beam emits
 * eos_id; next_ids / next_parents / next_add ([B * beam]) are the inputs of the next hirest_caption_decode_step.  Done samples keep
 * their state and get inert rows.  tokens / backptr: int32 [B, max_steps, beam]. */
int hirest_beam_advance(const float* val, const int32_t* idx, int32_t B, int32_t beam, int32_t vocab, int32_t step, int32_t max_steps,
                        int32_t eos_id, float* scores, int32_t* tokens, int32_t* backptr, int32_t* n_steps, int32_t* done,
                        int32_t* next_ids, int32_t* next_parents, float* next_add, void* stream);
/* The read-out of a finished search (clip4caption/train.py:590-599 with n_best = 1): per sample the best beam — highest score, lowest beam
 * number among equal scores — walked back through backptr.  out int32 [B, max_steps + 1]: out[b][0] = n_steps[b], then that many words. */
int hirest_beam_backtrack(const float* scores, const int32_t* tokens, const int32_t* backptr, const int32_t* n_steps, int32_t B,
                          int32_t beam, int32_t max_steps, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Joint model, training side: backward of MomentModel.train_moment_retrieval (modeling.py:155-270) in exact fp32.
 * Matrix products of the backward pass are hirest_gemm_f32 calls on transposed operands (dX = dY W: A = dY, W-operand = W^T;
 * dW = dY^T X: A = dY^T, W-operand = X^T over the zero-padded row count); the entries below are everything that is not a GEMM.
 * ------------------------------------------------------------------------------------ */
/* C[m][n] = alpha * sum_k A[m][k] B[n][k] with A[m][k] at A + sam m + sak k and B[n][k] at B + sbn n + sbk k (one stride of
 * each pair must be 1): exact fp32 MFMA, 64 x 64 tiles.  Lets dX = dY W (B = W read column-wise) and dW = dY^T X (both operands
 * read column-wise) run without materialised transposes. */
int hirest_gemm_f32_strided(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk, float* C, int64_t ldc,
                            int32_t M, int32_t N, int32_t K, float alpha, void* stream);
/* out[c][r] = r < R ? in[r][c] : 0 for r < Rp (Rp >= R; pad the reduction dimension of a dW GEMM to a multiple of 16) */
int hirest_transpose_pad_f32(const float* in, int64_t ld_in, int32_t R, int32_t C, float* out, int32_t Rp, void* stream);
/* out[c] = sum_r w(r) x[r][c], w(r) = (row_weight ? row_weight[r] : 1) * (row_select ? row_select[r] == select_value : 1):
 * bias gradients (no weights), LayerNorm gamma / beta gradients, nn.Embedding(2, E) rows (select), Linear(D, 1) head weights and
 * the Linear(1, E) time embedding (weights) */
int hirest_weighted_colsum_f32(const float* x, int64_t ldx, const float* row_weight, const int32_t* row_select,
                               int32_t select_value, int32_t R, int32_t C, float* out, void* stream);
/* The same sums for up to any number of matrices in one call (launches of HIREST_COLSUM_GROUP_MAX items): a training step's bias,
 * LayerNorm and embedding gradients are ~36 such sums of a few blocks each, which cost more in launches than in work.  Each item is
 * computed exactly as by hirest_weighted_colsum_f32 (same bits).  `items` is a HOST array, copied into the kernel arguments; the
 * matrices it points to must stay alive and unchanged until the call has been enqueued AND executed on `stream`. */
#define HIREST_COLSUM_GROUP_MAX 40
typedef struct hirest_colsum_item {
    const float*   x;            /* [R, C] fp32, row stride ldx */
    const float*   row_weight;   /* [R] or NULL */
    const int32_t* row_select;   /* [R] or NULL: only rows with row_select[r] == select_value count */
    float*         out;          /* [C] */
    int64_t        ldx;
    int32_t        R, C, select_value, reserved;
} hirest_colsum_item;
int hirest_weighted_colsum_grouped_f32(const hirest_colsum_item* items, int32_t count, void* stream);
/* y = act(pre) and dx = dy * act'(pre);  act: 0 identity, 1 gelu (erf form), 2 tanh; backward only: 3 = tanh given its OUTPUT
 * in `pre` (1 - y^2) */
int hirest_act_f32(const float* pre, float* y, int64_t n, int32_t act, void* stream);
/* x[i] *= *scalar (one device float): the upstream gradient of loss.backward() applied to the loss gradient without a host read */
int hirest_scale_by_device_scalar_f32(float* x, const float* scalar, int64_t n, void* stream);
int hirest_act_bwd_f32(const float* pre, const float* dy, float* dx, int64_t n, int32_t act, void* stream);
/* y[i] = (resid ? resid[i] : 0) + x[i] * keep(seed, i) / (1 - p): nn.Dropout with a counter-based mask, fused with the residual
 * add that follows it in VisualSelfOutput / VisualOutput (p = 0: a plain add; the same call on dy is the backward) */
int hirest_dropout_add_f32(const float* x, const float* resid, float* y, int64_t n, float p, uint32_t seed, void* stream);
/* LayerNorm backward (biased variance, eps inside the root: until_module.py:40-53 and nn.LayerNorm alike): dx, and
 * dyxhat = dy * xhat whose column sums are dgamma (column sums of dy are dbeta) */
int hirest_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float eps, float* dx, float* dyxhat,
                             int32_t R, int32_t D, void* stream);
/* Self-attention that keeps its probabilities (module_visual.py:150-183 in train mode): qkv f32 [B*T, 3*H*64];
 * P f32 [B,H,T,T] = softmax(fl(fl(q.k*scale) + add_const)), ctx [B*T, H*64] = dropout(P) v.  Backward: dS workspace [B,H,T,T],
 * dqkv [B*T, 3*H*64]. */
int hirest_attention_train_fwd_f32(const float* qkv, float* P, float* ctx, int32_t B, int32_t T, int32_t H, int32_t dh,
                                   float scale, float add_const, float drop_p, uint32_t seed, void* stream);
int hirest_attention_train_bwd_f32(const float* qkv, const float* P, const float* dctx, float* dS, float* dqkv, int32_t B,
                                   int32_t T, int32_t H, int32_t dh, float scale, float drop_p, uint32_t seed, void* stream);
/* General forms (caption decoder, module_decoder.py:192-262): separate q [B*Tq, ldq] and k / v [B*Tk, ldkv], an optional
 * additive mask [B, Tq, Tk] (the decoder's -10000 on future and padded keys) on top of add_const. */
/* 1 (default): the attention products as batched MFMA GEMMs + row kernels; 0: the one-wave-per-score-row kernels of round 2
 * (A/B timing, tests).  Same interface and buffers either way. */
int hirest_attention_train_select(int32_t which);
int hirest_attention_train_fwd_qkv_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                       const float* mask_add, float* P, float* ctx, int64_t ldctx, int32_t B, int32_t Tq, int32_t Tk,
                                       int32_t H, int32_t dh, float scale, float add_const, float drop_p, uint32_t seed, void* stream);
int hirest_attention_train_bwd_qkv_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* P,
                                       const float* dctx, int64_t ldctx, float* dS, float* dq, int64_t lddq, float* dk, float* dv,
                                       int64_t lddkv, int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t dh, float scale,
                                       float drop_p, uint32_t seed, void* stream);
/* DecoderEmbeddings (module_decoder.py:309-321): out[r] = table[ids[r]] + pos[r % T]; backward: dtable_accum[ids[r]] += dx[r]
 * (atomic adds into a zeroed or pre-filled [vocab, D] buffer: the table is tied to the LM head, whose dW is already there) */
int hirest_embedding_fwd_f32(const int32_t* ids, const float* table, const float* pos, float* out, int64_t rows, int32_t T,
                             int32_t D, void* stream);
/* the same with explicit position rows: out[r] = table[ids[r]] + pos[pos_ids[r]]  (packed ragged sequences) */
int hirest_embedding_pos_fwd_f32(const int32_t* ids, const int32_t* pos_ids, const float* table, const float* pos, float* out,
                                 int64_t rows, int32_t D, void* stream);
int hirest_embedding_bwd_f32(const int32_t* ids, const float* dx, float* dtable_accum, int64_t rows, int32_t D, void* stream);
/* CrossEntropyLoss(ignore_index = -1) over R rows of V logits (row stride ld): *loss_accum += weight * mean over the n_valid rows
 * with target >= 0; dlogits = its gradient (ignored rows: zeros).  modeling.py:140, 519 */
int hirest_ce_rows_f32(const float* logits, int64_t ld, const int32_t* target, int32_t R, int32_t V, float weight, int32_t n_valid,
                       float* loss_accum, float* dlogits, void* stream);
/* *loss_accum += weight * sum(mask * bce_with_logits(logits, onehot(target))) / max(sum mask, 1); dlogits = its gradient
 * (modeling.py:249-263) */
int hirest_bce_masked_f32(const float* logits, const int32_t* target, const int32_t* mask, int32_t B, int32_t T, float weight,
                          float* loss_accum, float* dlogits, void* stream);
/* backward of feats = v * tn[:, None, :] (modeling.py:163): dv = dbase * tn, dtn[b] = sum_t dbase * v */
int hirest_joint_base_bwd_f32(const float* dbase, const float* v, const float* tn, float* dv, float* dtn, int32_t B, int32_t T,
                              int32_t E, void* stream);
/* backward of tn = t / |t| (modeling.py:162) */
int hirest_l2norm_bwd_f32(const float* t, const float* dtn, float* dt, int32_t B, int32_t E, void* stream);
/* *loss_accum += weight * mean_b CE(softmax over the frames with mask[b,t] != 0, target[b]); dlogits = its gradient, 0 on masked
 * frames (modeling.py:343-344: logits[moment_mask == 0] = -finfo.max; F.cross_entropy) */
int hirest_ce_masked_f32(const float* logits, const int32_t* mask, const int32_t* target, int32_t B, int32_t T, float weight,
                         float* loss_accum, float* dlogits, void* stream);
/* dfeats[r] = sum_h dlogits[h*rows + r] * w_h   (backward of hirest_linear_heads with respect to its input) */
int hirest_heads_bwd_f32(const float* dlogits, int64_t rows, int32_t D, int32_t nheads, const float* w0, const float* w1,
                         const float* w2, float* dfeats, void* stream);

/* out[b] = argmax_t (mask[b,t] ? logits[b,t] : fill), first maximum (modeling.py:294-298) */
int hirest_masked_argmax(const float* logits, const int32_t* mask, float fill, int32_t B, int32_t T, int32_t* out,
                         void* stream);
/* one iteration of test_moment_segmentation's loop body for all samples, on device (modeling.py:393-433) */
int hirest_segmentation_step(const float* logits, int32_t* moment_mask, int32_t* boundary_mask, int32_t B, int32_t T,
                             double threshold, int32_t* steps, int32_t* nsteps, int32_t max_steps, float* probs_out,
                             void* stream);

/* ------------------------------------------------------------------------------------
 * Frame preprocessing on device (eva_clip.py:125-153 / clip.py:79-86):
 *   Resize(size, BICUBIC) -> CenterCrop(size) [-> ToTensor -> Normalize]
 * on raw RGB uint8 frames [B, in_h, in_w, 3], bit-exact with Pillow's 8-bit resampler
 * (Image.resize(..., BICUBIC): antialiased separable filter, 22-bit fixed-point weights,
 * horizontal pass rounded to uint8, then vertical) and torchvision's size / crop rules.
 * Only the size x size crop is computed.
 *
 * The weight tables depend on (in_h, in_w, size) only: hirest_preprocess_plan fills a HOST
 * blob of hirest_preprocess_plan_bytes bytes (double-precision arithmetic identical to
 * Pillow's precompute_coeffs/normalize_coeffs_8bpc); the caller copies it to the device once
 * and passes the device copy to every hirest_preprocess_u8 call.
 * out_kind 0: uint8 NHWC [B,size,size,3] (feed hirest_vision_forward in_dtype 2: normalisation
 *             fused into patch extraction);
 * out_kind 1: f32 NCHW [B,3,size,size] = (x/255 - mean)/std, the reference transform's tensor.
 * workspace: hirest_preprocess_workspace_bytes (the horizontally resampled rows).
 * ------------------------------------------------------------------------------------ */
int64_t hirest_preprocess_plan_bytes(int32_t in_h, int32_t in_w, int32_t size);
int hirest_preprocess_plan(int32_t in_h, int32_t in_w, int32_t size, void* host_plan, int64_t plan_bytes);
int64_t hirest_preprocess_workspace_bytes(int32_t in_h, int32_t in_w, int32_t size, int32_t B);
int hirest_preprocess_u8(const uint8_t* frames, int32_t B, int32_t in_h, int32_t in_w, int32_t size,
                         const void* plan_dev, void* out, int32_t out_kind, const float* mean3, const float* std3,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Moment-task evaluation on device (evaluate.py), double precision with Python's operation order:
 *   hirest_interval_iou_f64     compute_iou (evaluate.py:24-31) of n interval pairs a[i] (= interval_1), b[i];
 *                               R@tIoU of evaluate_moment_retrieval (:83-121) is !(iou < tIoU).
 *   hirest_step_bound_pr        per-video recall / precision of compute_step_bound_scores (:123-188): refs / preds
 *                               are ragged [sum,2] interval lists with CSR offsets [V+1]; a pair matches when
 *                               compute_iou(pred, ref) > tiou (strict).  best_iou (optional) [sum preds] = max over refs.
 *   hirest_preprocess_moment_bounds   preprocess_moment_bounds + NMS (:300-412): per video, predictions strictly inside
 *                               gt_minmax[v] = (first gt start, last gt end) -> greedy NMS from the last index ->
 *                               sorted by start, all gaps filled.  out [V,max_out,2]; out_count[v] may exceed max_out
 *                               (then the list was truncated: enlarge and rerun).  At most 128 predictions per video.
 * ------------------------------------------------------------------------------------ */
int hirest_interval_iou_f64(const double* a, const double* b, int64_t n, double* iou, void* stream);
int hirest_step_bound_pr(const double* refs, const int32_t* ref_off, const double* preds, const int32_t* pred_off,
                         int32_t V, double tiou, double* recall, double* precision, double* best_iou, void* stream);
int hirest_preprocess_moment_bounds(const double* preds, const int32_t* pred_off, const double* gt_minmax, int32_t V,
                                    double* out, int32_t* out_count, int32_t max_out, void* stream);

/* ------------------------------------------------------------------------------------
 * Frame index <-> timestamp (hirest_dataset.py:12-68; run.py:731-732,769-770 turn predicted frame indices into the
 * seconds that evaluate.py scores).  bins = np.linspace(0, int(duration) - 1, n), n = n_frames (< 0: int(duration)),
 * evaluated in double exactly as numpy builds it (arange * step, last element = stop).  Element i belongs to video
 * i / per_video (per_video = 2 for [V,2] moment bounds); n_frames is per video [V] or NULL (then n_frames_all).
 *   hirest_frame_to_timestamp   timestamp[i] = int(bins[frame[i]]) (negative indices wrap like numpy); INT64_MIN where the
 *                               reference raises IndexError (index outside the bins, or a video shorter than 1 s).
 *   hirest_timestamp_to_frame   frame[i] = min(np.digitize(t[i], bins, right=True), n - 1); INT64_MIN for videos < 1 s.
 * ------------------------------------------------------------------------------------ */
int hirest_frame_to_timestamp(const int64_t* frame, const double* duration, const int32_t* n_frames, int32_t n_frames_all,
                              int64_t per_video, int64_t n, int64_t* timestamp, void* stream);
int hirest_timestamp_to_frame(const double* t, const double* duration, const int32_t* n_frames, int32_t n_frames_all,
                              int64_t per_video, int64_t n, int64_t* frame, void* stream);

/* ------------------------------------------------------------------------------------
 * Optional per-launch timing (bench.py's live roofline measurement).  When enabled, every
 * GEMM / attention / LayerNorm launch is bracketed by hipEventRecord on ITS launch stream;
 * hirest_profile_collect synchronises those events and returns one record per launch.
 * Off by default; costs two event records per launch when on.  Not thread-safe.
 * ------------------------------------------------------------------------------------ */
enum hirest_prof_kind { HIREST_PROF_GEMM = 0, HIREST_PROF_ATTENTION = 1, HIREST_PROF_LAYERNORM = 2 };
typedef struct hirest_prof_record {
    int32_t kind;        /* enum hirest_prof_kind */
    int32_t tag;         /* GEMM: epilogue id; attention: causal flag; LN: 0 */
    int64_t d0, d1, d2;  /* GEMM: M,N,K; attention: B*H, N, dh; LN: rows, D, 0 */
    float ms;            /* elapsed device time of this launch */
} hirest_prof_record;
int hirest_profile_enable(int32_t on);                 /* also discards pending records */
int hirest_profile_collect(hirest_prof_record* out, int32_t max_records); /* returns count (<0 on error) */

/* ---------------------------------------------------------------------------------------------------------------------------------
 * Joint model, second precision (round 6): the VisualModel encoder of modeling.py:196-211 / module_visual.py:104-264,396-424 with every
 * linear layer's product on split operands (HIREST_GEMM_X3: bf16 hi + lo of both fp32 operands, three bf16 MFMAs per product, fp32
 * accumulation — ~16-bit products), everything else as the fp32 path runs it: exact-fp32 attention with the reference's uniform -10000
 * shift, fp32 LayerNorm (TF style, eps inside the root), erf-GELU in fp32, fp32 residual adds.  The reference's own reduced-precision mode
 * is torch.cuda.amp.autocast() under --fp16 (run.py:549-551); this one keeps 16 significand bits in the products, so indices, boundary
 * lists and token ids stay those of the fp32 run (tests/test_gpu_joint.py gates them on every real-reference golden).
 * Weights are split once per checkpoint with hirest_split2_bf16 ([N, K] fp32 -> [N, 2K] bf16); activations are split by the kernel that
 * produces them (LayerNorm writes fp32 + split, the GELU epilogue writes split only), the attention output by one extra pass.
 * One C call = embeddings GEMM + position rows + LayerNorm + `layers` post-LN blocks (8 launches per block) on rows = B * T. */
typedef struct hirest_joint_layer_x3 {
    const hirest_bf16* qkv_w2; const float* qkv_b;        /* [3 width, 2 width] split (query | key | value rows), bias [3 width]            */
    const hirest_bf16* ao_w2;  const float* ao_b;         /* attention.output.dense [width, 2 width]                                       */
    const float* ln1_g; const float* ln1_b;               /* attention.output.LayerNorm                                                    */
    const hirest_bf16* fc1_w2; const float* fc1_b;        /* intermediate.dense [mlp_dim, 2 width] (+ erf-GELU)                            */
    const hirest_bf16* fc2_w2; const float* fc2_b;        /* output.dense [width, 2 mlp_dim]                                               */
    const float* ln2_g; const float* ln2_b;               /* output.LayerNorm                                                              */
} hirest_joint_layer_x3;
typedef struct hirest_joint_encoder_x3 {
    uint64_t struct_size;
    int32_t layers, heads, width, mlp_dim, in_dim, max_pos;
    float ln_eps, attn_shift;                             /* 1e-12, -10000 (module_visual.py:404-406: the all-ones mask's uniform shift)    */
    const hirest_bf16* emb_w2; const float* emb_b;        /* embeddings.word_embeddings (a Linear) [width, 2 in_dim]                       */
    const float* pos;                                     /* embeddings.position_embeddings [max_pos, width]                               */
    const float* emb_ln_g; const float* emb_ln_b;
    const hirest_joint_layer_x3* layer;                   /* HOST array [layers]                                                           */
} hirest_joint_encoder_x3;
size_t hirest_joint_encoder_x3_workspace_bytes(const hirest_joint_encoder_x3* e, int32_t B, int32_t T);
/* f fp32 [B*T, in_dim] (hirest_joint_mask_add's output) -> out fp32 [B*T, width] */
int hirest_joint_encoder_x3_forward(const hirest_joint_encoder_x3* e, const float* f, int32_t B, int32_t T, float* out,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* LayerNorm over the last dim (hirest_layernorm's arithmetic) of x[r] (+ add[r % period] when add != NULL) written as fp32 (out32, may be
 * NULL) and / or in the split operand format (out2 [rows, >= 2 D] bf16, may be NULL).  D % 32 == 0, D <= 2048 (else HIREST_E_SHAPE). */
int hirest_layernorm_f32_split2(const float* x, int64_t ldx, const float* add, int32_t period, const float* gamma, const float* beta, float eps,
                                float* out32, int64_t ldo32, hirest_bf16* out2, int64_t ldo2, int32_t rows, int32_t D, void* stream);


/* ------------------------------------------------------------------------------------
 * One post-LN block of the clip4caption VisualModel in TRAIN mode (module_visual.py:132-264; modeling.py:196-211 under
 * MomentModel.train_step, run.py:238-295) issued by one C call per direction: the same kernels, operands and order as the
 * per-kernel calls above (hirest_gemm_f32_ws, hirest_attention_train_*, hirest_dropout_add_f32, hirest_layernorm,
 * hirest_act_*, hirest_layernorm_bwd_f32, hirest_gemm_f32_layouts) — bit-identical results — without ~37 trips through the
 * host language per block: at B = 5, T = 300 the step is paced by the host's enqueue rate, not by a kernel.
 *   forward:  qkv = x Wqkv^T + b;  (P, cx) = attention(qkv);  a_pre = x + drop(cx Wo^T + bo);  aa = LN1(a_pre);
 *             hpre = aa W1^T + b1;  hh = gelu(hpre);  x_pre = aa + drop(hh W2^T + b2);  out = LN2(x_pre)
 *   backward: dX products on `stream`; the four weight-gradient products dW = dY^T X on `side_stream` (NULL: on `stream`)
 *             behind an event that marks their dY complete — the caller joins the side stream before reading g_w*; the
 *             twelve column sums (bias and LayerNorm gradients) are APPENDED to `items` for one
 *             hirest_weighted_colsum_grouped_f32 by the caller once the whole backward is enqueued.
 * precision 0: exact fp32 products (hirest_gemm_f32*).  precision 1 ("bf16x3"): the forward products and the dX products on
 * split operands (HIREST_GEMM_X3 | HIREST_GEMM_X3_T128; weights split per call into `scratch`), dW stays fp32.
 * All buffers fp32, caller-allocated; `scratch` of *_scratch_bytes() must stay alive and untouched until the grouped column
 * sums and the side stream's products have executed. */
typedef struct hirest_train_block {
    uint64_t struct_size;
    int32_t B, T, heads, width, mlp, precision;
    float ln_eps, drop;
    uint32_t seed_attn, seed_ao, seed_out, reserved;      /* dropout counters: attention probabilities, attention.output, output */
    const float *wqkv, *bqkv;                             /* [3 width, width] (query | key | value rows), [3 width]             */
    const float *wo, *bo, *ln1_g, *ln1_b;                 /* attention.output.dense / LayerNorm                                 */
    const float *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;       /* intermediate.dense [mlp, width], output.dense [width, mlp], output.LayerNorm */
    const float* x;                                       /* block input [B T, width]                                           */
    float *qkv, *P, *cx, *a_pre, *aa, *hpre, *hh, *x_pre, *out;   /* kept for the backward: [R,3W] [B,H,T,T] [R,W] [R,W] [R,W] [R,mlp] [R,mlp] [R,W]; out [R,W] */
    const hirest_bf16* x2; hirest_bf16* out2;             /* precision 1 only, both optional: x / out in the split operand format [R, 2W] (a block's out2
                                                             is the next block's x2; without x2 the block splits x itself)              */
    const hirest_bf16 *wqkv2, *wo2, *w12, *w22;           /* precision 1, optional (all four or none): the weights already split (hirest_split2_bf16 /
                                                             _grouped): [3W, 2W] [W, 2W] [mlp, 2W] [W, 2 mlp]; NULL: split by the forward call       */
    const hirest_bf16 *wqkvT2, *woT2, *w1T2, *w2T2;       /* likewise their transposes for the backward's dX products (hirest_split2_transposed_bf16):
                                                             [W, 6W] [W, 2W] [W, 2 mlp] [mlp, 2W]; NULL: split by the backward call                  */
    void* ws; size_t ws_bytes;                            /* split-form scratch of the fp32 GEMMs on `stream` (hirest_gemm_f32_workspace_bytes) */
} hirest_train_block;
typedef struct hirest_train_block_grads {
    uint64_t struct_size;
    const float* dout;                                    /* d loss / d out [R, W]                                              */
    float* dx;                                            /* d loss / d x   [R, W]                                              */
    float *g_wqkv, *g_wo, *g_w1, *g_w2;                   /* [3W, W] [W, W] [mlp, W] [W, mlp]                                   */
    float *g_bqkv, *g_bo, *g_b1, *g_b2, *g_ln1_g, *g_ln1_b, *g_ln2_g, *g_ln2_b;
    hirest_colsum_item* items; int32_t* n_items; int32_t max_items, reserved;
    void* side_stream; void* side_ws; size_t side_ws_bytes;
    void* scratch; size_t scratch_bytes;
} hirest_train_block_grads;
/* The part of the training step's backward below the encoder blocks — VisualModel embeddings (module_visual.py:56-81) and the fusion of
 * modeling.py:155-195: LayerNorm / dropout backward, position rows, word_embeddings, mask / boundary embeddings, v * tn, the temporal MLP,
 * the ASR projection, normalize_video, clip_g_map, clip_g_map_text — issued by one C call (hirest_train_block's conventions: dX products on
 * `stream`, dW products on `side_stream`, column sums appended to `items`, `scratch` alive until both have executed; bit-identical to the
 * per-kernel calls).  g_pos [max_pos, W] is zero-filled here.  asr_dim = 0: no ASR branch; boundary = 0: no boundary embedding. */
typedef struct hirest_train_fusion_bwd {
    uint64_t struct_size;
    int32_t B, T, E, W, vis_dim, text_dim, asr_dim, boundary, max_pos, reserved;
    float drop; uint32_t seed_emb;
    const float *w_emb, *emb_ln_g, *t2_w, *asr1_w, *asr0_g, *norm_g;             /* [W,E] [W] [E,E] [E,asr_dim] [asr_dim] [E]                  */
    const float *x0, *f, *v, *tn, *tin, *a0, *asr2, *v0, *vis2, *t, *text;       /* kept by the forward                                         */
    const int32_t *mm32, *bm32, *n_valid;
    const float* dx;                                                             /* d loss / d (embedding output) [R, W]                        */
    float *g_emb_ln_g, *g_emb_ln_b, *g_pos, *g_w_emb, *g_b_emb, *g_mask, *g_bound, *g_t2_w, *g_t2_b, *g_t0_w, *g_t0_b,
          *g_asr1_w, *g_asr1_b, *g_asr0_g, *g_asr0_b, *g_norm_g, *g_norm_b, *g_vis_w, *g_vis_b, *g_text_w, *g_text_b;
    hirest_colsum_item* items; int32_t* n_items; int32_t max_items, reserved2;
    void* ws; size_t ws_bytes;
    void* side_stream; void* side_ws; size_t side_ws_bytes;
    void* scratch; size_t scratch_bytes;
} hirest_train_fusion_bwd;
size_t hirest_train_fusion_backward_scratch_bytes(const hirest_train_fusion_bwd* f);
int hirest_train_fusion_backward(const hirest_train_fusion_bwd* f, void* stream);
size_t hirest_train_block_forward_scratch_bytes(const hirest_train_block* b);
int hirest_train_block_forward(const hirest_train_block* b, void* scratch, size_t scratch_bytes, void* stream);
size_t hirest_train_block_backward_scratch_bytes(const hirest_train_block* b);
int hirest_train_block_backward(const hirest_train_block* b, const hirest_train_block_grads* g, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIREST_HIP_H */
