/*
 * pafuse_hip.h - C ABI of the MI355X (gfx950) implementation of the PAFUSE hot path:
 * the D3DP DDIM denoising loop over three per-body-part MixSTE spatio-temporal transformers.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream (as void*); no torch types.
 * All tensors are dense row-major fp32 in device memory unless stated; "t" vectors are int64.
 * Return value: 0 on success, a negative PAFUSE_E_* code otherwise (pafuse_last_error() has the text).
 * Nothing here allocates, synchronises or touches the default stream: launches go to `stream` only, scratch
 * comes from the caller's workspace, so every call can be captured into a hipGraph.
 *
 * Reference interface each entry point replaces (paths relative to valeoai/PAFUSE):
 *   pafuse_linear            torch.nn.Linear (+ nn.GELU) as used in common/mixste.py:30-43,54,57
 *   pafuse_layernorm         torch.nn.LayerNorm as used in common/mixste.py:96,101,203-204,208
 *   pafuse_attention         Attention.forward between qkv and proj, common/mixste.py:65-79
 *   pafuse_block_forward     Block.forward, common/mixste.py:113-116
 *   pafuse_attention_backward, pafuse_linear_weight_grad   the backward of the two above as torch autograd derives it
 *                            (unit entries of the training step's kernels)
 *   pafuse_time_embed        MixSTE2.time_mlp, common/mixste.py:127-139,179-184
 *   pafuse_mixste2_forward   MixSTE2.forward (is_train=False), common/mixste.py:278-298
 *   pafuse_d3dp_sample       D3DP.ddim_sample_flip / ddim_sample, common/diffusionpose.py:227-316
 *                            (with model_predictions[_fliping] :174-225, pred_parts/split_data :163-172,328-335)
 *   pafuse_embed             the input stage of model_predictions_fliping + split_data + the first lines of
 *                            MixSTE2.STE_forward: common/diffusionpose.py:193-198,328-335, common/mixste.py:227-235
 *   pafuse_ddim_finalize     the output stage of a DDIM step: part concat + un-flip + TTA mean + clamp
 *                            (common/diffusionpose.py:165-171,211-218), epsilon in fp64 (:157-161,222-223), update (:302-312)
 *   pafuse_hypothesis_errors the per-joint part of evaluate()'s aggregation, main_h3wb.py:327-362
 *   pafuse_mixste2_train_*   MixSTE2.forward (is_train=True) + autograd backward, common/mixste.py:215-225,260-298
 *   pafuse_d3dp_qsample      D3DP.prepare_diffusion_concat / q_sample, common/diffusionpose.py:319-326,358-374
 */
#ifndef PAFUSE_HIP_H
#define PAFUSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAFUSE_OK 0
#define PAFUSE_E_ARG (-1)       /* null pointer / bad size */
#define PAFUSE_E_SHAPE (-2)     /* a width this build has no kernel for */
#define PAFUSE_E_WORKSPACE (-3) /* workspace too small */
#define PAFUSE_E_HIP (-4)       /* a HIP launch failed */
#define PAFUSE_E_RANGE (-5)     /* pafuse_d3dp_check_range: a denoiser output was not finite ('f16x2': an activation left the fp16 range) */

#define PAFUSE_MAX_DEPTH 16
#define PAFUSE_MAX_PARTS 4

/* Parameters of one transformer block; names follow the reference state dict
 * ({STE,TTE}blocks.<i>.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.{weight,bias}). Linear weights are
 * [out,in] row-major exactly as torch stores them - nothing is re-packed. */
typedef struct pafuse_block_weights {
    const float *norm1_w, *norm1_b; /* [C] */
    const float *qkv_w, *qkv_b;     /* [3C,C], [3C] */
    const float *proj_w, *proj_b;   /* [C,C], [C] */
    const float *norm2_w, *norm2_b; /* [C] */
    const float *fc1_w, *fc1_b;     /* [H,C], [H]   (H = mlp hidden width, 2C by default) */
    const float *fc2_w, *fc2_b;     /* [C,H], [C] */
    /* split-precision modes only (pafuse_mixste2_weights.operand_bf16 == 2 or 3): the pre-split images of the four linear
     * weights, made by pafuse_split_weights from the fp32 tensors above with layout 2 (qkv, fc1), 1 (proj, fc2) -
     * mode 3: with PAFUSE_SPLIT_F16X2 - (a cache - remake after a weight changes); NULL otherwise */
    const void *qkv_ws, *proj_ws, *fc1_ws, *fc2_ws;
    /* split-precision modes, optional: LayerNorm folded into the GEMM that consumes it (qkv_lt and fc1_lt set, in every block
     * of a denoiser, or neither).  With them set, qkv_ws / fc1_ws must be the images of W (.) g - the weight scaled along its
     * input axis by norm1 / norm2's weight g - and
     *   qkv_lt[n] = sum_k beta_k W_nk + b_n     (norm1 -> qkv; fc1_lt likewise with norm2)
     * formed in fp64 and rounded once.  The whole-row kernels then store the residual stream CENTRED on its row mean
     * (x - mean(x): every reader of the stream is a LayerNorm or the residual add that feeds one, and LayerNorm does not see
     * a row's mean) and emit the row's (mean, rstd) instead of the normalised row; qkv / fc1 read the centred row and apply
     * rstd acc + lt in their epilogue: the same function as LN(x) W^T + b (common/mixste.py:113-116) with one [M,C] store
     * and one normalise pass less per whole-row launch, and nothing that cancels on rows whose mean is large against their
     * spread.  qkv_ls / fc1_ls (sum_k g_k W_nk: round 3's uncentred form) are not read any more; leave them NULL.
     * pafuse_block_forward ignores the fold (it is handed a normalised-input-free block). */
    const float *qkv_ls, *qkv_lt, *fc1_ls, *fc1_lt;
    /* split-precision mode, optional: the qkv projection and the attention of a block in ONE kernel (fqa_kernel: one
     * workgroup = whole sequences x one head; q, k, v never reach memory).  qkv_hs is the layout-2 image of the HEAD-MAJOR
     * qkv weight: for head h = 0 .. heads-1 the d rows of q_h, then zero rows up to DP, then k_h and v_h likewise
     * (DP = 32 for head dim d <= 32, else 48; [heads * 3 * DP, C] in all) - of W (.) g when the LayerNorm is folded;
     * qkv_hb [heads * 3 * DP] is the bias (folded: qkv_lt) in that row order (qkv_hl: round 3's uncentred term, not read - leave it NULL), zeros in the
     * padding.  Used when set and the sequence length has a fused form (L <= 48; in f16x2 mode L <= 80 at head dim <= 32:
     * the 68 joints of the face); otherwise, and for
     * pafuse_block_forward, the block runs qkv GEMM + attention from qkv_ws.  Same arithmetic per product and the same
     * attention as the two kernels. */
    const void *qkv_hs;
    const float *qkv_hb, *qkv_hl;
    /* f16x2 mode with the LayerNorm folded, optional: fc1 -> GELU -> fc2 (+ residual, + the LayerNorms behind it) of a block in
     * ONE kernel (hmlp_kernel: the hidden activations [M, 2C] stay in registers, common/mixste.py:37-43,115).  fc2_hp is the
     * PAFUSE_SPLIT_F16X2 image of fc2.weight with its columns (hidden units) re-ordered inside every group of 16:
     *   column 16 b + 8 h + i of the image = column 16 b + 8 (i >> 2) + 4 h + (i & 3) of the weight   (h = 0, 1; i = 0 .. 7)
     * - the order in which the first layer's accumulators hold a token's hidden units.  Used when set, the channel width is
     * 224 / 256 / 384, the hidden width 2C and the residual stream is kept as its H image only (keep_f32_residual == 0);
     * otherwise the block runs the two launches from fc1_ws / fc2_ws.  Same products, same K order between 16-deep steps. */
    const void *fc2_hp;
} pafuse_block_weights;

/* One MixSTE2 (common/mixste.py:141-210): F frames, J joints of this part, C channels, `depth` spatial +
 * `depth` temporal blocks, `heads` heads (C % heads == 0), mlp hidden = 2C unless mlp_hidden says otherwise. */
typedef struct pafuse_mixste2_weights {
    int32_t frames, joints, channels, depth, heads, in_chans; /* in_chans must be 5 (2-D + 3-D) */
    int32_t operand_bf16; /* matrix-product mode of the linear layers (LayerNorm, softmax and the attention arithmetic are fp32 in
                             every mode, and so is everything in memory except the operand images of modes 3 and 4;
                             pafuse_amd.D3DP's default is 2):
                             0: fp32-input matrix cores (v_mfma_f32_32x32x2_f32): a k-ordered fp32 FMA chain.
                             2: split precision "bf16x3": every fp32 operand is the exact sum of three bf16 slices,
                                products keep the six terms above 2^-24 relative on the bf16 matrix cores with fp32
                                accumulation - fp32-equivalent results (closer to exact arithmetic than the FMA
                                chain) at 2.7x the matrix rate.  Inference entry points need the *_ws weight images;
                                the training entry points do not (weights change every step: each pass makes the images
                                of its weights itself, in one launch) and run every GEMM of the step this way.
                             3: split precision "f16x2" (inference only): activations as two fp16 slices, weights as two
                                stored + one derived (PAFUSE_SPLIT_F16X2 images in *_ws), three products per k on the fp16
                                matrix cores - fp32-equivalent results at 5.3x the fp32 matrix rate.
                             4: split precision "bf16x3" on the image pipeline (round 5, inference only, opt-in: it measures
                                85 hypotheses/s against mode 2's 100): the arithmetic of mode 2 - every operand the exact sum of three bf16 slices,
                                the six products above 2^-24 relative, fp32 accumulation - with both GEMM operands as pre-split
                                "X images" (PAFUSE_SPLIT_X images in *_ws / qkv_hs; activations split once by their producer),
                                qkv + attention fused in every block (qkv_hs, qkv_hb required), and - with the LayerNorm
                                folded - the residual stream kept only as the X image of x - mean(row): exactly that fp32 row.
                                Widths 224 / 256 / 384, sequence lengths <= 80 (head dim <= 32) / <= 32 (head dim <= 48).
                             1: opt-in reduced precision - operands rounded to ONE bf16 (RNE), fp32 accumulate
                                (BASELINE configs[1]; inference only) */
    int32_t mlp_hidden;   /* MixSTE2(mlp_ratio=...): hidden width int(C * mlp_ratio) of every block's MLP, a multiple of 32,
                             at most 3C; 0 = 2C (PAFUSE: mlp_ratio = 2, common/diffusionpose.py:146).  Inference only. */
    float qk_scale;       /* MixSTE2(qk_scale=...): attention logit scale; 0 = head_dim^-0.5 (common/mixste.py:52).
                             Inference only.  (qkv_bias=False: point qkv_b at zeros.) */
    int32_t keep_f32_residual; /* operand_bf16 == 3 with the LayerNorm folded: 0 (default) keeps the residual stream between the
                             blocks in memory ONLY as its H image - of x - mean(row): every reader of the stream is a LayerNorm or
                             the residual add that feeds one, so a row's mean never reaches the output and is not carried - as
                             hi + 2^-11 lo (22-23 significant bits); the whole-row kernels read it as the residual and write it
                             back in place (a quarter of their HBM traffic less); 1 also keeps the fp32 rows and adds those. */
    const float *patch_w, *patch_b;                           /* Spatial_patch_to_embedding [C,5], [C] */
    const float *pos_spatial;                                 /* Spatial_pos_embed [J,C] */
    const float *pos_temporal;                                /* Temporal_pos_embed [F,C] */
    const float *tm1_w, *tm1_b, *tm3_w, *tm3_b;               /* time_mlp.1 [2C,C],[2C]; time_mlp.3 [C,2C],[C] */
    const float *freqs;                                       /* [C/2] sinusoid frequencies (mixste.py:134-136) */
    const float *snorm_w, *snorm_b, *tnorm_w, *tnorm_b;       /* Spatial_norm, Temporal_norm (eps 1e-6) */
    const float *hnorm_w, *hnorm_b;                           /* head.0 LayerNorm (eps 1e-5) */
    const float *head_w, *head_b;                             /* head.1 [3,C], [3] */
    pafuse_block_weights ste[PAFUSE_MAX_DEPTH];
    pafuse_block_weights tte[PAFUSE_MAX_DEPTH];
} pafuse_mixste2_weights;

/* Everything D3DP's sampler needs besides the tensors (common/diffusionpose.py:59-155). */
typedef struct pafuse_d3dp_config {
    int32_t num_parts;                 /* 3: body, face, hands (iteration order = concat order) */
    int32_t num_kps;                   /* 134 */
    int32_t frames;                    /* 27 */
    int32_t flip;                      /* 1: ddim_sample_flip, 0: ddim_sample */
    double scale;                      /* args.ft2d.scale (a Python float in the reference: the clamp bound is
                                          (float)(1.1 * scale) formed in fp64, multiplier / divisor are (float)scale,
                                          exactly as torch demotes Python scalars against fp32 tensors) */
    pafuse_mixste2_weights part[PAFUSE_MAX_PARTS];
    const int32_t *part_joints[PAFUSE_MAX_PARTS]; /* device: joint indices of each part (split_data) */
    const int32_t *joint_part;         /* device [num_kps]: part id of every joint */
    const int32_t *joint_local;        /* device [num_kps]: index of the joint inside its part */
    const int32_t *flip_perm;          /* device [num_kps]: flipped[j] = orig[perm[j]] (diffusionpose.py:197-198) */
    int32_t part_by_part_launches;     /* launch-schedule option of the single-stream bf16x3 schedule (no aux streams): 0 (default) runs
                                          the same layer of the independent body-part denoisers in ONE grid (grouped_*_kernel), 1 launches
                                          every layer part by part.  A tile's arithmetic does not depend on the grid it runs in, so both
                                          give the same bits (tests/test_hip_fullsize.py); it exists for that test and for A/B timing.
                                          Per call: the library keeps no process-wide schedule state and reads no environment variable. */
} pafuse_d3dp_config;

/* Per-step scalars of the loop, computed by the host exactly as the reference does in fp64
 * (common/diffusionpose.py:157-161, 302-306). */
typedef struct pafuse_ddim_step {
    int64_t time;                /* timestep fed to the denoisers */
    int32_t last;                /* 1 when time_next < 0: img = x_start, no update */
    double sqrt_recip_acp;       /* sqrt_recip_alphas_cumprod[time] */
    double sqrt_recipm1_acp;     /* sqrt_recipm1_alphas_cumprod[time] */
    double sqrt_alpha_next, c, sigma;
} pafuse_ddim_step;

const char *pafuse_version(void);
const char *pafuse_last_error(void);
/* Layout version of the structs above (bumped whenever a field is added, moved or removed: this header has no size fields).
 * A caller built against another header gets shifted pointers, not an error - compare with PAFUSE_ABI_VERSION at load time
 * (pafuse_amd/_lib.py does; tests/cabi/linear_smoke.c shows the C side). */
#define PAFUSE_ABI_VERSION 6
int pafuse_abi_version(void);

/* out[M,N] = act(A[M,K] @ W[N,K]^T + bias), act: 0 none, 1 exact-erf GELU; +2: bf16 operands (see operand_bf16).
 * K,N multiples of 32. */
int pafuse_linear(const float *A, const float *W, const float *bias, float *out, int64_t M, int32_t N, int32_t K,
                  int32_t act, void *stream);

/* Split-precision weight image: W [N,K] fp32 (K % 32 == 0) -> `out`, pafuse_split_weights_bytes(N, K) = 6*N*K bytes
 * ([K/c][N][6c B] with c = 32 or 16: per row and K chunk, sub-blocks of 8 k x 3 bf16 slices, laid out as the kernels'
 * LDS image).  `layout` names the layer the image is for - it decides c and the rotation of the sub-blocks inside a row:
 *   0  pafuse_linear_split without PAFUSE_LINEAR_QKV_IMAGE: the 32x32x16-MFMA plain kernel (mlp.fc1 until ABI 5)
 *   1  attn.proj, mlp.fc2: the whole-row kernels (c = 16 at the widths that have an LDS-DMA tile)
 *   2  attn.qkv, mlp.fc1: the 16x16x32-MFMA kernels (pafuse_block_weights.qkv_ws and - since ABI 6 - fc1_ws must be made with layout 2)
 * pafuse_linear_split is pafuse_linear on such an image (act: 0 none, 1 GELU, + PAFUSE_LINEAR_QKV_IMAGE when the image
 * has layout 2): the unit entry of the split-precision products, which replace the same nn.Linear call sites
 * (common/mixste.py:38-42,65,80). */
#define PAFUSE_LINEAR_QKV_IMAGE 2
size_t pafuse_split_weights_bytes(int64_t N, int64_t K);
/* the bytes pafuse_split_weights writes for `layout` (the layout bits + PAFUSE_SPLIT_F16X2 / PAFUSE_SPLIT_X): 6 N K for the
 * bf16x3 images, 4 N K + 256 for the f16x2 H image; pafuse_split_weights_bytes (6 N K) is large enough for every layout */
size_t pafuse_split_image_bytes(int64_t N, int64_t K, int32_t layout);
int pafuse_split_weights(const float *W, int32_t N, int32_t K, int32_t layout, void *out, void *stream);
int pafuse_linear_split(const float *A, const void *Wsplit, const float *bias, float *out, int64_t M, int32_t N,
                        int32_t K, int32_t act, void *stream);

/* Second split-precision scheme, "f16x2" (operand_bf16 == 3, round 4): THREE fp16 MFMA products per fp32-equivalent product.
 * Both GEMM operands are "H images": rows of 4 K bytes, per sub-block of 8 k the 16 bytes of the first fp16 slice, then the
 * 16 of the second - [rows][K/8][hi 8 x f16 | lo 8 x f16]:
 *   activation a   hi = f16(a), lo = f16((a - hi) * 2^11); written in this form by the kernel that PRODUCES the tensor
 *                  (whole-row epilogues, attention, the fc1 epilogue, the embedding) - same bytes as fp32, split once;
 *   weight W       pafuse_split_weights(W, N, K, PAFUSE_SPLIT_F16X2, out, stream): the image of Ws = 2^k W (the tensor's
 *                  largest |Ws| in [2^14, 2^15)): w0 = f16(Ws), w1 = f16(Ws - w0); 4 N K bytes + a 256-byte tail whose
 *                  first float is 2^-k (a buffer of pafuse_split_weights_bytes(N, K) bytes is large enough); every layer
 *                  uses this one geometry, whatever the value of the layout bits.
 * Kept per product: hi w0 + hi w1 + lo (w0 2^-11), fp32 accumulate, the accumulator times 2^-k in the epilogue; dropped
 * terms and slice roundings ~ 2^-22 relative.  fp32-equivalent: the same bounds against exact arithmetic as bf16x3 hold
 * (tests/test_hip_parity.py).  Activations must stay below 65504 in magnitude (beyond it hi is inf and the affected outputs
 * are inf / NaN, never silently wrong).  Inference only; channel widths 224 / 256 / 384, plain-layer widths that are
 * multiples of 128 or 224.
 * pafuse_hsplit_rows: X [R,K] fp32 -> its activation H image (4 R K bytes).  pafuse_linear_h: nn.Linear on H images
 * (act: 0 none, 1 GELU): fp32 `out` [M,N], or - when out_h is given - the H image of the output instead (either may be
 * NULL, not both).  N a multiple of 128 or 224, K of 32.  Replaces the same nn.Linear call sites (common/mixste.py:38-42,65,80). */
#define PAFUSE_SPLIT_F16X2 4

/* The bf16x3 scheme on the image pipeline (operand_bf16 == 4, round 5): the products of mode 2, both operands as "X images" -
 * rows of 6 K bytes (K % 32 == 0), per chunk of 32 k the 32 bf16 of slice 0, then of slice 1, then of slice 2:
 * [rows][K/32][3][32 x bf16].  x = s0 + s1 + s2 exactly (s0 = bf16(x), s1 = bf16(x - s0), s2 = bf16(x - s0 - s1), RNE, the
 * subtractions exact), so an X image IS the fp32 tensor, bit for bit; no scaling, no range limit beyond fp32's, inf / NaN stay
 * inf / NaN (in every slice).
 *   weight W      pafuse_split_weights(W, N, K, PAFUSE_SPLIT_X, out, stream): 6 N K bytes = pafuse_split_weights_bytes(N, K);
 *                 one geometry for every layer, whatever the value of the layout bits;
 *   activation    written in this form by the kernel that PRODUCES the tensor; pafuse_xsplit_rows: X [R,K] fp32 -> its X image
 *                 (unit tests, pafuse_block_forward).
 * pafuse_linear_x: nn.Linear on X images (act: 0 none, 1 GELU): fp32 `out` [M,N], or - when out_x is given - the X image of the
 * output instead.  N a multiple of 128, 224 or 96, K of 32.  Replaces the same nn.Linear call sites (common/mixste.py:38-42,65,80). */
#define PAFUSE_SPLIT_X 8
int pafuse_xsplit_rows(const float *X, int64_t R, int32_t K, void *out, void *stream);
int pafuse_linear_x(const void *Ax, const void *Wx, const float *bias, float *out, void *out_x, int64_t M, int32_t N, int32_t K,
                    int32_t act, void *stream);

int pafuse_hsplit_rows(const float *X, int64_t R, int32_t K, void *out, void *stream);
int pafuse_linear_h(const void *Ah, const void *Wh, const float *bias, float *out, void *out_h, int64_t M, int32_t N, int32_t K,
                    int32_t act, void *stream);

/* qkv projection + attention of whole sequences in ONE kernel on operand images - the unit entry of hfqa_kernel (scheme
 * PAFUSE_SPLIT_F16X2: H images) and xfqa_kernel (PAFUSE_SPLIT_X: X images); common/mixste.py:65-79:
 *   (q | k | v)[m] = rstd_m * (x[m] W^T) + b   (rstd_m = stats[2 m + 1], stats NULL: 1 - with stats the image is the CENTRED row and
 *   W is W (.) g, b = W beta + b: the folded LayerNorm);   o[sequence, head] = softmax(q k^T * scale) v
 * x_img: the image of x [M,C]; qkv_hs / qkv_hb: the head-major weight image [heads * 3 * DP, C] and bias of
 * pafuse_block_weights.qkv_hs / qkv_hb; o_img: the image of o [M,C] (rows no sequence touches are left alone).  Sequence map as
 * pafuse_attention; qk_scale 0 = head_dim^-0.5.  Shapes with a fused form only (sequence length <= 80 at head dim <= 32, <= 32
 * at head dim <= 48; PAFUSE_SPLIT_X at head dim > 32: an even number of heads). */
int pafuse_qkv_attention_image(int32_t scheme, const void *x_img, const float *stats, const void *qkv_hs, const float *qkv_hb,
                               void *o_img, int64_t M, int64_t nseq, int32_t L, int32_t C, int32_t heads, int64_t group,
                               int64_t group_stride, int64_t seq_stride, int64_t tok_stride, float qk_scale, void *stream);

/* The MLP of a block on H images in one kernel (the unit entry of hmlp_kernel; common/mixste.py:37-43,115 with norm2 folded):
 *   y = xc + GELU(rstd * (xc W1^T) + bias1) W2^T + bias2,   xc = the rows of `xh` (the CENTRED image of the residual stream),
 *   rstd = stats_in[2 m + 1] (NULL: 1);  out_xh = the H image of y - mean(y),  stats_out[m] = (mean(y), 1 / sqrt(var(y) + eps)).
 * W1h: the PAFUSE_SPLIT_F16X2 image of fc1.weight [2C, C] (of W1 (.) g for a folded LayerNorm, bias1 = W1 beta + b1);
 * W2hp: the image of fc2.weight [C, 2C] in the column order of pafuse_block_weights.fc2_hp.  C = 224, 256 or 384.
 * out_xh may be xh itself (a workgroup owns its rows). */
int pafuse_mlp_h(const void *xh, const float *stats_in, const void *W1h, const float *bias1, const void *W2hp, const float *bias2,
                 void *out_xh, float *stats_out, int64_t M, int32_t C, float eps, void *stream);

/* out[M,C] = LayerNorm(x[M,C]) * w + b  (biased variance, eps inside the sqrt). */
int pafuse_layernorm(const float *x, const float *w, const float *b, float *out, int64_t M, int32_t C, float eps,
                     void *stream);

/* o[M,C] = softmax(q k^T * d^-1/2) v per (sequence, head) on qkv[M,3C] laid out [.., 3, heads, d].
 * Sequence s (0..nseq) holds rows  (s / inner) * inner * L * tstride_outer ... described by:
 *   row(s, t) = (s / group) * group_stride + (s % group) * seq_stride + t * tok_stride
 * spatial: group=1, group_stride=L, seq_stride=0, tok_stride=1; temporal: group=J, group_stride=F*J,
 * seq_stride=1, tok_stride=J. */
int pafuse_attention(const float *qkv, float *o, int64_t nseq, int32_t L, int32_t C, int32_t heads, int64_t group,
                     int64_t group_stride, int64_t seq_stride, int64_t tok_stride, void *stream);

/* Unit entry points of the training step's backward kernels (round 5; the step itself is pafuse_mixste2_train_backward).
 * pafuse_attention_backward: dqkv[M,3C] from qkv[M,3C] and dO[M,C] - what autograd derives from common/mixste.py:66-79
 * (S = q k^T d^-1/2, P = softmax(S), O = P v:  dV = P^T dO, dP = dO V^T, dS = P (dP - rowsum(P dP)), dQ = dS K d^-1/2,
 * dK = dS^T Q d^-1/2), sequence map as pafuse_attention.  Rows no sequence touches are left alone.
 * pafuse_linear_weight_grad: dW[N,K] += dY[M,N]^T X[M,K] and, with db, db[N] += column sums of dY - the weight and bias
 * gradient of nn.Linear (common/mixste.py:30-43,54,57) as the training step forms them: operand_bf16 0 = fp32 matrix cores,
 * 2 = split (bf16x3) products on the large tiles where a PAFUSE width divides; partial sums in a fixed order (bit-reproducible).
 * workspace: pafuse_linear_weight_grad_bytes() bytes. */
int pafuse_attention_backward(const float *qkv, const float *d_o, float *dqkv, int64_t nseq, int32_t L, int32_t C, int32_t heads,
                              int64_t group, int64_t group_stride, int64_t seq_stride, int64_t tok_stride, void *stream);
size_t pafuse_linear_weight_grad_bytes(void);
int pafuse_linear_weight_grad(const float *dY, const float *X, float *dW, float *db, int64_t M, int32_t N, int32_t K,
                              int32_t operand_bf16, void *workspace, size_t workspace_bytes, void *stream);

/* x[S*L,C] <- Block(x) in place for S contiguous sequences of L tokens (Block.forward, eps 1e-6).  operand_bf16: the
 * matrix-product mode of the four linear layers (as pafuse_mixste2_weights.operand_bf16; 2, 3 and 4 need the *_ws images of
 * their scheme, 4 also qkv_hs / qkv_hb). */
size_t pafuse_block_workspace_bytes(int64_t rows, int32_t C);
int pafuse_block_forward(const pafuse_block_weights *w, float *x, int64_t S, int32_t L, int32_t C, int32_t heads,
                         int32_t operand_bf16, void *workspace, size_t workspace_bytes, void *stream);

/* temb[B,C] = time_mlp(t[B]);  hid_scratch: [B,2C] floats of device scratch */
int pafuse_time_embed(const pafuse_mixste2_weights *w, const int64_t *t, int32_t B, float *temb, float *hid_scratch,
                      void *stream);

/* 1 when the inference kernels of matrix-product mode `mode` (pafuse_mixste2_weights.operand_bf16) exist for a denoiser of
 * channel width C, MLP hidden width `hidden` (0 = 2C), `heads` heads and sequences of `joints` / `frames` tokens, else 0 -
 * callers pick mode 4 where it is supported and mode 2 elsewhere (the single-model variant, width 288). */
int pafuse_mode_supported(int32_t mode, int32_t C, int32_t hidden, int32_t heads, int32_t joints, int32_t frames);

/* How many of the 2 * depth blocks of `w` will run qkv + attention as the one fused kernel (pafuse_block_weights.qkv_hs
 * set and the block's sequence length / head dim have a fused form); negative = an error code.  Callers and tests use
 * it to know which path a configuration takes. */
int pafuse_mixste2_fused_blocks(const pafuse_mixste2_weights *w);

/* MixSTE2.forward, eval: x2d[B,F,J,2], x3d[B,P,F,J,3], t[B] -> out[B,P,F,J,3]. */
size_t pafuse_mixste2_workspace_bytes(const pafuse_mixste2_weights *w, int32_t B, int32_t P);
int pafuse_mixste2_forward(const pafuse_mixste2_weights *w, const float *x2d, const float *x3d, const int64_t *t,
                           int32_t B, int32_t P, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* The whole DDIM loop for B clips x P hypotheses and `nsteps` sampling steps.
 *   x2d, x2d_flip [B,F,J,2] (x2d_flip ignored when cfg->flip == 0)
 *   noise [n_draws,B,P,F,J,3]: draw 0 is the initial img, draw k the randn_like of the k-th update
 *         (n_draws >= 1 + number of steps with last == 0)
 *   out   [B,nsteps,P,F,J,3]: x_start of every step (torch.stack(preds_all, dim=1))
 * `aux_streams` (may be NULL / n_aux 0): extra HIP streams; a step's work is cut into (part, hypothesis-group) lanes,
 * one stream each, groups = (n_aux + 1) / parts (>= 1): 2 aux streams = the three parts side by side (fastest
 * measured; more lanes cost more in small launches than they gain in overlap).  Events are created and destroyed
 * inside the call only when aux streams are given.  The library must be built without packed-fp32 VALU instructions
 * for this (as __graft_entry__.build() does, -DPAFUSE_NO_PACKED_F32): beside v_mfma_f32_32x32x16_bf16 waves of another
 * queue a v_pk_*_f32 with a high-register src1 select returns wrong lanes on MI355X
 * (profiles/r03_bf16_mfma_concurrency.md).  A build without that flag ignores the aux streams when any part runs a
 * bf16-MFMA mode (operand_bf16 != 0) and pafuse_d3dp_lanes returns 1. */
size_t pafuse_d3dp_workspace_bytes(const pafuse_d3dp_config *cfg, int32_t B, int32_t P);
int pafuse_d3dp_sample(const pafuse_d3dp_config *cfg, const pafuse_ddim_step *steps, int32_t nsteps,
                       const float *x2d, const float *x2d_flip, const float *noise, int32_t n_draws, int32_t B,
                       int32_t P, float *out, void *workspace, size_t workspace_bytes, void *stream,
                       void *const *aux_streams, int32_t n_aux);

/* Loud failure through the loop.  pafuse_d3dp_sample keeps one int32 in its workspace (byte offset pafuse_d3dp_range_flag_offset,
 * zeroed at the start of every call) that the output stage of every step sets when a denoiser prediction is not finite.  The
 * values themselves flow on as NaN - the clamps of the loop keep a NaN a NaN, exactly as torch.clamp does
 * (common/diffusionpose.py:193,216-217) - so an 'f32' / 'bf16x3' run returns what the reference returns.  In 'f16x2' a non-finite
 * prediction is how an activation beyond the fp16 range shows (|a| >= 65504: its high slice is inf): callers of that mode ask
 * pafuse_d3dp_check_range after the call - the ONE entry point of this library that synchronises (`stream`) - and get
 * PAFUSE_E_RANGE; pafuse_amd.D3DP raises.  (Inside a stream capture, read the word yourself after the replay.) */
size_t pafuse_d3dp_range_flag_offset(const pafuse_d3dp_config *cfg, int32_t B, int32_t P);
int pafuse_d3dp_check_range(const pafuse_d3dp_config *cfg, int32_t B, int32_t P, const void *workspace, void *stream);

/* Number of distinct HIP streams pafuse_d3dp_sample will launch on when handed `n_aux` aux streams (the caller's
 * stream + the first (result - 1) aux streams).  A caller that forks its aux streams into a stream capture must fork
 * exactly those: the library joins the streams it used, an extra forked stream would leave the capture unjoined. */
int pafuse_d3dp_lanes(const pafuse_d3dp_config *cfg, int32_t B, int32_t P, int32_t n_aux);

/* ---- the two index-carrying stages of a DDIM step as unit entry points (SURVEY.md 8b "finer ops") ---------------
 * pafuse_embed: input stage of one body part.  For every row (fl, b, p, f, j) of the part's token matrix
 * (fl < nflip; row = (((fl*B + b)*P + p)*F + f)*J + j):
 *   jj = joints ? joints[j] : j                      split_data's gather            diffusionpose.py:328-335
 *   v3 = x3d[b, p, f, fl ? perm[jj] : jj, :]         L/R swap of the flipped copy   diffusionpose.py:197-198
 *   if do_clamp: v3 = clamp(v3, +-1.1*scale) / scale                                diffusionpose.py:193-194
 *   if fl: v3.x = -v3.x                                                             diffusionpose.py:196
 *   v2 = (fl ? x2d_flip : x2d)[b, f, jj, :]          2-D input broadcast over P     mixste.py:228-229
 *   x[row]  = patch_w [v2, v3] + patch_b + pos_spatial[j] + temb[b]                 mixste.py:230-235
 *   xn[row] = LayerNorm(x[row]; norm_w, norm_b, norm_eps)                           (norm1 of the first block)
 * x3d [B,P,F,J3,3], x2d/x2d_flip [B,F,J3,2], joints int32 [J] (values < J3; NULL = identity, needs J3 == J),
 * perm int32 [J3] (needed when nflip == 2), patch_w [C,5], patch_b [C], pos_spatial [J,C], temb [B,C],
 * x / xn [nflip*B*P*F*J, C].  C % 4 == 0, C <= 384.  The index tables live on the device and are trusted. */
int pafuse_embed(const float *x3d, const float *x2d, const float *x2d_flip, const int32_t *joints, const int32_t *perm,
                 const float *patch_w, const float *patch_b, const float *pos_spatial, const float *temb,
                 const float *norm_w, const float *norm_b, float norm_eps, int32_t B, int32_t P, int32_t F, int32_t J,
                 int32_t J3, int32_t C, int32_t nflip, int32_t do_clamp, double scale, float *x, float *xn, void *stream);

/* pafuse_ddim_finalize: output stage of DDIM step `step` of T.  pred[i] = part i's denoiser output
 * [nflip*B*P*F*part_joints[i], 3] (host array of `num_parts` device pointers; flipped half second).  Per (b,p,f,j):
 *   x0 = pred[joint_part[j]][.., joint_local[j], :]                       torch.cat(dim=-2)   diffusionpose.py:171
 *   if flip: u = flipped prediction of joint flip_perm[j], u.x = -u.x ; x0 = (x0 + u) / 2     :211-215
 *   x0 = clamp(x0 * scale, +-1.1*scale) -> out[b, step, p, f, j, :]                            :216-218, :298
 *   last step: img = x0.  Otherwise eps = (sqrt_recip_acp*img - x0) / sqrt_recipm1_acp in fp64 (:157-161) and
 *   img = x0*sqrt_alpha_next + c*eps + sigma*noise (:308-312; scalar demotion as torch does it per sampler).
 * img [B,P,F,J,3] in/out, noise [B,P,F,J,3] (may be NULL on the last step), out [B,T,P,F,J,3];
 * joint_part / joint_local / flip_perm int32 [J] on the device (trusted). */
int pafuse_ddim_finalize(const float *const *pred, const int32_t *part_joints, int32_t num_parts,
                         const int32_t *joint_part, const int32_t *joint_local, const int32_t *flip_perm, float *img,
                         const float *noise, float *out, int32_t B, int32_t P, int32_t F, int32_t J, int32_t T,
                         int32_t step, int32_t flip, double scale, const pafuse_ddim_step *st, void *stream);

/* Hypothesis aggregation on the gathered predictions - the caller-side step of main_h3wb.py:327-362
 * (wb_pose_from_parts common/utils.py:113-126, project_to_2d common/camera.py:30-60, the per-joint parts of
 * mpjpe_diffusion* common/loss.py:36-168).  pred [B,T,P,F,J,3] and gt [B,F,J,3] are part-centred; x2d [B,F,J,2];
 * traj [B,F,3]; cam [9]; conn/pbroot int32 [J] (connection joint / part root of every joint).
 * Outputs: e3, epb [B,T,P,F,J] (per-hypothesis whole-body / part-re-centred errors), jbest, pagg, jagg,
 * paggpb [B,T,F,J].  The means over (b,f,j) and the P-Best argmin are left to the caller (a few tiny reductions). */
int pafuse_hypothesis_errors(const float *pred, const float *gt, const float *x2d, const float *traj, const float *cam,
                             const int32_t *conn, const int32_t *pbroot, int32_t B, int32_t T, int32_t P, int32_t F,
                             int32_t J, float *e3, float *epb, float *jbest, float *pagg, float *jagg, float *paggpb,
                             void *stream);

/* Replays the GEMM launches of one flip-TTA denoiser pass (all three parts) back to back on `stream`, for
 * bench.py's per-kernel roofline measurement: returns the number of launches, adds their algorithmic FLOPs to
 * *flops.  Uses (and overwrites) the workspace like pafuse_d3dp_sample does. */
int pafuse_d3dp_replay_gemms(const pafuse_d3dp_config *cfg, int32_t B, int32_t P, void *workspace,
                             size_t workspace_bytes, void *stream, double *flops);
/* The same for a subset of the four linear layers of every block: layer_mask bits 1 = qkv, 2 = proj, 4 = fc1 (+GELU),
 * 8 = fc2 (15 = pafuse_d3dp_replay_gemms).  bench.py times each layer kind alone for the per-kernel roofline lines. */
int pafuse_d3dp_replay_layers(const pafuse_d3dp_config *cfg, int32_t B, int32_t P, void *workspace,
                              size_t workspace_bytes, void *stream, int32_t layer_mask, double *flops);

/* ---- Training: MixSTE2 in train mode and its backward -------------------------------------------
 * Replaces MixSTE2.forward with is_train=True (common/mixste.py:215-225,260-298: no hypothesis axis, DropPath on
 * both residual branches of every Block :113-116) and what autograd derives from it.
 *   x2d [B,F,J,2], x3d [B,F,J,3] (the noised target), t [B] -> out [B,F,J,3]
 *   drop_path: NULL (no DropPath) or float [2*depth][2][B*max(F,J)] - for block k in execution order (STE0, TTE0,
 *              STE1, ...) and branch (0 attention, 1 MLP) the factor mask/keep_prob of every sequence (spatial
 *              blocks: B*F sequences (b,f); temporal blocks: B*J sequences (b,j)), as timm's DropPath draws it.
 *   saved:     pafuse_mixste2_train_bytes(w, B) bytes; the forward leaves the activations the backward needs there.
 * Backward: dout [B,F,J,3] -> ADDS the gradient of every parameter into the buffer the like-named pointer of `grads`
 * addresses (same struct as the weights; `freqs` and the dimensions are ignored).  All row reductions are two-stage
 * in a fixed order: bit-reproducible, no atomics.  `side_stream` (may be NULL): a second HIP stream the weight-gradient
 * GEMMs run on next to the input-gradient chain (forked and joined with events inside the call; results identical).
 * w->operand_bf16: 0 = every GEMM on the fp32 matrix cores; 2 = split-precision products in the GEMMs of the step
 * (qkv, fc1, the whole-row proj / fc2, every dX and dW; fp32-equivalent, gradients within the same bounds); 1 is refused. */
size_t pafuse_mixste2_train_bytes(const pafuse_mixste2_weights *w, int32_t B);
int pafuse_mixste2_train_forward(const pafuse_mixste2_weights *w, const float *x2d, const float *x3d, const int64_t *t,
                                 int32_t B, const float *drop_path, float *out, void *saved, size_t saved_bytes,
                                 void *stream);
int pafuse_mixste2_train_backward(const pafuse_mixste2_weights *w, const pafuse_mixste2_weights *grads,
                                  const float *dout, int32_t B, const float *drop_path, void *saved, size_t saved_bytes,
                                  void *stream, void *side_stream);

/* D3DP.prepare_diffusion_concat + q_sample (common/diffusionpose.py:319-326,358-374) for B samples of per_sample
 * floats: out = clamp(sqrt_acp[t_b] * (x0*scale) + sqrt_1m_acp[t_b] * noise, +-1.1 scale) / scale, in fp64 as the
 * reference's fp64 buffers force, cast to fp32. */
int pafuse_d3dp_qsample(const float *x0, const float *noise, const int64_t *t, const double *sqrt_alphas_cumprod,
                        const double *sqrt_one_minus_alphas_cumprod, double scale, float *out, int32_t B,
                        int64_t per_sample, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PAFUSE_HIP_H */
