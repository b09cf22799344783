/*
 * b4c.h -- C ABI of libb4c_hip.so: the MI355X (gfx950) hot path of BERT4ClickPath's
 * BERT4Rec forward / Cloze-training step.
 *
 * The reference (MiladShahidi/BERT4ClickPath, pure Python on TensorFlow 2.3.1) has no
 * native layer; its hot path bottoms out in TF ops.  Each entry point below replaces the
 * TF call sites cited next to it (paths relative to the reference root).  A binding for
 * the reference's own Python is shown in INTEGRATION.md (ctypes).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory unless named h_*;
 *  - no allocation, no global state, no synchronisation inside: the caller owns every
 *    buffer and passes the hipStream_t (as void*); calls are thread-safe per stream;
 *  - return 0 on success, a negative B4C_E* code otherwise (b4c_last_error() gives text);
 *  - dtype: B4C_F32 = exact fp32 path (parity), B4C_BF16 = bf16 storage / fp32 accumulate
 *    (throughput).  "T" below means that element type.  Weights master copies, biases,
 *    LayerNorm params, statistics, losses and gradients of weights are always fp32;
 *  - ld* are row pitches in ELEMENTS.  Vector paths need pitches and inner sizes that are
 *    multiples of 8 elements; the host pads (zeros) where the model's sizes are not.
 */
#ifndef B4C_H
#define B4C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B4C_F32 0
#define B4C_BF16 1
#define B4C_I32 2    /* (b4c_poison_rows only: int32 rows, poisoned with -1) */

#define B4C_OK 0
#define B4C_EINVAL (-1)   /* bad argument (shape / alignment / dtype)              */
#define B4C_ELAUNCH (-2)  /* the HIP runtime refused the launch (see last_error)    */
#define B4C_EUNSUPPORTED (-3)

#define B4C_MAX_FEATURES 4
#define B4C_MAX_TOPK 16

#define B4C_ACT_NONE 0
#define B4C_ACT_RELU 1

#define B4C_CE_TF 0    /* clip[1e-7,1-1e-7] -> log -> log-softmax (tf.keras.backend, TF 2.3.1) */
#define B4C_CE_PLAIN 1 /* -log p_y                                                            */

int b4c_abi_version(void);
const char *b4c_last_error(void);

/* ---- R6 + R4: embedding stage ------------------------------------------------------
 * replaces transformer.py:376-398 (Embedding per feature -> concat -> * sqrt(d) -> + PE)
 * and create_padding_mask :38-41, plus Encoder.call's input dropout :263.
 *   out[t, off_f + j] = drop( table_f[ids_f[t], j] * scale + pe[s, off_f + j] )
 *   key_pad[t] = (ids_0[t] == 0)
 * h_ids / h_tables / h_dims / h_rows: HOST arrays of n_feat entries (device pointers inside).
 * ids are int64 (B*S); ids outside [0, rows) are clamped.  pe: fp32 [>=S][d_model].
 * The features are concatenated when their dims add up to d_model (the reference, :384-388).  SUMMED FEATURES (no reference
 * counterpart; BASELINE.json configs[3] words the two-feature input as "gather + sum"): n_feat >= 2 and EVERY h_dims[f] ==
 * d_model -- out[t, j] = drop( (sum_f table_f[ids_f[t], j]) * scale + pe[s, j] ), rows added in feature order in fp32, and
 * the backward entry points give every table the gradient of the one d_model-wide row.
 * dropout: keep(e) from the counter hash b4c_keep(seed, e), e = t*d_model + col; rate 0 = off. */
int b4c_embed_concat_pe_fwd(int n_feat, const int64_t *const *h_ids, const float *const *h_tables,
                            const int *h_dims, const int64_t *h_rows, const float *pe, float scale,
                            void *out, int ld_out, uint8_t *key_pad, int B, int S, int d_model,
                            float dropout_rate, uint64_t seed, int dtype, void *stream);

/* gradient of the above w.r.t. the tables (fp32, accumulated with float atomics into
 * h_dtables[f], which the caller zeroes): dtable_f[id, j] += scale * dropmask * dout[t, off_f+j] */
int b4c_embed_concat_pe_bwd(int n_feat, const int64_t *const *h_ids, float *const *h_dtables,
                            const int *h_dims, const int64_t *h_rows, float scale, const void *dout,
                            int ld_dout, int B, int S, int d_model, float dropout_rate, uint64_t seed,
                            int dtype, void *stream);
/* same, given for every feature the token indices sorted by id (order[f][p], int32, any order among equal ids):
 * runs of one id are summed in registers and written once -- two float atomics per distinct id and wave boundary
 * instead of one per token and column.  The sort is the caller's (one radix sort of B*S keys per feature). */
int b4c_embed_concat_pe_bwd_sorted(int n_feat, const int64_t *const *h_ids, const int32_t *const *h_order,
                                   float *const *h_dtables, const int *h_dims, const int64_t *h_rows, float scale,
                                   const void *dout, int ld_dout, int B, int S, int d_model, float dropout_rate,
                                   uint64_t seed, int dtype, void *stream);
/* (ABI 9) the same with a caller scratch: workspace != NULL selects the DETERMINISTIC form -- runs that cross the 64-entry ranges
 * of the kernel's waves are summed in range order through the scratch instead of meeting through float atomics, so two launches
 * on the same inputs give the same bits.  workspace == NULL: the form above. */
int64_t b4c_embed_concat_pe_bwd_sorted_workspace_bytes(int n_feat, const int *h_dims, int B, int S);
int b4c_embed_concat_pe_bwd_sorted_ws(int n_feat, const int64_t *const *h_ids, const int32_t *const *h_order,
                                      float *const *h_dtables, const int *h_dims, const int64_t *h_rows, float scale,
                                      const void *dout, int ld_dout, int B, int S, int d_model, float dropout_rate,
                                      uint64_t seed, void *workspace, int64_t workspace_bytes, int dtype, void *stream);

/* ---- dense layers (R8 projections, R9 FFN, R12 head) --------------------------------
 * replaces tf.keras.layers.Dense call sites transformer.py:112-116,165-166; head.py:35-36.
 * pack: fp32 Keras kernel [K][N] -> compute copy in T.
 *   transpose=1: dst[n][k] = src[k][n]  (dst is [N][ld_dst], forward operand)
 *   transpose=0: dst[k][n] = src[k][n]  (dst is [K][ld_dst], backward-dX operand)
 * Only the K x N valid elements are written: the caller zero-initialises padded buffers once
 * (pad rows / columns never change afterwards). */
int b4c_pack_weight(const float *src, int K, int N, void *dst, int ld_dst, int transpose, int dtype,
                    void *stream);

/* All layers' compute copies in one launch.  One descriptor per Keras kernel; fused layers (Q|K|V) use
 * several descriptors writing at different col_off of the same wt / wc / bias buffers.
 *   wt[(col_off + n)][k]  (pitch ld_t)   = src[k][n]      forward operand, may be NULL
 *   wc[k][col_off + n]    (pitch ld_c)   = src[k][n]      backward-dX operand, may be NULL
 *   bias_dst[col_off + n]                = bias_src[n]    fp32, may be NULL
 * d_desc is a DEVICE array; max_tiles = max over descriptors of ceil(K/32)*ceil(N/32). */
typedef struct {
    const float *src;
    const float *bias_src;
    void *wt;
    void *wc;
    float *bias_dst;
    int32_t K, N, ld_t, ld_c, col_off, _pad;
} b4c_pack_desc;
int b4c_pack_weights_batched(const b4c_pack_desc *d_desc, int n_desc, int max_tiles, int dtype, void *stream);
/* (max_tiles: the largest ceil(K / 64) * ceil(N / 64) over the descriptors -- the launch is max_tiles x n_desc workgroups) */

/* C[M][N] = epilogue( A[M][K] . Bt[N][K]^T )     (both operands K-contiguous)
 *   v = acc + bias[n]            (bias fp32 or NULL)
 *   v = relu(v)                  if act == B4C_ACT_RELU
 *   v = v * (gate[m][n] > 0)     if gate != NULL   (ReLU backward: gate = saved activation, pitch ldg)
 *   v = v + residual[m][n]       if residual != NULL (T, pitch ldr)
 * out_dtype chooses C's element type (T of `dtype`, or B4C_F32 for fp32 logits from bf16 inputs).
 * K % 8 == 0, lda/ldb % 8 == 0 (bf16) or % 4 (fp32). */
int b4c_gemm_nt(const void *A, int lda, const void *Bt, int ldb, void *C, int ldc, int M, int N, int K,
                const float *bias, int act, const void *gate, int ldg, const void *residual, int ldr,
                int dtype, int out_dtype, void *stream);

/* R9/R10 fused with the GEMM that feeds them (bf16, N <= 256; N > 128 runs 64-row x 256-column workgroups):
 *   y = A . Bt^T + bias;  z = x + dropout(y);  out = LayerNorm(z) * gamma + beta;  stats = (mean, rstd)
 * == b4c_gemm_nt followed by b4c_add_dropout_layernorm_fwd, bit for bit, without y in HBM.
 * z / out [M][N] bf16 (pitch N), stats [M][2] fp32, x pitch ldx. */
int b4c_gemm_nt_add_ln(const void *A, int lda, const void *Bt, int ldb, const float *bias, const void *x, int ldx,
                       const float *gamma, const float *beta, void *z, void *out, float *stats, int M, int N, int K,
                       float eps, float dropout_rate, uint64_t seed, int dtype, void *stream);

/* dW[K][N] += A[M][K]^T . G[M][N]   and  db[N] += colsum(G)   (fp32 outputs ADDED to what is there: the
 * caller zeroes or accumulates; db may be NULL).  The reduction over the M (token) axis is split over
 * workgroups when the output has few 128x128 tiles; the partial tiles are then summed
 *   - through `workspace` (device memory, 16-B aligned, >= b4c_gemm_tn_workspace_bytes(M,K,N,dtype)) in a
 *     fixed order: deterministic, no atomics.  The workspace is scratch: it may be shared by every call on
 *     the same stream;
 *   - with float atomics if workspace is NULL or too small (order-dependent rounding). */
int64_t b4c_gemm_tn_workspace_bytes(int M, int K, int N, int dtype);
int b4c_gemm_tn(const void *A, int lda, const void *G, int ldg, float *dW, int ldw, float *db, int M, int K,
                int N, int dtype, void *workspace, int64_t workspace_bytes, void *stream);
/* same with N = n_seg * seg_width cut into n_seg (<= 4) column segments, each accumulated into its own
 * dW_i [K][seg_width] / db_i [seg_width] (HOST arrays of device pointers): the fused Q|K|V projection
 * adds straight into the three gradient tensors. */
int b4c_gemm_tn_seg(const void *A, int lda, const void *G, int ldg, int n_seg, float *const *h_dW,
                    float *const *h_db, int seg_width, int M, int K, int dtype, void *workspace,
                    int64_t workspace_bytes, void *stream);

/* (ABI 9) the whole backward of one Dense layer of the encoder in ONE pass over its output gradient (transformer.py:112-116,
 * 158, 163-167 seen from the backward pass):   dX = G Wc^T (+ residual)   dW_s += X^T G_s   db_s += colsum(G_s)
 * X [M][128] the layer's input, G [M][128 n_seg] (n_seg = 3: q | k | v column blocks, 2: k | v, 1: a plain layer), Wc [128][128 n_seg]
 * (row = input feature, the dX operand of b4c_gemm_nt), residual [M][128] or NULL, dX [M][128]; dW_s fp32 [128][ld_dw] Keras
 * layout, db_s fp32 [128] or NULL (h_dW / h_db: HOST arrays of n_seg device pointers).  bf16, 128-wide layers only; G is read from
 * HBM once (as b4c_gemm_nt + b4c_gemm_tn it is read twice).  Deterministic: per-workgroup partial sums meet in workgroup order
 * through the caller's scratch. */
int64_t b4c_gemm_dxdw_workspace_bytes(int64_t M, int n_seg);
int b4c_gemm_dxdw(const void *X, int ldx, const void *G, int ldg, const void *Wc, int ldw, const void *residual, int ldr,
                  void *dX, int ldo, int n_seg, float *const *h_dW, float *const *h_db, int ld_dw, int64_t M,
                  void *workspace, int64_t workspace_bytes, void *stream);

/* (ABI 10) the whole backward of the position-wise feed-forward block of one encoder layer (transformer.py:154-170 seen from the
 * backward pass: LayerNormalization, dropout, residual add, Dense(d_model), Dense(dff, relu)) in ONE pass:
 *   dz, dy   = LayerNorm / dropout backward of dout (as b4c_add_dropout_layernorm_bwd: z [M][128] the saved LayerNorm input,
 *              stats [M][2], gamma [128]; dropout_rate / seed of the forward pass's mask)          -- never written to HBM
 *   dW2 += H^T dy,  db2 += colsum(dy),  dh = (dy W2c^T) o [H > 0]                                  -- dh never written to HBM
 *   dW1 += X^T dh,  db1 += colsum(dh),  dX = dh W1c^T + dz,   dgamma / dbeta += the LayerNorm's parameter gradients
 * H [M][ldh] = relu(X W1 + b1) with Fp stored columns (F valid, Fp = F rounded up to 8, <= 128), X [M][ldx] (128 columns),
 * W2c [Fp][ldw2] (row = hidden column, 128 entries) and W1c [128][ldw1] (row = input feature, Fp entries): the dX operands of
 * b4c_gemm_nt for the two layers.  dW1 fp32 [128][ld_dw1], dW2 fp32 [F][ld_dw2] (Keras layouts), db1 [F] / db2 [128] or NULL.
 * bf16, d_model = 128 only.  Reads dout, z, H, X once and writes dX: 1,240 B per token against 3,340 for the five kernels it
 * replaces.  Deterministic (no float atomics: per-workgroup partial sums meet in a fixed order through the caller's scratch). */
int64_t b4c_ffn_bwd_workspace_bytes(int64_t M);
int b4c_ffn_bwd(const void *dout, const void *z, const float *stats, const float *gamma, float dropout_rate, uint64_t seed,
                const void *H, int ldh, const void *X, int ldx, const void *W2c, int ldw2, const void *W1c, int ldw1,
                int F, int Fp, void *dX, int ldo, float *dW1, int ld_dw1, float *db1, float *dW2, int ld_dw2, float *db2,
                float *dgamma, float *dbeta, int64_t M, void *workspace, int64_t workspace_bytes, void *stream);

/* (ABI 11) the backward of the attention block's tail (transformer.py:158-162, 204-207 seen from the backward pass: LayerNormalization,
 * dropout, residual add, the output projection Dense(d_model)) in ONE pass:
 *   dz, dy = LayerNorm / dropout backward of dout (z, stats, gamma, dropout_rate, seed as b4c_add_dropout_layernorm_bwd);
 *   dZ [M][128] = dz (the residual branch's gradient: the caller adds it to the block input's gradient);   dy never reaches HBM;
 *   dO [M][ld_do] = dy Wc^T,   dW += O^T dy,   db += colsum(dy),   dgamma / dbeta += the LayerNorm's parameter gradients.
 * O [M][ldo_in] the projection's input (128 columns), Wc [128][ldw] (row = input feature of the projection: the dX operand of
 * b4c_gemm_nt), dW fp32 [128][ld_dw] Keras layout, db fp32 [128] or NULL.  bf16, d_model = 128 only.  1,288 B per token against 1,800 for
 * b4c_add_dropout_layernorm_bwd + b4c_gemm_dxdw.  Deterministic (fixed-order reduction through the caller's scratch). */
int64_t b4c_attn_out_bwd_workspace_bytes(int64_t M);
int b4c_attn_out_bwd(const void *dout, const void *z, const float *stats, const float *gamma, float dropout_rate, uint64_t seed,
                     const void *O, int ldo_in, const void *Wc, int ldw, void *dZ, void *dO, int ld_do,
                     float *dW, int ld_dw, float *db, float *dgamma, float *dbeta, int64_t M,
                     void *workspace, int64_t workspace_bytes, void *stream);

/* (ABI 12) the forward of the position-wise feed-forward block (transformer.py:154-170) in ONE pass:
 *   H = relu(X W1 + b1) [M][ldh] (Fp stored columns, F valid; the backward pass reads it),   Z = X + dropout(H W2 + b2) (or NULL),
 *   Out = LayerNorm(Z) gamma + beta,   stats [M][2] = (mean, 1 / std) of Z's rows (or NULL).
 * W1t [Fp][ldw1] (row = hidden column, 128 entries) and W2t [128][ldw2] (row = output column, Fp entries): the forward operands of
 * b4c_gemm_nt for the two layers; b1 [Fp], b2 / gamma / beta [128] fp32.  bf16, d_model = 128, F <= 128 only.  X is read once and H is
 * not read back: 446 MB per launch at 456 k rows against 658 for b4c_gemm_nt + b4c_gemm_nt_add_ln. */
int b4c_ffn_fwd(const void *X, int ldx, const void *W1t, int ldw1, const float *b1, const void *W2t, int ldw2, const float *b2,
                const float *gamma, const float *beta, int F, int Fp, void *H, int ldh, void *Z, void *Out, float *stats,
                int64_t M, float eps, float dropout_rate, uint64_t seed, void *stream);

/* several dW problems over the SAME M tokens in one launch (bf16; the four weight gradients of an encoder layer):
 * the ~256 workgroups of the split are shared by all problems, so every output tile has ~256 / (total tiles)
 * partial sums, and the group needs one main + one reduce kernel.  Problem i: dW_i[K][n_seg * seg_width] split into
 * n_seg column segments as b4c_gemm_tn_seg.  h_desc is a HOST array.  Always deterministic (workspace required). */
typedef struct {
    const void *A;          /* [M][lda], K columns used */
    const void *G;          /* [M][ldg], n_seg * seg_width columns used */
    float *dW[4];           /* device pointers, [K][ldw] each */
    float *db[4];           /* or NULL */
    int32_t lda, ldg, K, n_seg, seg_width, ldw;
} b4c_tn_desc;
int64_t b4c_gemm_tn_group_workspace_bytes(const b4c_tn_desc *h_desc, int n_prob, int M);
int b4c_gemm_tn_group(const b4c_tn_desc *h_desc, int n_prob, int M, int dtype, void *workspace, int64_t workspace_bytes,
                      void *stream);

/* ---- R8: attention -------------------------------------------------------------------
 * replaces MultiHeadAttention.split_heads + scaled_dot_product_attention + merge
 * (transformer.py:64-97, 130-156).  qkv: [B*S][ld_qkv] with q | k | v column blocks of d_model
 * each (head h = columns h*dh..), key_pad [B*S] from the embedding stage (key-side mask, -1e9).
 *   o[B*S][ld_o] = softmax(q k^T / sqrt(dh) + pad * -1e9) v   (heads merged)
 *   lse[B][H][S] = log-sum-exp of the masked, scaled logits (saved for backward) */
int b4c_attn_fwd(const void *qkv, int ld_qkv, const uint8_t *key_pad, void *o, int ld_o, float *lse, int B,
                 int S, int H, int dh, int dtype, void *stream);
/* dqkv [B*S][ld_dqkv] from do; delta[B][H][S] is scratch (fp32): the row kernels keep rowsum(dO o O) there, the
 * bf16 MFMA kernel computes it in LDS and uses the first word as the work counter of its persistent grid. */
int b4c_attn_bwd(const void *qkv, int ld_qkv, const uint8_t *key_pad, const void *o, int ld_o,
                 const void *d_o, int ld_do, const float *lse, float *delta, void *dqkv, int ld_dqkv, int B,
                 int S, int H, int dh, int dtype, void *stream);

/* same with a caller-provided scratch: bf16 sequences of 256 < S <= 512 run the MFMA backward block by block of 256 keys
 * (one launch) and sum the partial dQ in an fp32 accumulator of b4c_attn_bwd_workspace_bytes(B, S, H, dh, dtype) bytes (0 when no
 * workspace is needed).  Without it those shapes fall back to the fp32-math row kernels (a notice is printed once). */
int64_t b4c_attn_bwd_workspace_bytes(int B, int S, int H, int dh, int dtype);
int b4c_attn_bwd_ws(const void *qkv, int ld_qkv, const uint8_t *key_pad, const void *o, int ld_o,
                    const void *d_o, int ld_do, const float *lse, float *delta, void *dqkv, int ld_dqkv, int B,
                    int S, int H, int dh, void *workspace, int64_t workspace_bytes, int dtype, void *stream);

/* ---- R10: residual + dropout + LayerNorm ---------------------------------------------
 * replaces EncoderLayer.call's  LN(x + dropout(y))  (transformer.py:204-206, 209-211),
 * LayerNormalization(epsilon) :183-184 (biased variance, eps inside rsqrt).
 *   z = x + drop(y);  out = (z - mean) * rstd * gamma + beta;  stats[row] = {mean, rstd}
 * z (T, pitch d) is saved for backward; d % 8 == 0. */
int b4c_add_dropout_layernorm_fwd(const void *x, const void *y, const float *gamma, const float *beta,
                                  void *z, void *out, float *stats, int64_t rows, int d, float eps,
                                  float dropout_rate, uint64_t seed, int dtype, void *stream);
/* dz -> residual branch; dy = dropmask * dz (written only if dropout_rate > 0, else may be NULL);
 * dgamma/dbeta fp32 [d], accumulated with atomics (caller zeroes). */
int b4c_add_dropout_layernorm_bwd(const void *dout, const void *z, const float *stats, const float *gamma,
                                  void *dz, void *dy, float *dgamma, float *dbeta, int64_t rows, int d,
                                  float dropout_rate, uint64_t seed, int dtype, void *stream);
/* (ABI 9) the same with a caller scratch (b4c_add_dropout_layernorm_bwd_workspace_bytes): workspace != NULL selects the
 * DETERMINISTIC form of dgamma / dbeta -- per-workgroup sums in a fixed order, added block by block -- instead of float atomics. */
int64_t b4c_add_dropout_layernorm_bwd_workspace_bytes(int64_t rows, int d);
int b4c_add_dropout_layernorm_bwd_ws(const void *dout, const void *z, const float *stats, const float *gamma,
                                     void *dz, void *dy, float *dgamma, float *dbeta, int64_t rows, int d,
                                     float dropout_rate, uint64_t seed, void *workspace, int64_t workspace_bytes, int dtype,
                                     void *stream);

/* ---- R11: [MASK]-position index generation and row gather -----------------------------
 * replaces _gather_output_by_raw_value (clickstream_transformer.py:260-297):
 * tf.where(raw == value) row-major, ragged per batch row, gather_nd, to_tensor(0).
 * counts[B], offsets[B+1] (exclusive scan; offsets[B] = R), flat_idx[cap] = b*S+s in row-major
 * order (entries >= R untouched), maxcount[1].  All int32 except ids (int64).
 * More matches than `cap` (a caller-limited row count, e.g. B x max_masked_per_row of the sync-free Cloze path): the
 * offsets are clamped to cap -- consumers that size their row tensors by cap never index past them -- maxcount[0] comes
 * back as -(longest row) - 1 (negative even when that is 0) and `poison` (optional int32 flag) is set to -1, for the caller to
 * fold into the loss as NaN. */
int b4c_mask_positions(const int64_t *ids, int B, int S, int64_t value, int32_t *counts, int32_t *offsets,
                       int32_t *flat_idx, int32_t cap, int32_t *maxcount, int32_t *poison, void *stream);
/* padded_idx[B*M] = flat index of the m-th match of row b, or -1 (pad slot). */
int b4c_padded_index(const int32_t *counts, const int32_t *offsets, const int32_t *flat_idx, int B, int M,
                     int32_t *padded_idx, void *stream);
/* out[r][:] = idx[r] >= 0 ? in[idx[r]][:] : 0      (width % 8 == 0) */
int b4c_gather_rows(const void *in, int ld_in, const int32_t *idx, void *out, int ld_out, int64_t n_out,
                    int width, int dtype, void *stream);
/* dst (n_dst rows) = 0, then dst[idx[r]][:] = src[r][:] for idx[r] >= 0 (indices unique). */
int b4c_scatter_rows(const void *src, int ld_src, const int32_t *idx, void *dst, int ld_dst, int64_t n_src,
                     int64_t n_dst, int width, int dtype, void *stream);

/* ---- R12 tail, R13, R14: softmax, masked sparse cross-entropy --------------------------
 * replaces Dense(V, softmax)'s activation (head.py:36), cloze_output_adaptor + MaskedLoss with
 * tf.keras.backend.sparse_categorical_crossentropy (utils.py:56-134, losses.py:31-98, main.py:89).
 * probs[r][0..V) = softmax(logits[r][0..V)); pad columns V..ld are written 0. */
int b4c_softmax_rows(const void *logits, int ld_in, void *probs, int ld_out, int64_t R, int V, int dtype,
                     void *stream);
/* per-row loss on PROBABILITIES (the reference's dataflow): labels fp32 as the reference keeps
 * them (-1 = pad -> loss 0, not counted).  item_loss[R], n_valid[1] (+= count). */
int b4c_sparse_ce_from_probs(const void *probs, int ld, const float *labels, float *item_loss,
                             float *n_valid, int64_t R, int V, int variant, int dtype, void *stream);
/* fused training form: logits -> per-row loss and, IN PLACE, dlogits = grad_scale[0] * d loss/d logits
 * (grad_scale is a device scalar, e.g. 1/R_valid).  labels int32 label-space ids; label < 0 or >= V
 * -> row ignored (zero gradient).  pad columns get 0. */
int b4c_softmax_ce_fwd_bwd(void *logits, int ld, const int32_t *labels, float *item_loss,
                           const float *grad_scale, int64_t R, int V, int variant, int dtype, void *stream);

/* ---- R12-R14 without the logits in HBM (bf16 training path) -------------------------------
 * replaces, for training, Dense(V, softmax) (head.py:36) + cloze_output_adaptor + MaskedLoss +
 * sparse_categorical_crossentropy (utils.py:56-134, losses.py:31-98, main.py:89) AND their backward:
 * the (R x V) logits are recomputed tile by tile in MFMA accumulators instead of being stored.
 *   h [R][ld_h] bf16 (head input), wt [V][ld_w] bf16 (vocabulary-major projection weights, K columns),
 *   bias [V] fp32 or NULL, labels [R] int32 (< 0: ignored row), grad_scale: device scalar d total / d row loss.
 * b4c_vocab_ce_fwd:  item_loss[R], dh [R][ld_dh] bf16 = grad_scale * d row_loss / d h, rowscal [R][8] fp32
 *   (scratch handed to b4c_vocab_ce_dw); workspace >= b4c_vocab_ce_workspace_bytes(R, V, K), 16-B aligned.
 * b4c_vocab_ce_dw:   dW [K][ldw] fp32 += d loss / d kernel (Keras layout [in][out]), db [V] += (or NULL);
 *   same workspace (scratch, contents not carried over).
 * K in {64, 128}; variant B4C_CE_TF (clip-renormalised, as the TF backend) or B4C_CE_PLAIN. */
int64_t b4c_vocab_ce_workspace_bytes(int64_t R, int V, int K);
int b4c_vocab_ce_fwd(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, const int32_t *labels,
                     const float *grad_scale, float *item_loss, void *dh, int ld_dh, float *rowscal,
                     void *workspace, int64_t workspace_bytes, int64_t R, int V, int K, int variant, void *stream);
int b4c_vocab_ce_dw(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, const int32_t *labels,
                    const float *rowscal, float *dW, int ldw, float *db, void *workspace, int64_t workspace_bytes,
                    int64_t R, int V, int K, int deterministic, void *stream);
/* (ABI version 8) `deterministic` != 0 (b4c_vocab_ce_dw and its pieces): dW / db are summed in a fixed order -- one workgroup
 * per vocabulary tile walks every token (plain adds instead of float atomics over token splits), and the label term goes
 * through a stable sort of the rows by label with one wave summing each run: the same bits on every run of the same inputs.
 * Costs the sweep its token split (fewer workgroups than CUs' worth of rounds); workspace as b4c_vocab_ce_workspace_bytes. */
/* (ABI version 5) b4c_vocab_ce_dw in pieces, so that the sweep can run BESIDE the HBM-bound encoder backward:
 * b4c_vocab_ce_dw_sweep adds the dlogit part of dW / db for the 128-id vocabulary tiles [tile_begin, tile_end) only
 * (tiles own disjoint columns of dW: any partition of [0, ceil(V / 128)) over any number of calls gives the full sweep);
 * background_workgroups > 0 launches it as a background kernel -- at most that many 256-thread workgroups (one wave per
 * SIMD, 220 registers) that walk the units, leaving the rest of every CU's registers and issue slots to the kernels of
 * another stream -- 0 launches the foreground form of b4c_vocab_ce_dw.  b4c_vocab_ce_dw_labels adds the label term
 * (dW[:, y] -= yd h_row, db[y] -= yd) once; workspace >= V * K * 4 bytes.  Sweep + labels == b4c_vocab_ce_dw. */
int b4c_vocab_ce_dw_sweep(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, const float *rowscal,
                          float *dW, int ldw, float *db, int64_t R, int V, int K, int tile_begin, int tile_end,
                          int background_workgroups, int deterministic, void *stream);
int b4c_vocab_ce_dw_labels(const void *h, int ld_h, const int32_t *labels, const float *rowscal, float *dW, int ldw,
                           float *db, void *workspace, int64_t workspace_bytes, int64_t R, int V, int K, int deterministic,
                           void *stream);
/* ---- (ABI version 4) R12 for scoring: Dense(V, softmax) (head.py:36) with ONE pass over the (R x V) tensor ------------------
 * replaces the materialised projection + softmax (b4c_gemm_nt + b4c_softmax_rows: write, read, write) of the
 * bf16 path: b4c_vocab_lse recomputes the logits in MFMA accumulators (nothing reaches HBM) and leaves
 * lse2[row] = log2 sum_j 2^(x_j log2 e); b4c_gemm_nt_softmax then writes probs [R][ldc] bf16 =
 * 2^((h wt^T + bias) log2 e - lse2[row]) straight from its accumulators.  K in {64, 128} (lse) / K <= 128 (projection),
 * N and ldc multiples of 8, operands 16-B aligned; workspace >= b4c_vocab_ce_workspace_bytes(R, V, K). */
int b4c_vocab_lse(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, float *lse2, void *workspace,
                  int64_t workspace_bytes, int64_t R, int V, int K, void *stream);
int b4c_gemm_nt_softmax(const void *A, int lda, const void *Bt, int ldb, void *C, int ldc, int M, int N, int K,
                        const float *bias, const float *lse2, void *stream);

/* ---- (ABI version 8) R15 without the (R x V) scores in memory: ranking over the vocabulary for the bf16 scoring path -----
 * replaces tf.math.top_k + the Recall / NDCG bookkeeping (utils.py:161-190, 225-255) on MATERIALISED probabilities
 * (b4c_gemm_nt_softmax + b4c_topk_rows: 4.1 GB written and read back at C2) by sweeps that keep the logits tile in MFMA
 * accumulators.  Softmax is monotone: the logits rank as the probabilities do.  h [R][ld_h] bf16 (trunk output), wt
 * [>= V][ld_w] bf16 vocabulary-major, bias fp32 [V] or NULL, K in {64, 128}; workspace >= b4c_vocab_rank_workspace_bytes.
 * Scores are formed as b4c_gemm_nt forms them (sum over k from zero, bias last).
 *   b4c_vocab_rank: rank[r] = number of items ranked before the label y_r = #{j : x_j > x_y} + #{j < y : x_j == x_y}
 *     (ties -> lower index first, as tf.math.top_k); negative for rows without a valid label (y < 0 or y >= V).
 *     HitRate@k = [rank < k], NDCG@k = [rank < k] / log2(rank + 2) for every k: b4c_rank_metrics.
 *   b4c_vocab_topk: idx [R][k] int32 = the k best item ids in order (k <= B4C_MAX_TOPK), optional hit / ndcg [R] of `labels`.
 *     Rows with more than 128 candidates at the selection threshold (mass ties) get ids -1 (hit / ndcg NaN) and are counted
 *     in overflow[0]: the caller ranks those rows on materialised scores (b4c_gemm_nt + b4c_topk_rows). */
int64_t b4c_vocab_rank_workspace_bytes(int64_t R, int V, int K);
int b4c_vocab_rank(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, const int32_t *labels, int32_t *rank,
                   void *workspace, int64_t workspace_bytes, int64_t R, int V, int K, void *stream);
int b4c_rank_metrics(const int32_t *rank, int64_t R, int k, float *hit, float *ndcg, void *stream);
int b4c_vocab_topk(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, int k, int32_t *idx,
                   const int32_t *labels, float *hit, float *ndcg, int32_t *overflow, void *workspace,
                   int64_t workspace_bytes, int64_t R, int V, int K, void *stream);

/* ---- (ABI version 4) scalar bookkeeping of the masked mean (losses.py:80-98) and of the head's backward, fused:
 * b4c_label_scale: out[0] = 1 / n_valid (0 if none), out[1] = n_valid, valid = 0 <= label < V (the mask of losses.py:80).
 * b4c_sum_scaled:  out[0] = scale[0] * sum item[0..R) in a fixed order; NaN if poison != NULL and poison[0] < 0.
 * b4c_vocab_ce_apply_grad: folds the upstream gradient g (device scalar) into what b4c_vocab_ce_fwd left for the
 *   backward: dh_out = g * dh, rowscal_out = rowscal with its gradient-linear columns times g.
 * b4c_relu_gate:   out = act > 0 ? g : 0 (n elements, a multiple of 8): relu'(Dense) of head.py:35 on a gradient. */
int b4c_label_scale(const int32_t *labels, int64_t R, int V, float *out, void *stream);
int b4c_sum_scaled(const float *item, int64_t R, const float *scale, const int32_t *poison, float *out, void *stream);
int b4c_vocab_ce_apply_grad(const void *dh, int ld, const float *rowscal, const float *g, void *dh_out, int ld_out,
                            float *rowscal_out, int64_t R, int K, void *stream);
int b4c_relu_gate(const void *g, const void *act, void *out, int64_t n, int dtype, void *stream);

/* ---- R15: top-k ids, HitRate@k / NDCG@k -------------------------------------------------
 * replaces tf.math.top_k + the Recall / NDCG update_state arithmetic (utils.py:161-190, 225-255).
 * topk_idx[R][k] int32, largest first, ties -> lower index.  labels (int32, may be NULL):
 * hit[r] = any(topk == label), ndcg[r] = sum_k (topk[k]==label) / log2(k+2). */
int b4c_topk_rows(const void *scores, int ld, int64_t R, int V, int k, int32_t *topk_idx,
                  const int32_t *labels, float *hit, float *ndcg, int dtype, void *stream);

/* same with a scratch int32 [R] (`redo`): rows are first handled by a threshold-selection kernel that reads each row
 * from HBM once (per-thread maxima -> k-th best of them -> the ~k candidates not worse than it -> ordered pick);
 * rows with more than 1024 candidates (massive ties) are flagged there and redone by the list kernel.  Same results. */
int b4c_topk_rows_ws(const void *scores, int ld, int64_t R, int V, int k, int32_t *topk_idx,
                     const int32_t *labels, float *hit, float *ndcg, int32_t *redo, int dtype, void *stream);

/* ---- R16: Adam (Keras semantics, eps outside the sqrt) ------------------------------------
 * replaces tf.keras.optimizers.Adam(1e-3, .9, .999, 1e-9) (main.py:87), dense update over a flat
 * fp32 arena: m,v EMA; p -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (host).
 * grad_mul scales g first (1/world_size for mean reduction; 1 for the reference's sum). */
int b4c_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float lr_t, float beta1,
                  float beta2, float eps, float grad_mul, void *stream);

/* Adam for a row-sparse table inside the arena (config 5: 2M-row tables, < 0.1 % of the rows touched per step) -- the same
 * dense-equivalent Keras update (main.py:87), evaluated lazily per row: stamp[row] (int32, 0 = never touched) is the last
 * step the row is current through; a row is brought up to date by replaying the zero-gradient steps it missed with the
 * fp32 operations of b4c_adam_step in the same order (lr_hist[s] = the lr_t the host used at step s, s = 1 .. t), so
 * the table equals the dense kernel's bit for bit once every row is caught up.
 *   mode 0: bring the rows to step t (zero-gradient steps only): before anything reads them.
 *   mode 1: bring the rows to step t - 1, take step t with the gradient rows of g, zero those gradient rows.
 * ids (int64 [n], clamped to the table; a row named several times is handled once) or, ids == NULL, the rows
 * [row_lo, row_lo + n).  p, g, m, v: the table's [rows][width] slices of the four arenas (width % 4 == 0). */
int b4c_adam_rows(float *p, float *g, float *m, float *v, int32_t *stamp, const int64_t *ids, int64_t n, int64_t row_lo,
                  int64_t rows, int width, const float *lr_hist, int t, float beta1, float beta2, float eps,
                  float grad_mul, int mode, void *stream);

/* keep-mask hash used by every dropout site (exposed so hosts/tests can regenerate masks):
 * returns 1 if element e is kept under (seed, rate). */
int b4c_keep(uint64_t seed, uint64_t e, float rate);

/* ==== round-2 additions (ABI version 3) ================================================== */

/* ---- stand-alone dropout: Encoder.call's input dropout (transformer.py:263) when the Encoder is used
 * without the embedding stage that normally fuses it.  y[e] = keep(seed, e) ? x[e] / (1 - rate) : 0, e = flat
 * element index; applied to a gradient it is its own backward.  n % 8 == 0. */
int b4c_dropout(const void *x, void *y, int64_t n, float rate, uint64_t seed, int dtype, void *stream);

/* ---- backward of the materialised-probability route: the reference trains through
 * loss(y, model(x)) with model(x) = (B, M, V) probabilities (main.py:159-165, head.py:36-47, losses.py:31-98).
 * softmax:  dlogits_j = p_j (g_j - sum_i g_i p_i); pad columns V..ldx of dlogits are written 0.
 * sparse CE on probabilities (b4c_sparse_ce_from_probs): dprobs = gscale[0] * d item_loss / d probs for rows
 * whose label is not the pad (-1), 0 for pad rows; gscale is a DEVICE scalar (upstream gradient / n_valid). */
int b4c_softmax_rows_bwd(const void *probs, int ldp, const void *dprobs, int ldg, void *dlogits, int ldx,
                         int64_t R, int V, int dtype, void *stream);
int b4c_sparse_ce_from_probs_bwd(const void *probs, int ld, const float *labels, const float *gscale, void *dprobs,
                                 int ld_dp, int64_t R, int V, int variant, int dtype, void *stream);

/* ---- the other heads (head.py:4-26, 50-69): Dense(activation='sigmoid') and its backward
 * (dx = dy * y * (1 - y), y = the forward's output).  n % 8 == 0. */
int b4c_sigmoid_fwd(const void *x, void *y, int64_t n, int dtype, void *stream);
int b4c_sigmoid_bwd(const void *y, const void *dy, void *dx, int64_t n, int dtype, void *stream);

/* MaskedLoss with tf.keras.backend.binary_crossentropy on probabilities and the optional pos_weight
 * (losses.py:31-98): o = clip(p, 1e-7, 1 - 1e-7); bce = -(t log(o + 1e-7) + (1 - t) log(1 - o + 1e-7));
 * weight = pos_weight where t == 1 (pos_weight <= 0: none).  labels fp32, -1 = pad (loss 0, not counted).
 * sums[0] += sum of weighted item losses, sums[1] += number of non-pad items (caller zeroes);
 * item_loss [n] and dprobs [n] (fp32, = weight * d bce / d p, unscaled) may be NULL. */
int b4c_masked_bce(const void *probs, const float *labels, float pos_weight, float *item_loss, float *sums,
                   float *dprobs, int64_t n, int dtype, void *stream);

/* counts behind PositiveRate, PredictedPositives and F1Score (metrics.py:5-87) in one pass; out6 += :
 * [0] sum mask*y_true  [1] sum mask  [2] sum mask*round(y_pred)  [3] tp  [4] condition_true  [5] predicted_true
 * (mask = y_true != -1; tf.round = half to even; the F1 counts are not masked, as in the reference). */
int b4c_binary_counts(const float *y_true, const void *y_pred, float *out6, int64_t n, int dtype, void *stream);

/* ---- labels of the sync-free Cloze step: padded (B, M) fp32 labels (-1 pad; row b's labels are its first
 * counts[b] entries, input_pipeline.py:198-214) -> compact int32 [cap] in the row-major order of
 * b4c_mask_positions; entries at and beyond R = offsets[B] are set to -1 in `out` (ignored rows) and, when
 * flat_idx is given, to -1 there too (b4c_gather_rows then yields zero rows). */
int b4c_compact_labels(const float *labels, int B, int M, const int32_t *counts, const int32_t *offsets,
                       int32_t *out, int32_t *flat_idx, int32_t cap, void *stream);

/* ---- attention weights on request: the second result of MultiHeadAttention.call /
 * scaled_dot_product_attention (transformer.py:64-97, 137-160), which the encoder discards (:203).
 * weights fp32 [B][H][S][S] = exp(q k^T / sqrt(dh) + pad * -1e9 - lse). */
int b4c_attn_weights(const void *qkv, int ld_qkv, const uint8_t *key_pad, const float *lse, float *weights, int B,
                     int S, int H, int dh, int dtype, void *stream);

/* ---- tied-weight head (north-star extension, no reference counterpart): dst[n][k] += src[k][n], fp32 --
 * the projection gradient [K][V] added into rows of the embedding-table gradient [V][K]. */
int b4c_transpose_add(const float *src, int ld_src, float *dst, int ld_dst, int K, int N, void *stream);

/* ---- row-sparse gradient exchange: gather the touched rows of an fp32 gradient table
 * (idx int64, < 0 -> zero row) and add received rows back (float atomics; idx < 0 skipped).  width % 4 == 0. */
int b4c_rows_gather_f32(const float *src, int ld_src, const int64_t *idx, float *out, int ld_out, int64_t n,
                        int width, void *stream);
int b4c_rows_scatter_add_f32(const float *src, int ld_src, const int64_t *idx, float *dst, int ld_dst, int64_t n,
                             int width, void *stream);

/* ---- sampled-softmax head (BASELINE.json configs[4], SURVEY D10: north_star extension, NO reference counterpart) ----
 * Shared negatives from a log-uniform sampler WITH replacement over [0, range_max): P(c) = log((c+2)/(c+1)) / log(range_max+1)
 * (the distribution of tf.random.log_uniform_candidate_sampler), expected count Q(c) = n P(c).  Sample i uses the 24-bit
 * uniform (b4c rand64(seed, i) >> 40) / 2^24, so a host can regenerate the ids.  ids int64 [n], logq fp32 [n] = log Q(id). */
int b4c_log_uniform_sample(uint64_t seed, int n, int64_t range_max, int64_t *ids, float *logq, void *stream);
/* out[r] = sum_c a[r][c] b[r][c]  (fp32; width % 8 == 0): the true-class logits h_r . w_{y_r}. */
int b4c_row_dot(const void *a, int lda, const void *b, int ldb, float *out, int64_t R, int width, int dtype, void *stream);
/* tf.nn.sampled_softmax_loss semantics (remove_accidental_hits): Z [R][ld] holds the K negatives' logits with bias - logQ
 * folded in, ztrue[r] = h_r . w_y + b_y; the kernel subtracts logQ(y), drops negatives equal to the row's label, and
 * writes loss_r = logsumexp(z_true, z_neg) - z_true to item_loss, grad_scale[0] * softmax over the negatives IN PLACE of
 * Z, and grad_scale[0] * (softmax_true - 1) to dtrue.  Rows with label < 0 or >= range_max are ignored. */
int b4c_sampled_ce_fwd_bwd(void *Z, int ld, const float *ztrue, const int64_t *samples, const int32_t *labels,
                           int64_t range_max, float *item_loss, float *dtrue, const float *grad_scale, int64_t R,
                           int K, int dtype, void *stream);
/* dst[idx[i]] += src[i] (fp32 atomics, idx < 0 skipped);  out[r][:] = scale[r] * src[r][:] (fp32 out). */
int b4c_scatter_add_1d(const float *src, const int64_t *idx, float *dst, int64_t n, void *stream);
int b4c_row_scale_f32(const void *src, int ld, const float *scale, float *out, int ld_out, int64_t R, int width,
                      int dtype, void *stream);

/* ---- packed (padding-free) token layout -------------------------------------------------------------------------------
 * The reference pads every sequence to the batch maximum and runs the encoder on the pads too (transformer.py:376-402); pad
 * KEYS are masked out (:38-41, :90-91), pad QUERIES produce rows nobody reads, and under the Cloze loss their gradient is
 * exactly zero.  The throughput path therefore drops the pad positions: the encoder runs on the T_real real tokens only,
 * every row-wise kernel (GEMM, LayerNorm, embedding, Adam) unchanged on [T_real][d] tensors, attention per sequence through
 * cu_seqlens.  Results at real positions are those of the dense layout.
 * b4c_nonpad_positions: counts[B] real tokens per sequence, cu_seqlens[B+1] (exclusive scan, cu[B] = T_real),
 *   token_src[cap] = b*S + s of every real token in row-major order, packed_of[B*S] = packed row of a dense position or -1
 *   (may be NULL), maxcount[1] = longest sequence (may be NULL).  `cap` is the caller's token count: cu_seqlens are
 *   clamped to it (a wrong count can truncate sequences but never index past tensors sized by it), and when the true total
 *   differs from it maxcount[0] is written as -(longest) - 1: a poison flag.
 * b4c_remap_index: out[i] = idx[i] >= 0 ? map[idx[i]] : -1 ([MASK] positions of the dense layout -> packed rows).
 * b4c_embed_concat_pe_fwd_packed: b4c_embed_concat_pe_fwd writing only rows t < n_tokens, row t taken from dense position
 *   token_src[t] (ids and positional row s = token_src[t] % S); the dropout counter is the packed element index.
 * b4c_attn_{fwd,bwd}_varlen: attention with sequence b in rows cu_seqlens[b] .. cu_seqlens[b+1] (bf16, head depth 32 / 64,
 *   max_len <= 512); lse / delta keep the shape [B][H][max_len]; workspace as b4c_attn_bwd_ws (max_len > 256). */
int b4c_nonpad_positions(const int64_t *ids, int B, int S, int64_t pad_value, int32_t *counts, int32_t *cu_seqlens,
                         int32_t *token_src, int32_t cap, int32_t *packed_of, int32_t *maxcount, void *stream);
int b4c_remap_index(const int32_t *idx, const int32_t *map, int32_t *out, int64_t n, void *stream);
int b4c_embed_concat_pe_fwd_packed(int n_feat, const int64_t *const *h_ids, const float *const *h_tables,
                                   const int *h_dims, const int64_t *h_rows, const float *pe, float scale,
                                   void *out, int ld_out, uint8_t *key_pad, int B, int S, int d_model,
                                   float dropout_rate, uint64_t seed, const int32_t *token_src, int64_t n_tokens,
                                   int dtype, void *stream);
int b4c_attn_fwd_varlen(const void *qkv, int ld_qkv, const uint8_t *key_pad, const int32_t *cu_seqlens, void *o, int ld_o,
                        float *lse, int B, int max_len, int H, int dh, int dtype, void *stream);
int b4c_attn_bwd_varlen(const void *qkv, int ld_qkv, const uint8_t *key_pad, const int32_t *cu_seqlens, const void *o,
                        int ld_o, const void *d_o, int ld_do, const float *lse, float *delta, void *dqkv, int ld_dqkv,
                        int B, int max_len, int H, int dh, void *workspace, int64_t workspace_bytes, int dtype,
                        void *stream);

/* ---- (ABI version 4) attention for a few query rows per sequence: the LAST encoder layer of the Cloze path ------------
 * The reference runs every layer on every position (transformer.py:262-270) and keeps the rows at the [MASK] positions
 * (clickstream_transformer.py:281-295); in the last layer all other rows are never read.  There the queries are the masked
 * rows of a sequence, the keys / values all of its tokens (transformer.py:64-97 for those rows, fp32 math, both dtypes):
 *   q / o / dq [R][ld]: head h in columns h*dh..;  kv / dkv [T][ld]: k in columns h*dh.., v in H*dh + h*dh..;
 *   sequence b = token rows cu_seqlens[b] .. cu_seqlens[b+1] and query rows q_offsets[b] .. q_offsets[b+1];
 *   key_pad [T] (1 = padded key, may be NULL); lse [R][H] (natural log of the row's softmax denominator).
 * b4c_attn_mq_bwd writes dq for the R query rows and dk | dv for EVERY token row (zeros where no query reads them).
 * dh in {32, 64}; max_len = longest sequence (LDS sizing). */
/* order[i] = position (0..n-1) of the i-th smallest id, ties in position order (a stable sort of the token positions by
 * table row: the order b4c_embed_concat_pe_bwd_sorted walks); ids are clamped to [0, n_rows - 1] as the embedding kernels
 * clamp them.  LSD radix, 8-bit digits, 4 launches per pass; workspace >= b4c_sort_ids_workspace_bytes(n, n_rows), 4-B aligned. */
int64_t b4c_sort_ids_workspace_bytes(int64_t n, int n_rows);
int b4c_sort_ids(const int64_t *ids, int64_t n, int n_rows, int32_t *order, void *workspace, int64_t workspace_bytes, void *stream);
int b4c_gather_i64(const int64_t *src, const int32_t *idx, int64_t *out, int64_t n, void *stream);   /* out[i] = src[idx[i]] */
/* (ABI version 7) b4c_zero: nbytes of zeros at p (stream-ordered).  b4c_chain_ids: out[b] = [cls, sep, seq_0[b], sep, seq_1[b],
 * sep, ...] -- TransformerInputPrep._chain_sequences (clickstream_transformer.py:38-63) for already-mapped int64 ids;
 * seqs / lens / pitches are HOST arrays of n_seq (<= 8) device pointers, row lengths and row pitches (elements);
 * out has 2 + sum(lens) + n_seq columns on a pitch of ld_out. */
int b4c_zero(void *p, int64_t nbytes, void *stream);
int b4c_chain_ids(const int64_t *const *seqs, const int *lens, const int *pitches, int n_seq, int B, int64_t cls,
                  int64_t sep, int64_t *out, int ld_out, void *stream);
int b4c_rows_add(void *dst, int ld_dst, const int32_t *idx, const void *src, int ld_src, int64_t n_src, int width, int dtype,
                 int src_dtype, void *stream);   /* dst[idx[r]] += src[r] (idx distinct, < 0 skipped; src may be fp32 beside a
                                                  * bf16 dst): the query rows' gradient joins that of all token rows */
int b4c_poison_rows(void *x, int ld, int64_t rows, int width, const int32_t *flag, int dtype, void *stream);
                                                 /* x[rows][width] := NaN (B4C_I32: -1) when flag[0] < 0 (the negated maxcount of
                                                  * b4c_nonpad_positions / b4c_mask_positions: a caller-given count that the
                                                  * device's own contradicts), untouched otherwise: the scoring paths' answer to
                                                  * a wrong n_real_tokens, without a read-back (the loss paths fold the flag
                                                  * into the loss: b4c_sum_scaled) */
int b4c_attn_mq_fwd(const void *q, int ld_q, const void *kv, int ld_kv, const uint8_t *key_pad, const int32_t *cu_seqlens,
                    const int32_t *q_offsets, void *o, int ld_o, float *lse, int B, int max_len, int H, int dh, int dtype,
                    void *stream);
int b4c_attn_mq_bwd(const void *q, int ld_q, const void *kv, int ld_kv, const uint8_t *key_pad, const int32_t *cu_seqlens,
                    const int32_t *q_offsets, const void *o, int ld_o, const void *d_o, int ld_do, const float *lse,
                    void *dq, int ld_dq, void *dkv, int ld_dkv, int B, int max_len, int H, int dh, int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* B4C_H */
