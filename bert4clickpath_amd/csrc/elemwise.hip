// Small HBM-bound kernels around the hot path: stand-alone dropout (Encoder.call's input dropout when the
// Encoder is used without the embedding stage), the backward of the materialised-probability route
// (softmax rows, masked sparse CE on probabilities), the binary-task head pieces (sigmoid, masked binary CE,
// PositiveRate / PredictedPositives / F1 counts), label compaction for the sync-free Cloze step, the
// attention-weight matrix on request, a transposed accumulate for the tied-weight head, a row-sparse
// gather / scatter-add pair for the embedding-gradient exchange and a per-row dot product.
// gfx950, wave64, 16-byte accesses; one workgroup per row for the vocabulary-wide kernels.
#include <math.h>

#include <type_traits>

#include "common.h"

#define KERAS_EPS 1e-7f

static inline int ew_grid(int64_t work_items, int block) {
    int64_t g = ceil_div64(work_items, block);
    const int64_t cap = 256 * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <int NW> __device__ __forceinline__ float ew_block_sum(float v, float *buf) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += buf[i];
    return t;
}

// ------------------------------------------------------------------------------------------
// dropout: y[e] = keep(seed, e) ? x[e] / (1 - rate) : 0     (the same op is its own backward)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) dropout_kernel(const T *__restrict__ x, T *__restrict__ y, int64_t n8, float rate, uint64_t seed) {
    const float inv_keep = 1.0f / (1.0f - rate);
    const uint32_t thr = b4c_keep_threshold(rate);
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        Vec8<T>::load(x + i * 8, v);
        const uint32_t km = b4c_keep8(seed, (uint64_t)i * 8, thr);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ((km >> k) & 1u) ? v[k] * inv_keep : 0.f;
        Vec8<T>::store(y + i * 8, v);
    }
}

extern "C" int b4c_dropout(const void *x, void *y, int64_t n, float rate, uint64_t seed, int dtype, void *stream) {
    B4C_REQUIRE(x && y && n >= 0 && n % 8 == 0, "dropout: n must be a multiple of 8");
    B4C_REQUIRE(rate > 0.f && rate < 1.f, "dropout: rate %f must be in (0, 1)", rate);
    if (n == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(n / 8, 256);
    if (dtype == B4C_F32) dropout_kernel<float><<<grid, 256, 0, st>>>((const float *)x, (float *)y, n / 8, rate, seed);
    else if (dtype == B4C_BF16) dropout_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t *)x, (bf16_t *)y, n / 8, rate, seed);
    else B4C_REQUIRE(false, "dropout: dtype %d", dtype);
    return b4c_check_launch("dropout");
}

// ------------------------------------------------------------------------------------------
// softmax backward on materialised probabilities: dx_j = p_j (g_j - sum_i g_i p_i)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_bwd_kernel(const T *__restrict__ p, int ldp, const T *__restrict__ g, int ldg,
                                                               T *__restrict__ dx, int ldx, int64_t R, int V) {
    __shared__ float buf[4];
    const int tid = threadIdx.x;
    const int nch = (V + 7) >> 3;
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const T *pr = p + row * ldp, *gr = g + row * ldg;
        T *xr = dx + row * ldx;
        float s = 0.f;
        for (int c = tid; c < nch; c += 256) {
            float a[8], b[8];
            Vec8<T>::load(pr + c * 8, a);
            Vec8<T>::load(gr + c * 8, b);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c * 8 + k < V) s += a[k] * b[k];
        }
        s = ew_block_sum<4>(s, buf);
        const int nch_out = ldx >> 3;
        for (int c = tid; c < nch_out; c += 256) {
            float a[8], b[8];
            if (c < nch) { Vec8<T>::load(pr + c * 8, a); Vec8<T>::load(gr + c * 8, b); }
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = (c * 8 + k < V) ? a[k] * (b[k] - s) : 0.f;
            Vec8<T>::store(xr + c * 8, a);
        }
    }
}

extern "C" int b4c_softmax_rows_bwd(const void *probs, int ldp, const void *dprobs, int ldg, void *dlogits, int ldx,
                                    int64_t R, int V, int dtype, void *stream) {
    B4C_REQUIRE(probs && dprobs && dlogits && R >= 0 && V > 0, "softmax_rows_bwd: bad argument");
    B4C_REQUIRE(ldp % 8 == 0 && ldg % 8 == 0 && ldx % 8 == 0 && ldp >= V && ldg >= V && ldx >= V, "softmax_rows_bwd: pitches");
    if (R == 0) return B4C_OK;
    const int grid = (int)(R < 4096 ? R : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_F32)
        softmax_rows_bwd_kernel<float><<<grid, 256, 0, st>>>((const float *)probs, ldp, (const float *)dprobs, ldg, (float *)dlogits, ldx, R, V);
    else if (dtype == B4C_BF16)
        softmax_rows_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t *)probs, ldp, (const bf16_t *)dprobs, ldg, (bf16_t *)dlogits, ldx, R, V);
    else B4C_REQUIRE(false, "softmax_rows_bwd: dtype %d", dtype);
    return b4c_check_launch("softmax_rows_bwd");
}

// ------------------------------------------------------------------------------------------
// backward of b4c_sparse_ce_from_probs:  loss_r = log sum_j clip(p_j) - log clip(p_y)   (TF variant)
//   d loss_r / d p_j = u_j (1/S - [j = y] / clip(p_y)),  u_j = [eps <= p_j <= 1 - eps]
//   plain variant: - [j = y] / p_y.       dprobs = gscale[0] * that for valid rows, 0 for pad rows.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) sparse_ce_probs_bwd_kernel(const T *__restrict__ p, int ld, const float *__restrict__ labels,
                                                                  const float *__restrict__ gscale, T *__restrict__ dp, int ld_dp,
                                                                  int64_t R, int V, int variant) {
    __shared__ float buf[4];
    const int tid = threadIdx.x;
    const int nch = (V + 7) >> 3, nch_out = ld_dp >> 3;
    const float gs = gscale[0];
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const float lab = labels[row];
        const T *pr = p + row * ld;
        T *dr = dp + row * ld_dp;
        const int y = (int)lab;
        const bool pad = lab == -1.0f;
        const bool bad = !pad && (y < 0 || y >= V);
        if (pad || bad) {
            for (int c = tid; c < nch_out; c += 256) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (bad && c * 8 + k < V) ? NAN : 0.f;
                Vec8<T>::store(dr + c * 8, v);
            }
            continue;
        }
        float invS = 0.f;
        if (variant == B4C_CE_TF) {
            float s = 0.f;
            for (int c = tid; c < nch; c += 256) {
                float v[8];
                Vec8<T>::load(pr + c * 8, v);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c * 8 + k < V) s += fminf(fmaxf(v[k], KERAS_EPS), 1.0f - KERAS_EPS);
            }
            invS = 1.0f / ew_block_sum<4>(s, buf);
        }
        const float py = (float)pr[y];
        const float pyc = fminf(fmaxf(py, KERAS_EPS), 1.0f - KERAS_EPS);
        for (int c = tid; c < nch_out; c += 256) {
            float v[8];
            if (c < nch) Vec8<T>::load(pr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = c * 8 + k;
                float g = 0.f;
                if (j < V) {
                    if (variant == B4C_CE_TF) {
                        const float u = (v[k] >= KERAS_EPS && v[k] <= 1.0f - KERAS_EPS) ? 1.f : 0.f;
                        g = u * (invS - (j == y ? 1.0f / pyc : 0.f));
                    } else {
                        g = (j == y) ? -1.0f / py : 0.f;
                    }
                }
                v[k] = g * gs;
            }
            Vec8<T>::store(dr + c * 8, v);
        }
    }
}

extern "C" int b4c_sparse_ce_from_probs_bwd(const void *probs, int ld, const float *labels, const float *gscale, void *dprobs,
                                            int ld_dp, int64_t R, int V, int variant, int dtype, void *stream) {
    B4C_REQUIRE(probs && labels && gscale && dprobs && R >= 0 && V > 0, "sparse_ce_from_probs_bwd: bad argument");
    B4C_REQUIRE(ld % 8 == 0 && ld >= V && ld_dp % 8 == 0 && ld_dp >= V, "sparse_ce_from_probs_bwd: pitch");
    B4C_REQUIRE(variant == B4C_CE_TF || variant == B4C_CE_PLAIN, "sparse_ce_from_probs_bwd: variant %d", variant);
    if (R == 0) return B4C_OK;
    const int grid = (int)(R < 4096 ? R : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_F32)
        sparse_ce_probs_bwd_kernel<float><<<grid, 256, 0, st>>>((const float *)probs, ld, labels, gscale, (float *)dprobs, ld_dp, R, V, variant);
    else if (dtype == B4C_BF16)
        sparse_ce_probs_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t *)probs, ld, labels, gscale, (bf16_t *)dprobs, ld_dp, R, V, variant);
    else B4C_REQUIRE(false, "sparse_ce_from_probs_bwd: dtype %d", dtype);
    return b4c_check_launch("sparse_ce_from_probs_bwd");
}

// ------------------------------------------------------------------------------------------
// sigmoid (Dense(activation='sigmoid') of the binary / multi-label heads, head.py:12,59) and its backward
// ------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ void __launch_bounds__(256) sigmoid_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out, int64_t n8) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float v[8], w[8];
        Vec8<T>::load(a + i * 8, v);
        if (BWD) {   // a = y (probabilities), b = dy  ->  dx = dy * y * (1 - y)
            Vec8<T>::load(b + i * 8, w);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = w[k] * v[k] * (1.0f - v[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 1.0f / (1.0f + expf(-v[k]));
        }
        Vec8<T>::store(out + i * 8, v);
    }
}

extern "C" int b4c_sigmoid_fwd(const void *x, void *y, int64_t n, int dtype, void *stream) {
    B4C_REQUIRE(x && y && n >= 0 && n % 8 == 0, "sigmoid_fwd: n must be a multiple of 8");
    if (n == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(n / 8, 256);
    if (dtype == B4C_F32) sigmoid_kernel<float, false><<<grid, 256, 0, st>>>((const float *)x, nullptr, (float *)y, n / 8);
    else if (dtype == B4C_BF16) sigmoid_kernel<bf16_t, false><<<grid, 256, 0, st>>>((const bf16_t *)x, nullptr, (bf16_t *)y, n / 8);
    else B4C_REQUIRE(false, "sigmoid_fwd: dtype %d", dtype);
    return b4c_check_launch("sigmoid_fwd");
}

extern "C" int b4c_sigmoid_bwd(const void *y, const void *dy, void *dx, int64_t n, int dtype, void *stream) {
    B4C_REQUIRE(y && dy && dx && n >= 0 && n % 8 == 0, "sigmoid_bwd: n must be a multiple of 8");
    if (n == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(n / 8, 256);
    if (dtype == B4C_F32) sigmoid_kernel<float, true><<<grid, 256, 0, st>>>((const float *)y, (const float *)dy, (float *)dx, n / 8);
    else if (dtype == B4C_BF16) sigmoid_kernel<bf16_t, true><<<grid, 256, 0, st>>>((const bf16_t *)y, (const bf16_t *)dy, (bf16_t *)dx, n / 8);
    else B4C_REQUIRE(false, "sigmoid_bwd: dtype %d", dtype);
    return b4c_check_launch("sigmoid_bwd");
}

// ------------------------------------------------------------------------------------------
// masked binary cross-entropy on probabilities (MaskedLoss with tf.keras.backend.binary_crossentropy,
// losses.py:31-98): o = clip(p, eps, 1-eps); bce = -(t log(o + eps) + (1-t) log(1 - o + eps));
// weight = pos_weight where t == 1 (if pos_weight > 0) else 1; pad labels (-1) give 0 and are not counted.
// sums[0] += sum weight * bce, sums[1] += number of non-pad items.
// dprobs (optional): weight * d bce / d p  (0 outside the clip range and at pads), unscaled.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) masked_bce_kernel(const T *__restrict__ p, const float *__restrict__ labels, float pos_weight,
                                                         float *__restrict__ item_loss, float *__restrict__ sums,
                                                         float *__restrict__ dprobs, int64_t n) {
    __shared__ float buf[4];
    float tot = 0.f, cnt = 0.f;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float t = labels[i];
        float loss = 0.f, d = 0.f;
        if (t != -1.0f) {
            const float pv = (float)p[i];
            const float o = fminf(fmaxf(pv, KERAS_EPS), 1.0f - KERAS_EPS);
            const float w = (pos_weight > 0.f && t == 1.0f) ? pos_weight : 1.0f;
            loss = -(t * logf(o + KERAS_EPS) + (1.0f - t) * logf(1.0f - o + KERAS_EPS)) * w;
            if (pv >= KERAS_EPS && pv <= 1.0f - KERAS_EPS) d = -(t / (o + KERAS_EPS) - (1.0f - t) / (1.0f - o + KERAS_EPS)) * w;
            cnt += 1.f;
        }
        tot += loss;
        if (item_loss) item_loss[i] = loss;
        if (dprobs) dprobs[i] = d;
    }
    tot = ew_block_sum<4>(tot, buf);
    cnt = ew_block_sum<4>(cnt, buf);
    if (threadIdx.x == 0) { atomicAdd(sums, tot); atomicAdd(sums + 1, cnt); }
}

extern "C" int b4c_masked_bce(const void *probs, const float *labels, float pos_weight, float *item_loss, float *sums,
                              float *dprobs, int64_t n, int dtype, void *stream) {
    B4C_REQUIRE(probs && labels && sums && n >= 0, "masked_bce: bad argument");
    if (n == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(n, 256) > 512 ? 512 : ew_grid(n, 256);
    if (dtype == B4C_F32) masked_bce_kernel<float><<<grid, 256, 0, st>>>((const float *)probs, labels, pos_weight, item_loss, sums, dprobs, n);
    else if (dtype == B4C_BF16) masked_bce_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t *)probs, labels, pos_weight, item_loss, sums, dprobs, n);
    else B4C_REQUIRE(false, "masked_bce: dtype %d", dtype);
    return b4c_check_launch("masked_bce");
}

// ------------------------------------------------------------------------------------------
// counts behind PositiveRate / PredictedPositives / F1Score (metrics.py:5-87), one pass:
//   out[0] += sum mask * y_true      out[1] += sum mask            (mask = y_true != -1)
//   out[2] += sum mask * round(y_pred)
//   out[3] += #(int(y_true) == 1 and int(round(y_pred)) == 1)   out[4] += #(int(y_true) == 1)
//   out[5] += #(int(round(y_pred)) == 1)            (tf.round: half to even; F1 counts are NOT masked, as the reference)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) binary_counts_kernel(const float *__restrict__ y_true, const T *__restrict__ y_pred,
                                                            float *__restrict__ out, int64_t n) {
    __shared__ float buf[4];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float t = y_true[i];
        const float r = rintf((float)y_pred[i]);
        const float m = (t != -1.0f) ? 1.f : 0.f;
        acc[0] += m * t;
        acc[1] += m;
        acc[2] += m * r;
        const bool ct = (int)t == 1, pt = (int)r == 1;
        acc[3] += (ct && pt) ? 1.f : 0.f;
        acc[4] += ct ? 1.f : 0.f;
        acc[5] += pt ? 1.f : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float s = ew_block_sum<4>(acc[k], buf);
        if (threadIdx.x == 0) atomicAdd(out + k, s);
    }
}

extern "C" int b4c_binary_counts(const float *y_true, const void *y_pred, float *out6, int64_t n, int dtype, void *stream) {
    B4C_REQUIRE(y_true && y_pred && out6 && n >= 0, "binary_counts: bad argument");
    if (n == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int g0 = ew_grid(n, 256), grid = g0 > 512 ? 512 : g0;
    if (dtype == B4C_F32) binary_counts_kernel<float><<<grid, 256, 0, st>>>(y_true, (const float *)y_pred, out6, n);
    else if (dtype == B4C_BF16) binary_counts_kernel<bf16_t><<<grid, 256, 0, st>>>(y_true, (const bf16_t *)y_pred, out6, n);
    else B4C_REQUIRE(false, "binary_counts: dtype %d", dtype);
    return b4c_check_launch("binary_counts");
}

// ------------------------------------------------------------------------------------------
// label compaction for the sync-free Cloze step: padded (B, M) float labels (-1 = pad, row b's labels are the
// first counts[b] entries, in mask order) -> compact int32 [cap] in the row-major order of b4c_mask_positions;
// entries >= R (= offsets[B]) become -1 (ignored rows) in `out`, and -1 (zero row) in flat_idx when given.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) compact_labels_kernel(const float *__restrict__ labels, int B, int M,
                                                             const int32_t *__restrict__ counts, const int32_t *__restrict__ offsets,
                                                             int32_t *__restrict__ out, int32_t *__restrict__ flat_idx, int32_t cap) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= cap) return;
    const int32_t R = offsets[B];
    if (i >= R) {
        out[i] = -1;
        if (flat_idx) flat_idx[i] = -1;
        return;
    }
    // destination i belongs to the row b with offsets[b] <= i < offsets[b + 1] (rows without a match are skipped)
    int lo = 0, hi = B;            // invariant: offsets[lo] <= i < offsets[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= (int32_t)i) lo = mid; else hi = mid;
    }
    const int m = (int)i - offsets[lo];
    float v = -1.0f;
    if (m < M) v = labels[(int64_t)lo * M + m];
    out[i] = (v == -1.0f) ? -1 : (int32_t)v;
}

extern "C" int b4c_compact_labels(const float *labels, int B, int M, const int32_t *counts, const int32_t *offsets,
                                  int32_t *out, int32_t *flat_idx, int32_t cap, void *stream) {
    B4C_REQUIRE(labels && counts && offsets && out && B > 0 && M > 0 && cap > 0, "compact_labels: bad argument");
    compact_labels_kernel<<<(int)ceil_div64(cap, 256), 256, 0, (hipStream_t)stream>>>(labels, B, M, counts, offsets, out, flat_idx, cap);
    return b4c_check_launch("compact_labels");
}

// ------------------------------------------------------------------------------------------
// attention weights on request (MultiHeadAttention.call's second result, transformer.py:64-97):
//   w[b][h][i][j] = exp(q_i . k_j / sqrt(dh) + pad_j * -1e9 - lse[b][h][i])          fp32 [B][H][S][S]
// 64 x 64 tile per workgroup, q and k rows staged in LDS as fp32.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) attn_weights_kernel(const T *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                           const float *__restrict__ lse, float *__restrict__ w, int S, int H,
                                                           int dh, float sqrt_dk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // q tile [64][dh+1], k tile [64][dh+1]
    const int tid = threadIdx.x;
    const int b = blockIdx.z / H, h = blockIdx.z % H, dm = H * dh;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int64_t tok0 = (int64_t)b * S;
    float *sq = sm, *sk = sm + 64 * (dh + 1);
    for (int e = tid; e < 64 * dh; e += 256) {
        const int r = e / dh, c = e % dh;
        sq[r * (dh + 1) + c] = (i0 + r < S) ? (float)qkv[(tok0 + i0 + r) * ld + h * dh + c] : 0.f;
        sk[r * (dh + 1) + c] = (j0 + r < S) ? (float)qkv[(tok0 + j0 + r) * ld + dm + h * dh + c] : 0.f;
    }
    __syncthreads();
    const int j = tid & 63;
    for (int i = tid >> 6; i < 64; i += 4) {
        if (i0 + i >= S || j0 + j >= S) continue;
        float s = 0.f;
        for (int c = 0; c < dh; ++c) s += sq[i * (dh + 1) + c] * sk[j * (dh + 1) + c];
        s /= sqrt_dk;
        if (key_pad[tok0 + j0 + j]) s += -1e9f;
        const float L = lse[((int64_t)b * H + h) * S + i0 + i];
        w[(((int64_t)b * H + h) * S + i0 + i) * S + j0 + j] = expf(s - L);
    }
}

extern "C" int b4c_attn_weights(const void *qkv, int ld_qkv, const uint8_t *key_pad, const float *lse, float *weights, int B,
                                int S, int H, int dh, int dtype, void *stream) {
    B4C_REQUIRE(qkv && key_pad && lse && weights && B > 0 && S > 0 && H > 0 && dh > 0, "attn_weights: bad argument");
    B4C_REQUIRE(ld_qkv >= 3 * H * dh, "attn_weights: bad pitch");
    B4C_REQUIRE((int64_t)B * H <= 65535, "attn_weights: B*H exceeds 65535");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((S + 63) / 64, (S + 63) / 64, B * H);
    const size_t shm = (size_t)2 * 64 * (dh + 1) * sizeof(float);
    const float sq = sqrtf((float)dh);
    if (dtype == B4C_F32) attn_weights_kernel<float><<<grid, 256, shm, st>>>((const float *)qkv, ld_qkv, key_pad, lse, weights, S, H, dh, sq);
    else if (dtype == B4C_BF16) attn_weights_kernel<bf16_t><<<grid, 256, shm, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, lse, weights, S, H, dh, sq);
    else B4C_REQUIRE(false, "attn_weights: dtype %d", dtype);
    return b4c_check_launch("attn_weights");
}

// ------------------------------------------------------------------------------------------
// dst[n][k] += src[k][n]   (fp32; the tied-weight head's projection gradient [K][V] added into the
// rows of the embedding-table gradient [V][K]); 32 x 32 LDS tile, both sides coalesced.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) transpose_add_kernel(const float *__restrict__ src, int lds_, float *__restrict__ dst, int ldd,
                                                            int K, int N) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        tile[r][tx] = (k < K && n < N) ? src[(int64_t)k * lds_ + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) dst[(int64_t)n * ldd + k] += tile[tx][r];
    }
}

extern "C" int b4c_transpose_add(const float *src, int ld_src, float *dst, int ld_dst, int K, int N, void *stream) {
    B4C_REQUIRE(src && dst && K > 0 && N > 0 && ld_src >= N && ld_dst >= K, "transpose_add: bad argument");
    dim3 grid((N + 31) / 32, (K + 31) / 32);
    transpose_add_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, ld_src, dst, ld_dst, K, N);
    return b4c_check_launch("transpose_add");
}

// ------------------------------------------------------------------------------------------
// row-sparse gradient exchange (config 5: a 2 GB embedding gradient of which <= B*S rows are touched):
//   b4c_rows_gather_f32:      out[r][:] = src[idx[r]][:]                         (idx < 0 -> zeros)
//   b4c_rows_scatter_add_f32: dst[idx[r]][:] += src[r][:]  with float atomics    (idx < 0 skipped)
// width % 4 == 0; one thread = 4 columns.
// ------------------------------------------------------------------------------------------
template <bool SCATTER>
__global__ void __launch_bounds__(256) rows_f32_kernel(const float *__restrict__ src, int ld_src, const int64_t *__restrict__ idx,
                                                       float *__restrict__ dst, int ld_dst, int64_t n, int width) {
    const int cpr = width >> 2;
    const int64_t total = n * cpr;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) << 2;
        const int64_t j = idx[r];
        if (SCATTER) {
            if (j < 0) continue;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + r * ld_src + c);
            float *d = dst + j * ld_dst + c;
            atomicAdd(d, v[0]); atomicAdd(d + 1, v[1]); atomicAdd(d + 2, v[2]); atomicAdd(d + 3, v[3]);
        } else {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j >= 0) v = *reinterpret_cast<const f32x4 *>(src + j * ld_src + c);
            *reinterpret_cast<f32x4 *>(dst + r * ld_dst + c) = v;
        }
    }
}

extern "C" int b4c_rows_gather_f32(const float *src, int ld_src, const int64_t *idx, float *out, int ld_out, int64_t n,
                                   int width, void *stream) {
    B4C_REQUIRE(src && idx && out && n >= 0 && width > 0 && width % 4 == 0 && ld_src % 4 == 0 && ld_out % 4 == 0, "rows_gather_f32: bad shape");
    if (n == 0) return B4C_OK;
    rows_f32_kernel<false><<<ew_grid(n * (width / 4), 256), 256, 0, (hipStream_t)stream>>>(src, ld_src, idx, out, ld_out, n, width);
    return b4c_check_launch("rows_gather_f32");
}

extern "C" int b4c_rows_scatter_add_f32(const float *src, int ld_src, const int64_t *idx, float *dst, int ld_dst, int64_t n,
                                        int width, void *stream) {
    B4C_REQUIRE(src && idx && dst && n >= 0 && width > 0 && width % 4 == 0 && ld_src % 4 == 0, "rows_scatter_add_f32: bad shape");
    if (n == 0) return B4C_OK;
    rows_f32_kernel<true><<<ew_grid(n * (width / 4), 256), 256, 0, (hipStream_t)stream>>>(src, ld_src, idx, dst, ld_dst, n, width);
    return b4c_check_launch("rows_scatter_add_f32");
}

// ------------------------------------------------------------------------------------------
// sampled-softmax head (BASELINE.json configs[4]; north_star extension with NO reference counterpart):
// K shared negatives per step from a log-uniform (Zipfian) sampler WITH replacement over [0, range_max):
//   P(c) = log((c + 2) / (c + 1)) / log(range_max + 1),   expected count Q(c) = K P(c)
// (the distribution of tf.random.log_uniform_candidate_sampler; ids are assumed sorted by decreasing frequency).
// One 24-bit uniform per sample from the counter hash: u = (rand64(seed, i) >> 40) / 2^24,
//   id = min(range_max - 1, floor(exp(u log(range_max + 1))) - 1),  logq = log(K P(id)).
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ float b4c_log_uniform_logq(int64_t c, int64_t range_max, int num_sampled) {
    // log((c + 2) / (c + 1)) as log1p(1 / (c + 1)): the quotient itself is 1 + 5e-7 at c = 2M, below fp32 resolution
    return logf((float)num_sampled * (log1pf(1.0f / ((float)c + 1.0f)) / logf((float)range_max + 1.0f)));
}

__global__ void __launch_bounds__(256) log_uniform_sample_kernel(uint64_t seed, int n, int64_t range_max, int64_t *__restrict__ ids,
                                                                 float *__restrict__ logq) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // double precision: the id is a floor() of an exponential, and a host regenerating the ids must land on the same side
    const double u = (double)(b4c_rand64(seed, (uint64_t)i) >> 40) * (1.0 / 16777216.0);
    int64_t c = (int64_t)floor(exp(u * log((double)range_max + 1.0))) - 1;
    c = c < 0 ? 0 : (c >= range_max ? range_max - 1 : c);
    ids[i] = c;
    if (logq) logq[i] = b4c_log_uniform_logq(c, range_max, n);
}

extern "C" int b4c_log_uniform_sample(uint64_t seed, int n, int64_t range_max, int64_t *ids, float *logq, void *stream) {
    B4C_REQUIRE(ids && n > 0 && range_max > 0, "log_uniform_sample: bad argument");
    log_uniform_sample_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(seed, n, range_max, ids, logq);
    return b4c_check_launch("log_uniform_sample");
}

// out[r] = sum_c a[r][c] * b[r][c]   (fp32 accumulate; one wave per row; width % 8 == 0)
template <typename T>
__global__ void __launch_bounds__(256) row_dot_kernel(const T *__restrict__ a, int lda, const T *__restrict__ b, int ldb,
                                                      float *__restrict__ out, int64_t R, int width) {
    const int lane = threadIdx.x & 63;
    const int cpr = width >> 3;
    for (int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < R; row += (int64_t)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < cpr; c += 64) {
            float x[8], y[8];
            Vec8<T>::load(a + row * lda + c * 8, x);
            Vec8<T>::load(b + row * ldb + c * 8, y);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += x[k] * y[k];
        }
        s = wave_sum(s);
        if (lane == 0) out[row] = s;
    }
}

extern "C" int b4c_row_dot(const void *a, int lda, const void *b, int ldb, float *out, int64_t R, int width, int dtype, void *stream) {
    B4C_REQUIRE(a && b && out && R >= 0 && width > 0 && width % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "row_dot: bad shape");
    if (R == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(R, 4);
    if (dtype == B4C_F32) row_dot_kernel<float><<<grid, 256, 0, st>>>((const float *)a, lda, (const float *)b, ldb, out, R, width);
    else if (dtype == B4C_BF16) row_dot_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t *)a, lda, (const bf16_t *)b, ldb, out, R, width);
    else B4C_REQUIRE(false, "row_dot: dtype %d", dtype);
    return b4c_check_launch("row_dot");
}

// Sampled-softmax cross-entropy with its backward, in place (tf.nn.sampled_softmax_loss semantics, remove_accidental_hits):
//   z_j  = Z[r][j]                      j-th shared negative: h . w_s + b_s - logQ(s), already folded in by the GEMM
//   z_t  = ztrue[r] - logQ(y_r)         true class (ztrue = h . w_y + b_y)
//   negatives with samples[j] == y_r are removed (-inf);   loss_r = logsumexp(z_t, z_.) - z_t
//   Z[r][j] <- gs * softmax_j,  dtrue[r] <- gs * (softmax_t - 1);  rows with label < 0 or >= range_max: loss 0, zero gradient.
template <typename T>
__global__ void __launch_bounds__(256) sampled_ce_kernel(T *__restrict__ Z, int ld, const float *__restrict__ ztrue,
                                                         const int64_t *__restrict__ samples, const int32_t *__restrict__ labels,
                                                         int64_t range_max, float *__restrict__ item_loss, float *__restrict__ dtrue,
                                                         const float *__restrict__ grad_scale, int64_t R, int K) {
    __shared__ float buf[4];
    const int tid = threadIdx.x;
    const int nch = ld >> 3;
    const float gs = grad_scale[0];
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        T *zr = Z + row * ld;
        const int y = labels[row];
        if (y < 0 || y >= range_max) {
            for (int c = tid; c < nch; c += 256) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
                Vec8<T>::store(zr + c * 8, v);
            }
            if (tid == 0) { item_loss[row] = 0.f; dtrue[row] = 0.f; }
            continue;
        }
        const float zt = ztrue[row] - b4c_log_uniform_logq(y, range_max, K);
        float m = zt;
        for (int c = tid; c < nch; c += 256) {
            float v[8];
            Vec8<T>::load(zr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = c * 8 + k;
                if (j < K && samples[j] != y) m = fmaxf(m, v[k]);
            }
        }
        m = wave_max(m);
        __syncthreads();
        if ((tid & 63) == 0) buf[tid >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(buf[0], buf[1]), fmaxf(buf[2], buf[3]));
        float s = 0.f;
        for (int c = tid; c < nch; c += 256) {
            float v[8];
            Vec8<T>::load(zr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = c * 8 + k;
                if (j < K && samples[j] != y) s += expf(v[k] - m);
            }
        }
        s = ew_block_sum<4>(s, buf) + expf(zt - m);
        const float inv = 1.0f / s;
        for (int c = tid; c < nch; c += 256) {
            float v[8];
            Vec8<T>::load(zr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = c * 8 + k;
                v[k] = (j < K && samples[j] != y) ? gs * expf(v[k] - m) * inv : 0.f;
            }
            Vec8<T>::store(zr + c * 8, v);
        }
        if (tid == 0) {
            item_loss[row] = logf(s) + m - zt;
            dtrue[row] = gs * (expf(zt - m) * inv - 1.0f);
        }
        __syncthreads();
    }
}

extern "C" int b4c_sampled_ce_fwd_bwd(void *Z, int ld, const float *ztrue, const int64_t *samples, const int32_t *labels,
                                      int64_t range_max, float *item_loss, float *dtrue, const float *grad_scale, int64_t R,
                                      int K, int dtype, void *stream) {
    B4C_REQUIRE(Z && ztrue && samples && labels && item_loss && dtrue && grad_scale && R >= 0 && K > 0 && range_max > 0,
                "sampled_ce_fwd_bwd: bad argument");
    B4C_REQUIRE(ld % 8 == 0 && ld >= K, "sampled_ce_fwd_bwd: pitch");
    if (R == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)(R < 4096 ? R : 4096);
    if (dtype == B4C_F32) sampled_ce_kernel<float><<<grid, 256, 0, st>>>((float *)Z, ld, ztrue, samples, labels, range_max, item_loss, dtrue, grad_scale, R, K);
    else if (dtype == B4C_BF16) sampled_ce_kernel<bf16_t><<<grid, 256, 0, st>>>((bf16_t *)Z, ld, ztrue, samples, labels, range_max, item_loss, dtrue, grad_scale, R, K);
    else B4C_REQUIRE(false, "sampled_ce_fwd_bwd: dtype %d", dtype);
    return b4c_check_launch("sampled_ce_fwd_bwd");
}

// dst[idx[i]] += src[i]  (fp32, float atomics; idx < 0 skipped): bias gradients of the sampled classes
__global__ void __launch_bounds__(256) scatter_add_1d_kernel(const float *__restrict__ src, const int64_t *__restrict__ idx,
                                                             float *__restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        if (idx[i] >= 0) atomicAdd(dst + idx[i], src[i]);
}

extern "C" int b4c_scatter_add_1d(const float *src, const int64_t *idx, float *dst, int64_t n, void *stream) {
    B4C_REQUIRE(src && idx && dst && n >= 0, "scatter_add_1d: bad argument");
    if (n == 0) return B4C_OK;
    scatter_add_1d_kernel<<<ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>(src, idx, dst, n);
    return b4c_check_launch("scatter_add_1d");
}

// out[r][:] = scale[r] * src[r][:]  (fp32 out; T in): the true-class rows' gradient dtrue_r * h_r
template <typename T>
__global__ void __launch_bounds__(256) row_scale_kernel(const T *__restrict__ src, int ld, const float *__restrict__ scale,
                                                        float *__restrict__ out, int ld_out, int64_t R, int width) {
    const int cpr = width >> 3;
    const int64_t total = R * cpr;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) << 3;
        float v[8];
        Vec8<T>::load(src + r * ld + c, v);
        const float s = scale[r];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= s;
        Vec8<float>::store(out + r * ld_out + c, v);
    }
}

extern "C" int b4c_row_scale_f32(const void *src, int ld, const float *scale, float *out, int ld_out, int64_t R, int width,
                                 int dtype, void *stream) {
    B4C_REQUIRE(src && scale && out && R >= 0 && width > 0 && width % 8 == 0 && ld % 8 == 0 && ld_out % 8 == 0, "row_scale_f32: bad shape");
    if (R == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(R * (width / 8), 256);
    if (dtype == B4C_F32) row_scale_kernel<float><<<grid, 256, 0, st>>>((const float *)src, ld, scale, out, ld_out, R, width);
    else if (dtype == B4C_BF16) row_scale_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t *)src, ld, scale, out, ld_out, R, width);
    else B4C_REQUIRE(false, "row_scale_f32: dtype %d", dtype);
    return b4c_check_launch("row_scale_f32");
}

// ------------------------------------------------------------------------------------------------------------------
// Small fused helpers of the loss's scalar bookkeeping (losses.py:80-98: mean over the non-pad rows).  Each replaces a
// chain of 3-12 one-element PyTorch kernels (~5 us apiece on the step's critical path).
// ------------------------------------------------------------------------------------------------------------------
// out[0] = 1 / n_valid (0 when no row is valid), out[1] = n_valid;  valid = 0 <= label < V.  One workgroup, fixed order.
__global__ void __launch_bounds__(1024) label_scale_kernel(const int32_t *__restrict__ labels, int64_t R, int V, float *__restrict__ out) {
    __shared__ int part[16];
    int c = 0;
    const int64_t R4 = ((reinterpret_cast<uintptr_t>(labels) & 15) == 0) ? (R >> 2) : 0;      // 16-byte loads over the aligned body
    for (int64_t i = threadIdx.x; i < R4; i += 1024) {
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const i32x4 y4 = reinterpret_cast<const i32x4 *>(labels)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) c += (y4[k] >= 0 && y4[k] < V) ? 1 : 0;
    }
    for (int64_t i = R4 * 4 + threadIdx.x; i < R; i += 1024) {
        const int y = labels[i];
        c += (y >= 0 && y < V) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (int w = 0; w < 16; ++w) n += part[w];
        out[0] = n > 0 ? 1.0f / (float)n : 0.f;
        out[1] = (float)n;
    }
}

extern "C" int b4c_label_scale(const int32_t *labels, int64_t R, int V, float *out, void *stream) {
    B4C_REQUIRE(out && R >= 0 && V > 0 && (labels || R == 0), "label_scale: bad argument");
    label_scale_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(labels, R, V, out);
    return b4c_check_launch("label_scale");
}

// out[0] = scale[0] * sum_r item[r]  (one workgroup, fixed order: repeatable); NaN when *poison < 0 (a device-side
// consistency flag, e.g. the token count of the packed layout)
__global__ void __launch_bounds__(1024) sum_scaled_kernel(const float *__restrict__ item, int64_t R, const float *__restrict__ scale,
                                                          const int32_t *__restrict__ poison, float *__restrict__ out) {
    __shared__ float part[16];
    float s = 0.f;
    const int64_t R4 = ((reinterpret_cast<uintptr_t>(item) & 15) == 0) ? (R >> 2) : 0;
    for (int64_t i = threadIdx.x; i < R4; i += 1024) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(item)[i];
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (int64_t i = R4 * 4 + threadIdx.x; i < R; i += 1024) s += item[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w];
        t *= scale[0];
        if (poison && poison[0] < 0) t = __uint_as_float(0x7fc00000u);
        out[0] = t;
    }
}

extern "C" int b4c_sum_scaled(const float *item, int64_t R, const float *scale, const int32_t *poison, float *out, void *stream) {
    B4C_REQUIRE(scale && out && R >= 0 && (item || R == 0), "sum_scaled: bad argument");
    sum_scaled_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(item, R, scale, poison, out);
    return b4c_check_launch("sum_scaled");
}

// dh_out = g * dh (bf16 rows of K), rowscal_out = rowscal with its gradient-linear columns (c, nb, yd) times g: the upstream gradient folded into what
// b4c_vocab_ce_fwd left for the backward (both are linear in it), in one launch; the inputs stay as they are.
__global__ void __launch_bounds__(256) vce_apply_grad_kernel(const bf16_t *__restrict__ dh, int ld, const float *__restrict__ rowscal,
                                                             const float *__restrict__ g, bf16_t *__restrict__ dh_out, int ld_out,
                                                             float *__restrict__ rowscal_out, int64_t R, int K) {
    const float gv = g[0];
    const int cpr = K >> 3;
    const int64_t total = R * (cpr + 1);
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (cpr + 1);
        const int c = (int)(i - r * (cpr + 1));
        if (c < cpr) {
            float v[8];
            Vec8<bf16_t>::load(dh + r * ld + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= gv;
            Vec8<bf16_t>::store(dh_out + r * ld_out + c * 8, v);
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(rowscal + r * 8), b = *reinterpret_cast<const f32x4 *>(rowscal + r * 8 + 4);
            // [lse2, c, nb, lo | yd, hi, -, -]: c, nb, yd are linear in the gradient scale; lo / hi are the clip range
            *reinterpret_cast<f32x4 *>(rowscal_out + r * 8) = (f32x4){a[0], a[1] * gv, a[2] * gv, a[3]};
            *reinterpret_cast<f32x4 *>(rowscal_out + r * 8 + 4) = (f32x4){b[0] * gv, b[1], b[2], b[3]};
        }
    }
}

extern "C" int b4c_vocab_ce_apply_grad(const void *dh, int ld, const float *rowscal, const float *g, void *dh_out, int ld_out,
                                       float *rowscal_out, int64_t R, int K, void *stream) {
    B4C_REQUIRE(dh && rowscal && g && dh_out && rowscal_out && R >= 0 && K > 0 && K % 8 == 0 && ld % 8 == 0 && ld_out % 8 == 0,
                "vocab_ce_apply_grad: bad argument");
    if (R == 0) return B4C_OK;
    vce_apply_grad_kernel<<<ew_grid(R * (K / 8 + 1), 256), 256, 0, (hipStream_t)stream>>>((const bf16_t *)dh, ld, rowscal, g, (bf16_t *)dh_out,
                                                                                    ld_out, rowscal_out, R, K);
    return b4c_check_launch("vocab_ce_apply_grad");
}

// out = act > 0 ? g : 0 (elementwise, n a multiple of 8): the ReLU of the head's trunk output applied to the gradient that
// arrives from the vocabulary head
template <typename T>
__global__ void __launch_bounds__(256) relu_gate_kernel(const T *__restrict__ g, const T *__restrict__ act, T *__restrict__ out, int64_t n8) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float a[8], b[8];
        Vec8<T>::load(g + i * 8, a);
        Vec8<T>::load(act + i * 8, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = b[k] > 0.f ? a[k] : 0.f;
        Vec8<T>::store(out + i * 8, a);
    }
}

extern "C" int b4c_relu_gate(const void *g, const void *act, void *out, int64_t n, int dtype, void *stream) {
    B4C_REQUIRE(g && act && out && n >= 0 && n % 8 == 0, "relu_gate: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_F32) relu_gate_kernel<float><<<ew_grid(n / 8, 256), 256, 0, st>>>((const float *)g, (const float *)act, (float *)out, n / 8);
    else if (dtype == B4C_BF16) relu_gate_kernel<bf16_t><<<ew_grid(n / 8, 256), 256, 0, st>>>((const bf16_t *)g, (const bf16_t *)act, (bf16_t *)out, n / 8);
    else B4C_REQUIRE(false, "relu_gate: dtype %d", dtype);
    return b4c_check_launch("relu_gate");
}

// dst[idx[r]][:] += src[r][:]  for idx[r] >= 0 (distinct indices: no atomics): the gradient of the masked query rows
// joins the gradient of all token rows (b4c_attn_mq_bwd's layer).  src may be fp32 beside a bf16 dst: one rounding less.
template <typename T, typename TS>
__global__ void __launch_bounds__(256) rows_add_kernel(T *__restrict__ dst, int ld_dst, const int32_t *__restrict__ idx,
                                                       const TS *__restrict__ src, int ld_src, int64_t n_src, int width) {
    const int cpr = width >> 3;
    const int64_t total = n_src * cpr;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) << 3;
        const int t = idx[r];
        if (t < 0) continue;
        float a[8], b[8];
        Vec8<T>::load(dst + (int64_t)t * ld_dst + c, a);
        Vec8<TS>::load(src + r * ld_src + c, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += b[k];
        Vec8<T>::store(dst + (int64_t)t * ld_dst + c, a);
    }
}

extern "C" int b4c_rows_add(void *dst, int ld_dst, const int32_t *idx, const void *src, int ld_src, int64_t n_src, int width,
                            int dtype, int src_dtype, void *stream) {
    B4C_REQUIRE(dst && idx && src && n_src >= 0 && width > 0 && width % 8 == 0 && ld_dst % 8 == 0 && ld_src % 8 == 0,
                "rows_add: bad shape (width %d)", width);
    B4C_REQUIRE(src_dtype == dtype || src_dtype == B4C_F32, "rows_add: src dtype %d beside dst dtype %d", src_dtype, dtype);
    if (n_src == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(n_src * (width / 8), 256);
    if (dtype == B4C_F32) rows_add_kernel<float, float><<<grid, 256, 0, st>>>((float *)dst, ld_dst, idx, (const float *)src, ld_src, n_src, width);
    else if (dtype == B4C_BF16 && src_dtype == B4C_BF16)
        rows_add_kernel<bf16_t, bf16_t><<<grid, 256, 0, st>>>((bf16_t *)dst, ld_dst, idx, (const bf16_t *)src, ld_src, n_src, width);
    else if (dtype == B4C_BF16)
        rows_add_kernel<bf16_t, float><<<grid, 256, 0, st>>>((bf16_t *)dst, ld_dst, idx, (const float *)src, ld_src, n_src, width);
    else B4C_REQUIRE(false, "rows_add: dtype %d", dtype);
    return b4c_check_launch("rows_add");
}

// x[rows][width] := NaN when the device flag is negative, untouched (not even read) otherwise: how a scoring path reports a
// caller-given count that disagrees with the device's own (include/b4c.h b4c_poison_rows) without a read-back
template <typename T>
__global__ void __launch_bounds__(256) poison_rows_kernel(T *__restrict__ x, int ld, int64_t rows, int width, const int32_t *__restrict__ flag) {
    if (flag[0] >= 0) return;
    const int64_t total = rows * width;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / width;
        if constexpr (std::is_same<T, int32_t>::value) x[r * ld + (i - r * width)] = -1;
        else x[r * ld + (i - r * width)] = (T)__builtin_nanf("");
    }
}

extern "C" int b4c_poison_rows(void *x, int ld, int64_t rows, int width, const int32_t *flag, int dtype, void *stream) {
    B4C_REQUIRE(x && flag && rows >= 0 && width > 0 && ld >= width, "poison_rows: bad argument");
    if (rows == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(rows * width, 256);
    if (dtype == B4C_F32) poison_rows_kernel<float><<<grid, 256, 0, st>>>((float *)x, ld, rows, width, flag);
    else if (dtype == B4C_BF16) poison_rows_kernel<bf16_t><<<grid, 256, 0, st>>>((bf16_t *)x, ld, rows, width, flag);
    else if (dtype == B4C_I32) poison_rows_kernel<int32_t><<<grid, 256, 0, st>>>((int32_t *)x, ld, rows, width, flag);
    else B4C_REQUIRE(false, "poison_rows: dtype %d", dtype);
    return b4c_check_launch("poison_rows");
}

// ------------------------------------------------------------------------------------------------------------------
// Stable LSD radix sort of token positions by table row (the order the sorted embedding backward walks): 8-bit digits,
// one wave per 1024-key chunk, per pass a histogram, two scan and a scatter launch.  order[i] = position of the i-th
// smallest id (ties in position order) -- what torch.sort(stable) returned through 14 launches.
//   ids int64 [n] (clamped to [0, n_rows - 1] as the embedding kernels clamp them); workspace: see b4c_sort_ids_workspace_bytes
// ------------------------------------------------------------------------------------------------------------------
#define SORT_CHUNK 1024

__device__ __forceinline__ unsigned sort_key(const int64_t *ids, const int32_t *pos_in, int64_t i, int n_rows) {
    const int64_t p = pos_in ? pos_in[i] : i;
    int64_t v = ids[p];
    v = v < 0 ? 0 : (v >= n_rows ? n_rows - 1 : v);
    return (unsigned)v;
}

__global__ void __launch_bounds__(64) sort_hist_kernel(const int64_t *__restrict__ ids, const int32_t *__restrict__ pos_in, int64_t n,
                                                       int n_rows, int shift, int nblocks, int32_t *__restrict__ hist) {
    __shared__ int cnt[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) cnt[d] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t i0 = (int64_t)blockIdx.x * SORT_CHUNK;
    for (int g = 0; g < SORT_CHUNK; g += 64) {
        const int64_t i = i0 + g + lane;
        if (i < n) atomicAdd(&cnt[(sort_key(ids, pos_in, i, n_rows) >> shift) & 255], 1);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int d = lane; d < 256; d += 64) hist[(int64_t)d * nblocks + blockIdx.x] = cnt[d];      // digit-major: one scan orders it
}

// exclusive prefix sum over hist[256][nblocks] (digit-major) in place, in two launches of 256 workgroups: each digit's row
// is scanned on its own and leaves its total; then every row adds the totals of the digits before it.  (One workgroup
// walking all 256 * nblocks entries took 180 us.)
__global__ void __launch_bounds__(256) sort_rowscan_kernel(int32_t *__restrict__ hist, int nblocks, int32_t *__restrict__ totals) {
    __shared__ int part[256];
    const int tid = threadIdx.x;
    int32_t *row = hist + (int64_t)blockIdx.x * nblocks;
    const int per = (nblocks + 255) / 256;
    const int lo = tid * per, hi = min(nblocks, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += row[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = lo; i < hi; ++i) {
        const int c = row[i];
        row[i] = run;
        run += c;
    }
    if (tid == 255) totals[blockIdx.x] = part[255];
}
__global__ void __launch_bounds__(256) sort_addbase_kernel(int32_t *__restrict__ hist, int nblocks, const int32_t *__restrict__ totals) {
    __shared__ int sbase;
    if (threadIdx.x < 64) {
        int s = 0;
        for (int d = threadIdx.x; d < (int)blockIdx.x; d += 64) s += totals[d];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (threadIdx.x == 0) sbase = s;
    }
    __syncthreads();
    const int base = sbase;
    int32_t *row = hist + (int64_t)blockIdx.x * nblocks;
    for (int i = threadIdx.x; i < nblocks; i += 256) row[i] += base;
}

__global__ void __launch_bounds__(64) sort_scatter_kernel(const int64_t *__restrict__ ids, const int32_t *__restrict__ pos_in, int64_t n,
                                                          int n_rows, int shift, int nblocks, const int32_t *__restrict__ offs,
                                                          int32_t *__restrict__ pos_out) {
    __shared__ int base[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) base[d] = offs[(int64_t)d * nblocks + blockIdx.x];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int64_t i0 = (int64_t)blockIdx.x * SORT_CHUNK;
    for (int g = 0; g < SORT_CHUNK; g += 64) {
        const int64_t i = i0 + g + lane;
        const bool live = i < n;
        const unsigned digit = live ? ((sort_key(ids, pos_in, i, n_rows) >> shift) & 255u) : 256u;      // 256: matches nobody live
        // lanes with the same digit: eight ballots (one per bit), then the dead-lane bit
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? m : ~m;
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int rank = __popcll(same & lt), count = __popcll(same);
        int b0 = 0;
        if (live) b0 = base[digit];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (live) {
            pos_out[b0 + rank] = pos_in ? pos_in[i] : (int32_t)i;
            if (rank == count - 1) base[digit] = b0 + count;        // the group's last lane moves its digit's cursor
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

static int sort_passes(int n_rows) {
    int bits = 1;
    while ((1ll << bits) < n_rows) ++bits;
    return (bits + 7) / 8;
}
extern "C" int64_t b4c_sort_ids_workspace_bytes(int64_t n, int n_rows) {
    if (n <= 0) return 0;
    const int64_t nblocks = (n + SORT_CHUNK - 1) / SORT_CHUNK;
    return (256 * nblocks + 256 + n) * 4;           // histogram + digit totals + one ping-pong position array
}

extern "C" int b4c_sort_ids(const int64_t *ids, int64_t n, int n_rows, int32_t *order, void *workspace, int64_t workspace_bytes,
                            void *stream) {
    B4C_REQUIRE(n >= 0 && n_rows > 0 && n < (1ll << 31), "sort_ids: n=%lld n_rows=%d", (long long)n, n_rows);
    if (n == 0) return B4C_OK;
    B4C_REQUIRE(ids && order && workspace && workspace_bytes >= b4c_sort_ids_workspace_bytes(n, n_rows) && (((uintptr_t)workspace) & 3) == 0,
                "sort_ids: null pointer / workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nblocks = (int)((n + SORT_CHUNK - 1) / SORT_CHUNK);
    int32_t *hist = (int32_t *)workspace;
    int32_t *totals = hist + 256 * (int64_t)nblocks;
    int32_t *tmp = totals + 256;
    const int passes = sort_passes(n_rows);
    // the last pass must land in `order`: with an even number of passes the first goes to tmp
    int32_t *dst = (passes % 2) ? order : tmp;
    const int32_t *src = nullptr;                    // pass 0 reads positions 0..n-1 implicitly
    for (int p = 0; p < passes; ++p) {
        sort_hist_kernel<<<nblocks, 64, 0, st>>>(ids, src, n, n_rows, 8 * p, nblocks, hist);
        sort_rowscan_kernel<<<256, 256, 0, st>>>(hist, nblocks, totals);
        sort_addbase_kernel<<<256, 256, 0, st>>>(hist, nblocks, totals);
        sort_scatter_kernel<<<nblocks, 64, 0, st>>>(ids, src, n, n_rows, 8 * p, nblocks, hist, dst);
        src = dst;
        dst = (dst == order) ? tmp : order;
    }
    return b4c_check_launch("sort_ids");
}

// out[i] = src[idx[i]] (int64 values, int32 indices): the ids of the packed tokens for the embedding backward
__global__ void __launch_bounds__(256) gather_i64_kernel(const int64_t *__restrict__ src, const int32_t *__restrict__ idx,
                                                         int64_t *__restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = src[idx[i]];
}

extern "C" int b4c_gather_i64(const int64_t *src, const int32_t *idx, int64_t *out, int64_t n, void *stream) {
    B4C_REQUIRE(n >= 0 && (n == 0 || (src && idx && out)), "gather_i64: bad argument");
    if (n == 0) return B4C_OK;
    gather_i64_kernel<<<ew_grid(n, 256), 256, 0, (hipStream_t)stream>>>(src, idx, out, n);
    return b4c_check_launch("gather_i64");
}

// ---- (ABI 7) zero fill and id chaining: the last two things the training step asked PyTorch kernels for -------------
extern "C" int b4c_zero(void *p, int64_t nbytes, void *stream) {
    B4C_REQUIRE(nbytes >= 0 && (nbytes == 0 || p), "zero: bad argument");
    if (nbytes == 0) return B4C_OK;
    if (hipMemsetAsync(p, 0, (size_t)nbytes, (hipStream_t)stream) != hipSuccess) {
        b4c_set_error("zero: hipMemsetAsync of %lld bytes failed", (long long)nbytes);
        return B4C_ELAUNCH;
    }
    return B4C_OK;
}

// out[b] = [cls, sep, seq_0[b], sep, seq_1[b], sep, ...]   (TransformerInputPrep._chain_sequences, clickstream_transformer.py:38-63)
#define CHAIN_MAX 8
struct ChainArgs {
    const int64_t *seq[CHAIN_MAX];
    int len[CHAIN_MAX], pitch[CHAIN_MAX], start[CHAIN_MAX];      // start: first output column of the sequence
    int n, B, S, ld_out;
    int64_t cls, sep;
};
__global__ void __launch_bounds__(256) chain_ids_kernel(ChainArgs a, int64_t *__restrict__ out) {
    const int64_t total = (int64_t)a.B * a.S;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / a.S), c = (int)(i % a.S);
        int64_t v = c == 0 ? a.cls : a.sep;
#pragma unroll
        for (int s = 0; s < CHAIN_MAX; ++s)
            if (s < a.n && c >= a.start[s] && c < a.start[s] + a.len[s]) v = a.seq[s][(int64_t)b * a.pitch[s] + (c - a.start[s])];
        out[(int64_t)b * a.ld_out + c] = v;
    }
}
extern "C" int b4c_chain_ids(const int64_t *const *seqs, const int *lens, const int *pitches, int n_seq, int B, int64_t cls,
                             int64_t sep, int64_t *out, int ld_out, void *stream) {
    B4C_REQUIRE(seqs && lens && pitches && out && n_seq >= 1 && n_seq <= CHAIN_MAX && B >= 0, "chain_ids: bad argument (1..%d sequences)", CHAIN_MAX);
    ChainArgs a = {};
    int pos = 2;
    for (int s = 0; s < n_seq; ++s) {
        B4C_REQUIRE(lens[s] >= 0 && pitches[s] >= lens[s] && (lens[s] == 0 || B == 0 || seqs[s]), "chain_ids: sequence %d", s);
        a.seq[s] = seqs[s]; a.len[s] = lens[s]; a.pitch[s] = pitches[s]; a.start[s] = pos;
        pos += lens[s] + 1;
    }
    B4C_REQUIRE(ld_out >= pos, "chain_ids: ld_out %d < %d columns", ld_out, pos);
    a.n = n_seq; a.B = B; a.S = pos; a.ld_out = ld_out; a.cls = cls; a.sep = sep;
    if (B == 0) return B4C_OK;
    chain_ids_kernel<<<ew_grid((int64_t)B * pos, 256), 256, 0, (hipStream_t)stream>>>(a, out);
    return b4c_check_launch("chain_ids");
}
