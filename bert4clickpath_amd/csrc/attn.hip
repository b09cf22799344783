// Multi-head self-attention with a key-side padding mask (reference: transformer.py:64-97,
// 130-156).  qkv rows are [q | k | v] blocks of d_model columns; head h owns columns
// [h*dh, (h+1)*dh) of each block, so no split_heads / merge transposes ever touch HBM.
//
// "row" kernels (this file, both dtypes, fp32 math): one thread owns one query (forward, dQ)
// or one key (dK/dV); the other side streams through LDS in 64-row tiles and is read with
// wave-uniform (broadcast) ds_read_b128.  Scores never leave registers; softmax is online in
// forward and recomputed from the saved log-sum-exp in backward.  These are the exact fp32
// parity kernels; the bf16 MFMA kernels (attn_mfma.hip) replace them on the throughput path.
#include <math.h>

#include "common.h"

#define ATT_TILE 64

template <typename T, int DH>
__device__ __forceinline__ void load_row(const T *__restrict__ p, float (&v)[DH]) {
#pragma unroll
    for (int c = 0; c < DH; c += 8) {
        float t[8];
        Vec8<T>::load(p + c, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[c + k] = t[k];
    }
}
template <typename T, int DH>
__device__ __forceinline__ void store_row(T *__restrict__ p, const float (&v)[DH]) {
#pragma unroll
    for (int c = 0; c < DH; c += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[c + k];
        Vec8<T>::store(p + c, t);
    }
}
// stage rows [r0, r0+64) x DH columns starting at column col0 of a [B*S][ld] matrix into fp32 LDS
template <typename T, int DH>
__device__ __forceinline__ void stage_tile(const T *__restrict__ base, int ld, int64_t tok0, int r0, int S, int col0,
                                           float (*s)[DH], int tid) {
    constexpr int CPR = DH / 8;
    for (int c = tid; c < ATT_TILE * CPR; c += 256) {
        const int row = c / CPR, part = c % CPR;
        float t[8];
        if (r0 + row < S) Vec8<T>::load(base + (tok0 + r0 + row) * ld + col0 + part * 8, t);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s[row][part * 8 + k] = t[k];
    }
}
template <int DH> __device__ __forceinline__ float dot_lds(const float (&a)[DH], const float *__restrict__ row) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4 *>(row + d);
        s += a[d] * kv[0];
        s += a[d + 1] * kv[1];
        s += a[d + 2] * kv[2];
        s += a[d + 3] * kv[3];
    }
    return s;
}

template <typename T, int DH>
__global__ void __launch_bounds__(256) attn_fwd_row_kernel(const T *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                           T *__restrict__ o, int ld_o, float *__restrict__ lse, int S, int H,
                                                           float sqrt_dk) {
    __shared__ __attribute__((aligned(16))) float sK[ATT_TILE][DH];
    __shared__ __attribute__((aligned(16))) float sV[ATT_TILE][DH];
    __shared__ uint8_t sPad[ATT_TILE];
    const int tid = threadIdx.x;
    const int b = blockIdx.y / H, h = blockIdx.y % H, dm = H * DH;
    const int64_t tok0 = (int64_t)b * S;
    const int qi = blockIdx.x * 256 + tid;
    const bool active = qi < S;
    float q[DH], acc[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) { q[d] = 0.f; acc[d] = 0.f; }
    if (active) load_row<T, DH>(qkv + (tok0 + qi) * ld + h * DH, q);
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < S; k0 += ATT_TILE) {
        __syncthreads();
        stage_tile<T, DH>(qkv, ld, tok0, k0, S, dm + h * DH, sK, tid);
        stage_tile<T, DH>(qkv, ld, tok0, k0, S, 2 * dm + h * DH, sV, tid);
        if (tid < ATT_TILE) sPad[tid] = (k0 + tid < S) ? key_pad[tok0 + k0 + tid] : 1;
        __syncthreads();
        if (!active) continue;
        const int nk = min(ATT_TILE, S - k0);
        for (int j = 0; j < nk; ++j) {
            float s = dot_lds<DH>(q, &sK[j][0]) / sqrt_dk;
            if (sPad[j]) s += -1e9f;
            if (s > m) {
                const float corr = expf(m - s);
                l *= corr;
#pragma unroll
                for (int d = 0; d < DH; ++d) acc[d] *= corr;
                m = s;
            }
            const float p = expf(s - m);
            l += p;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const f32x4 vv = *reinterpret_cast<const f32x4 *>(&sV[j][d]);
                acc[d] += p * vv[0];
                acc[d + 1] += p * vv[1];
                acc[d + 2] += p * vv[2];
                acc[d + 3] += p * vv[3];
            }
        }
    }
    if (active) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] *= inv;
        store_row<T, DH>(o + (tok0 + qi) * ld_o + h * DH, acc);
        if (lse) lse[((int64_t)b * H + h) * S + qi] = m + logf(l);
    }
}

// dQ: thread per query; also writes delta[q] = sum_d dO[q][d] * O[q][d]
template <typename T, int DH>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const T *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                          const T *__restrict__ o, int ld_o, const T *__restrict__ d_o, int ld_do,
                                                          const float *__restrict__ lse, float *__restrict__ delta,
                                                          T *__restrict__ dqkv, int ld_dq, int S, int H, float sqrt_dk) {
    __shared__ __attribute__((aligned(16))) float sK[ATT_TILE][DH];
    __shared__ __attribute__((aligned(16))) float sV[ATT_TILE][DH];
    __shared__ uint8_t sPad[ATT_TILE];
    const int tid = threadIdx.x;
    const int b = blockIdx.y / H, h = blockIdx.y % H, dm = H * DH;
    const int64_t tok0 = (int64_t)b * S;
    const int qi = blockIdx.x * 256 + tid;
    const bool active = qi < S;
    float q[DH], g[DH], dq[DH];
    float dlt = 0.f, L = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) { q[d] = 0.f; g[d] = 0.f; dq[d] = 0.f; }
    if (active) {
        load_row<T, DH>(qkv + (tok0 + qi) * ld + h * DH, q);
        load_row<T, DH>(d_o + (tok0 + qi) * ld_do + h * DH, g);
        float ov[DH];
        load_row<T, DH>(o + (tok0 + qi) * ld_o + h * DH, ov);
#pragma unroll
        for (int d = 0; d < DH; ++d) dlt += g[d] * ov[d];
        L = lse[((int64_t)b * H + h) * S + qi];
        delta[((int64_t)b * H + h) * S + qi] = dlt;
    }
    for (int k0 = 0; k0 < S; k0 += ATT_TILE) {
        __syncthreads();
        stage_tile<T, DH>(qkv, ld, tok0, k0, S, dm + h * DH, sK, tid);
        stage_tile<T, DH>(qkv, ld, tok0, k0, S, 2 * dm + h * DH, sV, tid);
        if (tid < ATT_TILE) sPad[tid] = (k0 + tid < S) ? key_pad[tok0 + k0 + tid] : 1;
        __syncthreads();
        if (!active) continue;
        const int nk = min(ATT_TILE, S - k0);
        for (int j = 0; j < nk; ++j) {
            if (sPad[j]) continue;  // p == 0 exactly (exp(-1e9 - lse))
            const float s = dot_lds<DH>(q, &sK[j][0]) / sqrt_dk;
            const float p = expf(s - L);
            const float dp = dot_lds<DH>(g, &sV[j][0]);
            const float ds = p * (dp - dlt);
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const f32x4 kv = *reinterpret_cast<const f32x4 *>(&sK[j][d]);
                dq[d] += ds * kv[0];
                dq[d + 1] += ds * kv[1];
                dq[d + 2] += ds * kv[2];
                dq[d + 3] += ds * kv[3];
            }
        }
    }
    if (active) {
#pragma unroll
        for (int d = 0; d < DH; ++d) dq[d] /= sqrt_dk;
        store_row<T, DH>(dqkv + (tok0 + qi) * ld_dq + h * DH, dq);
    }
}

// dK, dV: thread per key; queries (q, dO, lse, delta) stream through LDS
template <typename T, int DH>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(const T *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                           const T *__restrict__ d_o, int ld_do, const float *__restrict__ lse,
                                                           const float *__restrict__ delta, T *__restrict__ dqkv, int ld_dq, int S,
                                                           int H, float sqrt_dk) {
    __shared__ __attribute__((aligned(16))) float sQ[ATT_TILE][DH];
    __shared__ __attribute__((aligned(16))) float sG[ATT_TILE][DH];
    __shared__ float sL[ATT_TILE], sD[ATT_TILE];
    const int tid = threadIdx.x;
    const int b = blockIdx.y / H, h = blockIdx.y % H, dm = H * DH;
    const int64_t tok0 = (int64_t)b * S;
    const int kj = blockIdx.x * 256 + tid;
    const bool inrange = kj < S;
    const bool active = inrange && !key_pad[tok0 + kj];
    float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    if (active) {
        load_row<T, DH>(qkv + (tok0 + kj) * ld + dm + h * DH, k);
        load_row<T, DH>(qkv + (tok0 + kj) * ld + 2 * dm + h * DH, v);
    }
    for (int q0 = 0; q0 < S; q0 += ATT_TILE) {
        __syncthreads();
        stage_tile<T, DH>(qkv, ld, tok0, q0, S, h * DH, sQ, tid);
        stage_tile<T, DH>(d_o, ld_do, tok0, q0, S, h * DH, sG, tid);
        if (tid < ATT_TILE) {
            const bool ok = q0 + tid < S;
            sL[tid] = ok ? lse[((int64_t)b * H + h) * S + q0 + tid] : 0.f;
            sD[tid] = ok ? delta[((int64_t)b * H + h) * S + q0 + tid] : 0.f;
        }
        __syncthreads();
        if (!active) continue;
        const int nq = min(ATT_TILE, S - q0);
        for (int i = 0; i < nq; ++i) {
            const float s = dot_lds<DH>(k, &sQ[i][0]) / sqrt_dk;
            const float p = expf(s - sL[i]);
            const float dp = dot_lds<DH>(v, &sG[i][0]);
            const float ds = p * (dp - sD[i]);
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const f32x4 gv = *reinterpret_cast<const f32x4 *>(&sG[i][d]);
                const f32x4 qv = *reinterpret_cast<const f32x4 *>(&sQ[i][d]);
                dv[d] += p * gv[0];
                dv[d + 1] += p * gv[1];
                dv[d + 2] += p * gv[2];
                dv[d + 3] += p * gv[3];
                dk[d] += ds * qv[0];
                dk[d + 1] += ds * qv[1];
                dk[d + 2] += ds * qv[2];
                dk[d + 3] += ds * qv[3];
            }
        }
    }
    if (inrange) {
#pragma unroll
        for (int d = 0; d < DH; ++d) dk[d] /= sqrt_dk;
        store_row<T, DH>(dqkv + (tok0 + kj) * ld_dq + dm + h * DH, dk);
        store_row<T, DH>(dqkv + (tok0 + kj) * ld_dq + 2 * dm + h * DH, dv);
    }
}

// bf16 MFMA forward/backward (attn_mfma.hip); return B4C_EUNSUPPORTED when the shape is not covered.
int b4c_attn_fwd_mfma(const void *qkv, int ld_qkv, const uint8_t *key_pad, void *o, int ld_o, float *lse, int B, int S,
                      int H, int dh, const int32_t *cu, hipStream_t st);
int b4c_attn_bwd_mfma(const void *qkv, int ld_qkv, const uint8_t *key_pad, const void *o, int ld_o, const void *d_o,
                      int ld_do, const float *lse, float *delta, void *dqkv, int ld_dqkv, int B, int S, int H, int dh,
                      void *workspace, int64_t workspace_bytes, const int32_t *cu, hipStream_t st);
int64_t b4c_attn_bwd_mfma_workspace_bytes(int B, int S, int H, int dh);

// bf16 shapes the MFMA kernels do not cover (S > 512 or head depth 16 / 128) run on the fp32-math row kernels, which are
// several times slower: say so once per process instead of degrading silently.
static void note_row_fallback(const char *who, int S, int dh) {
    static bool said = false;
    if (said) return;
    said = true;
    fprintf(stderr, "[b4c] %s: bf16 attention with S=%d, head depth %d is outside the MFMA kernels (S <= 512, depth 32 / 64); "
                    "using the fp32-math row kernels (exact, several times slower).  This notice is printed once.\n", who, S, dh);
}

static int check_attn(const char *who, int ld_qkv, int ld_o, int B, int S, int H, int dh) {
    B4C_REQUIRE(B > 0 && S > 0 && H > 0 && dh > 0, "%s: bad shape", who);
    B4C_REQUIRE(dh == 16 || dh == 32 || dh == 64 || dh == 128, "%s: head depth %d not in {16,32,64,128}", who, dh);
    B4C_REQUIRE(ld_qkv >= 3 * H * dh && ld_qkv % 8 == 0 && ld_o >= H * dh && ld_o % 8 == 0, "%s: bad pitch", who);
    B4C_REQUIRE((int64_t)B * H <= 65535, "%s: B*H = %lld exceeds the grid.y limit 65535", who, (long long)B * H);
    return B4C_OK;
}

#define ATT_DISPATCH_DH(dh, KERNEL, T, ...)                      \
    switch (dh) {                                                \
        case 16: KERNEL<T, 16> __VA_ARGS__; break;               \
        case 32: KERNEL<T, 32> __VA_ARGS__; break;               \
        case 64: KERNEL<T, 64> __VA_ARGS__; break;               \
        default: KERNEL<T, 128> __VA_ARGS__; break;              \
    }

static int g_attn_force_row = -1;
static bool force_row() {
    if (g_attn_force_row < 0) {
        const char *e = getenv("B4C_ATTN_ROW");
        g_attn_force_row = (e && e[0] == '1') ? 1 : 0;
    }
    return g_attn_force_row == 1;
}

extern "C" int b4c_attn_fwd(const void *qkv, int ld_qkv, const uint8_t *key_pad, void *o, int ld_o, float *lse, int B,
                            int S, int H, int dh, int dtype, void *stream) {
    B4C_REQUIRE(qkv && key_pad && o, "attn_fwd: null pointer");
    int rc = check_attn("attn_fwd", ld_qkv, ld_o, B, S, H, dh);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_BF16 && !force_row()) {
        rc = b4c_attn_fwd_mfma(qkv, ld_qkv, key_pad, o, ld_o, lse, B, S, H, dh, nullptr, st);
        if (rc != B4C_EUNSUPPORTED) return rc;
        note_row_fallback("attn_fwd", S, dh);
    }
    const float sq = sqrtf((float)dh);
    dim3 grid((S + 255) / 256, B * H);
    if (dtype == B4C_F32) {
        ATT_DISPATCH_DH(dh, attn_fwd_row_kernel, float, <<<grid, 256, 0, st>>>((const float *)qkv, ld_qkv, key_pad, (float *)o, ld_o, lse, S, H, sq))
    } else if (dtype == B4C_BF16) {
        ATT_DISPATCH_DH(dh, attn_fwd_row_kernel, bf16_t, <<<grid, 256, 0, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (bf16_t *)o, ld_o, lse, S, H, sq))
    } else
        B4C_REQUIRE(false, "attn_fwd: dtype %d", dtype);
    return b4c_check_launch("attn_fwd");
}

extern "C" int64_t b4c_attn_bwd_workspace_bytes(int B, int S, int H, int dh, int dtype) {
    return dtype == B4C_BF16 ? b4c_attn_bwd_mfma_workspace_bytes(B, S, H, dh) : 0;
}

extern "C" int b4c_attn_bwd(const void *qkv, int ld_qkv, const uint8_t *key_pad, const void *o, int ld_o,
                            const void *d_o, int ld_do, const float *lse, float *delta, void *dqkv, int ld_dqkv, int B,
                            int S, int H, int dh, int dtype, void *stream) {
    return b4c_attn_bwd_ws(qkv, ld_qkv, key_pad, o, ld_o, d_o, ld_do, lse, delta, dqkv, ld_dqkv, B, S, H, dh, nullptr, 0, dtype, stream);
}

extern "C" int b4c_attn_bwd_ws(const void *qkv, int ld_qkv, const uint8_t *key_pad, const void *o, int ld_o,
                               const void *d_o, int ld_do, const float *lse, float *delta, void *dqkv, int ld_dqkv, int B,
                               int S, int H, int dh, void *workspace, int64_t workspace_bytes, int dtype, void *stream) {
    B4C_REQUIRE(qkv && key_pad && o && d_o && lse && delta && dqkv, "attn_bwd: null pointer");
    int rc = check_attn("attn_bwd", ld_qkv, ld_o, B, S, H, dh);
    if (rc) return rc;
    B4C_REQUIRE(ld_do >= H * dh && ld_do % 8 == 0 && ld_dqkv >= 3 * H * dh && ld_dqkv % 8 == 0, "attn_bwd: bad pitch");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_BF16 && !force_row()) {
        rc = b4c_attn_bwd_mfma(qkv, ld_qkv, key_pad, o, ld_o, d_o, ld_do, lse, delta, dqkv, ld_dqkv, B, S, H, dh, workspace,
                               workspace_bytes, nullptr, st);
        if (rc != B4C_EUNSUPPORTED) return rc;
        note_row_fallback("attn_bwd", S, dh);
    }
    const float sq = sqrtf((float)dh);
    dim3 grid((S + 255) / 256, B * H);
    if (dtype == B4C_F32) {
        ATT_DISPATCH_DH(dh, attn_bwd_dq_kernel, float, <<<grid, 256, 0, st>>>((const float *)qkv, ld_qkv, key_pad, (const float *)o, ld_o, (const float *)d_o, ld_do, lse, delta, (float *)dqkv, ld_dqkv, S, H, sq))
        ATT_DISPATCH_DH(dh, attn_bwd_dkv_kernel, float, <<<grid, 256, 0, st>>>((const float *)qkv, ld_qkv, key_pad, (const float *)d_o, ld_do, lse, delta, (float *)dqkv, ld_dqkv, S, H, sq))
    } else if (dtype == B4C_BF16) {
        ATT_DISPATCH_DH(dh, attn_bwd_dq_kernel, bf16_t, <<<grid, 256, 0, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, lse, delta, (bf16_t *)dqkv, ld_dqkv, S, H, sq))
        ATT_DISPATCH_DH(dh, attn_bwd_dkv_kernel, bf16_t, <<<grid, 256, 0, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)d_o, ld_do, lse, delta, (bf16_t *)dqkv, ld_dqkv, S, H, sq))
    } else
        B4C_REQUIRE(false, "attn_bwd: dtype %d", dtype);
    return b4c_check_launch("attn_bwd");
}


// ---- packed (padding-free) layout: sequence b owns rows cu_seqlens[b] .. cu_seqlens[b+1] of qkv / o / d_o / dqkv; max_len is
// the longest sequence (LDS sizing, lse / delta pitch: [B][H][max_len]).  bf16, head depth 32 / 64, max_len <= 512 only: the
// packed layout exists for the throughput path, the fp32 parity path stays dense.  key_pad [total tokens]: all zeros unless
// the caller keeps masked keys inside the packed rows.
extern "C" int b4c_attn_fwd_varlen(const void *qkv, int ld_qkv, const uint8_t *key_pad, const int32_t *cu_seqlens, void *o, int ld_o,
                                   float *lse, int B, int max_len, int H, int dh, int dtype, void *stream) {
    B4C_REQUIRE(qkv && key_pad && cu_seqlens && o, "attn_fwd_varlen: null pointer");
    int rc = check_attn("attn_fwd_varlen", ld_qkv, ld_o, B, max_len, H, dh);
    if (rc) return rc;
    B4C_REQUIRE(dtype == B4C_BF16, "attn_fwd_varlen: bf16 only");
    rc = b4c_attn_fwd_mfma(qkv, ld_qkv, key_pad, o, ld_o, lse, B, max_len, H, dh, cu_seqlens, (hipStream_t)stream);
    B4C_REQUIRE(rc != B4C_EUNSUPPORTED, "attn_fwd_varlen: max_len %d / head depth %d outside the MFMA kernels (<= 512, 32 or 64)", max_len, dh);
    return rc;
}

extern "C" int b4c_attn_bwd_varlen(const void *qkv, int ld_qkv, const uint8_t *key_pad, const int32_t *cu_seqlens, const void *o,
                                   int ld_o, const void *d_o, int ld_do, const float *lse, float *delta, void *dqkv, int ld_dqkv,
                                   int B, int max_len, int H, int dh, void *workspace, int64_t workspace_bytes, int dtype,
                                   void *stream) {
    B4C_REQUIRE(qkv && key_pad && cu_seqlens && o && d_o && lse && delta && dqkv, "attn_bwd_varlen: null pointer");
    int rc = check_attn("attn_bwd_varlen", ld_qkv, ld_o, B, max_len, H, dh);
    if (rc) return rc;
    B4C_REQUIRE(dtype == B4C_BF16, "attn_bwd_varlen: bf16 only");
    B4C_REQUIRE(ld_do >= H * dh && ld_do % 8 == 0 && ld_dqkv >= 3 * H * dh && ld_dqkv % 8 == 0, "attn_bwd_varlen: bad pitch");
    rc = b4c_attn_bwd_mfma(qkv, ld_qkv, key_pad, o, ld_o, d_o, ld_do, lse, delta, dqkv, ld_dqkv, B, max_len, H, dh, workspace,
                           workspace_bytes, cu_seqlens, (hipStream_t)stream);
    B4C_REQUIRE(rc != B4C_EUNSUPPORTED, "attn_bwd_varlen: shape outside the MFMA kernels, or workspace missing for max_len > 256");
    return rc;
}
