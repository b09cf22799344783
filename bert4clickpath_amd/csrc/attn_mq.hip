// Attention for a FEW query rows per sequence against all of its keys: the last encoder layer of the Cloze path.
//
// The reference runs every encoder layer on every position (transformer.py:262-270) and then keeps the rows at the [MASK]
// positions only (clickstream_transformer.py:281-295): in the LAST layer the outputs of all other positions are never
// read.  There the queries are the ~10 masked rows of a sequence, the keys / values all of its tokens:
//     o_m = softmax(q_m K^T / sqrt(dk) + pad * -1e9) V        (transformer.py:64-97), m = the sequence's masked rows.
// With so few queries per (sequence, head) the arithmetic is tiny (2 * 10 * S * dh MACs) and the cost is reading K and V
// once: plain fp32 VALU math (exact parity arithmetic, both dtypes), no MFMA.
//
// Layouts: q / o / dq [R][ld] with head h in columns h*DH ..; kv / dkv [T][ld] with k in columns h*DH.. and v in
// H*DH + h*DH..; sequence b owns token rows cu[b] .. cu[b+1] and query rows moff[b] .. moff[b+1]; lse [R][H].
//
// One 256-thread workgroup per (sequence, head).  Queries go in chunks of MQ, keys in blocks of 128:
//   phase 1  two threads per key (lanes l and l + 32: one half of the head depth each, their partial dots meet by a
//            shuffle): s[m][key] (forward) or P, dP, dS and the key's dK / dV sums (backward) -- the key's K / V half rows
//            live in the thread's registers, q_m (and dO_m) are broadcast reads from LDS;
//   phase 2  lane per feature, queries dealt to the waves: o_m = sum_key p V[key] / dq_m = sum_key dS K[key], V / K rows
//            read from LDS (conflict-free: a wave reads one 128-B row).
#include <math.h>

#include "common.h"

#define MQ 16            // query rows per chunk
#define MQ_THREADS 256
#define MQ_KB 128        // keys per block: TWO threads per key (lanes l and l + 32 of a wave own the two halves of its row)

template <typename T, int N>
__device__ __forceinline__ void mq_load(const T *__restrict__ p, float (&v)[N]) {
#pragma unroll
    for (int c = 0; c < N; c += 8) {
        float t[8];
        Vec8<T>::load(p + c, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[c + k] = t[k];
    }
}
template <typename T, int N>
__device__ __forceinline__ void mq_store(T *__restrict__ p, const float (&v)[N]) {
#pragma unroll
    for (int c = 0; c < N; c += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[c + k];
        Vec8<T>::store(p + c, t);
    }
}
template <int N> __device__ __forceinline__ float mq_dot(const float (&a)[N], const float *__restrict__ row) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < N; d += 4) {
        const f32x4 q4 = *reinterpret_cast<const f32x4 *>(row + d);      // the 32 lanes of a half read one address: broadcast
        s += a[d] * q4[0];
        s += a[d + 1] * q4[1];
        s += a[d + 2] * q4[2];
        s += a[d + 3] * q4[3];
    }
    return s;
}

// LDS (dynamic): sQ [MQ][DH] f32 | sP [MQ][SP] f32 | sV [MQ_KB][DH] T          (SP = padded longest sequence)
template <typename T, int DH>
__global__ void __launch_bounds__(MQ_THREADS) attn_mq_fwd_kernel(const T *__restrict__ q, int ld_q, const T *__restrict__ kv, int ld_kv,
                                                                 const uint8_t *__restrict__ key_pad, const int32_t *__restrict__ cu,
                                                                 const int32_t *__restrict__ moff, T *__restrict__ o, int ld_o,
                                                                 float *__restrict__ lse, int H, int SP, float sqrt_dk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HD = DH / 2;                                // features per thread of a key pair
    float *sQ = reinterpret_cast<float *>(smem);
    float *sP = sQ + MQ * DH;
    T *sV = reinterpret_cast<T *>(sP + (size_t)MQ * SP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, kl = wave * 32 + (lane & 31);  // this thread: half `half` of key kl of the block
    const int b = blockIdx.x / H, h = blockIdx.x % H, dm = H * DH;
    const int64_t tok0 = cu[b];
    const int S = cu[b + 1] - cu[b];
    const int r0 = moff[b], M = moff[b + 1] - moff[b];
    if (M <= 0) return;
    constexpr int VCH = MQ_KB * (DH / 8) / MQ_THREADS;           // 16-B chunks of a V block per thread
    for (int mc = 0; mc < M; mc += MQ) {
        const int mq = min(MQ, M - mc);
        // every global load of the first key block goes out before the first wait: the K half row of this thread's key, the
        // V block (raw, on its way to LDS), the chunk's query rows -- one memory round trip per item instead of three
        float kr0[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) kr0[d] = 0.f;
        if (kl < S) mq_load<T, HD>(kv + (tok0 + kl) * ld_kv + h * DH + half * HD, kr0);
        float vraw[VCH][8];
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = tid + i * MQ_THREADS, row = c / (DH / 8), part = c % (DH / 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) vraw[i][k] = 0.f;
            if (row < S) Vec8<T>::load(kv + (tok0 + row) * ld_kv + dm + h * DH + part * 8, vraw[i]);
        }
        __syncthreads();                                      // the previous chunk's readers are done with sQ / sP / sV
        for (int c = tid; c < MQ * (DH / 8); c += MQ_THREADS) {       // rows past the chunk are zeros: the loops below run on
            const int m = c / (DH / 8), part = c % (DH / 8);          // groups of four queries without a bound check
            float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (m < mq) Vec8<T>::load(q + (int64_t)(r0 + mc + m) * ld_q + h * DH + part * 8, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) sQ[m * DH + part * 8 + k] = t[k];
        }
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int c = tid + i * MQ_THREADS, row = c / (DH / 8), part = c % (DH / 8);
            Vec8<T>::store(sV + row * DH + part * 8, vraw[i]);
        }
        __syncthreads();
        // phase 1: scores of every key against the chunk's queries (two threads per key, their half dots meet by a shuffle)
        const float rs = 1.0f / sqrt_dk;
        for (int k0 = 0; k0 < S; k0 += MQ_KB) {
            const int j = k0 + kl;
            float kr[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) kr[d] = kr0[d];
            if (k0 > 0) {
#pragma unroll
                for (int d = 0; d < HD; ++d) kr[d] = 0.f;
                if (j < S) mq_load<T, HD>(kv + (tok0 + j) * ld_kv + h * DH + half * HD, kr);
            }
            const bool pad = j < S && key_pad && key_pad[tok0 + j];
            for (int m0 = 0; m0 < mq; m0 += 4) {              // four independent dot -> shuffle chains in flight
                float s4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) s4[u] = mq_dot<HD>(kr, sQ + (m0 + u) * DH + half * HD);
#pragma unroll
                for (int u = 0; u < 4; ++u) s4[u] += __shfl_xor(s4[u], 32);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float sc = s4[u] / sqrt_dk;
                    if (pad) sc += -1e9f;
                    if (half == 0 && j < S) sP[(size_t)(m0 + u) * SP + j] = sc;
                }
            }
        }
        (void)rs;
        __syncthreads();
        // softmax over the keys: query m of the chunk belongs to wave m % 4
        for (int m = wave; m < mq; m += 4) {
            float *row = sP + (size_t)m * SP;
            float mx = -INFINITY;
            for (int j = lane; j < S; j += 64) mx = fmaxf(mx, row[j]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int j = lane; j < S; j += 64) sum += expf(row[j] - mx);
            sum = wave_sum(sum);
            const float L = mx + logf(sum);
            for (int j = lane; j < S; j += 64) row[j] = expf(row[j] - L);
            for (int j = S + lane; j < SP; j += 64) row[j] = 0.f;     // (the loop below runs over whole groups of four keys)
            if (lane == 0) lse[(int64_t)(r0 + mc + m) * H + h] = L;
        }
        // phase 2: o_m = sum_key p[m][key] V[key]: lane = feature, the wave's queries share every V read
        constexpr int QW = (MQ + 3) / 4;
        float acc[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) acc[i] = 0.f;
        for (int k0 = 0; k0 < S; k0 += MQ_KB) {
            if (k0 > 0) {
                __syncthreads();                              // the previous V block is consumed
                for (int c = tid; c < MQ_KB * (DH / 8); c += MQ_THREADS) {
                    const int row = c / (DH / 8), part = c % (DH / 8);
                    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (k0 + row < S) Vec8<T>::load(kv + (tok0 + k0 + row) * ld_kv + dm + h * DH + part * 8, t);
                    Vec8<T>::store(sV + row * DH + part * 8, t);          // rows past the sequence: zeros
                }
                __syncthreads();
            }
            const int nk = min(MQ_KB, S - k0);
            if (lane < DH) {
                // all reads of a step are requested before the first is used (the compiler waits once per group)
                const int nk4 = (nk + 3) & ~3;                // rows of sV and columns of sP past the sequence are zeros
                for (int j = 0; j < nk4; j += 4) {
                    f32x4 p4[QW];
#pragma unroll
                    for (int i = 0; i < QW; ++i) p4[i] = *reinterpret_cast<const f32x4 *>(sP + (size_t)(wave + 4 * i) * SP + k0 + j);
                    float v4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v4[u] = (float)sV[(j + u) * DH + lane];
#pragma unroll
                    for (int i = 0; i < QW; ++i)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[i] += p4[i][u] * v4[u];
                }
            }
        }
        if (lane < DH) {
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                const int m = wave + 4 * i;
                if (m < mq) o[(int64_t)(r0 + mc + m) * ld_o + h * DH + lane] = (T)acc[i];
            }
        }
    }
}

// LDS (dynamic): sQ [MQ][DH] | sG [MQ][DH] | sLse [MQ] | sDelta [MQ] | sDS [MQ][MQ_KB] (f32) | sK [MQ_KB][DH] T
template <typename T, int DH>
__global__ void __launch_bounds__(MQ_THREADS) attn_mq_bwd_kernel(const T *__restrict__ q, int ld_q, const T *__restrict__ kv, int ld_kv,
                                                                 const uint8_t *__restrict__ key_pad, const int32_t *__restrict__ cu,
                                                                 const int32_t *__restrict__ moff, const T *__restrict__ o, int ld_o,
                                                                 const T *__restrict__ d_o, int ld_do, const float *__restrict__ lse,
                                                                 T *__restrict__ dq, int ld_dq, T *__restrict__ dkv, int ld_dkv, int H,
                                                                 float sqrt_dk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HD = DH / 2;
    float *sQ = reinterpret_cast<float *>(smem);
    float *sG = sQ + MQ * DH;
    float *sLse = sG + MQ * DH;
    float *sDelta = sLse + MQ;
    float *sDS = sDelta + MQ;
    T *sK = reinterpret_cast<T *>(sDS + MQ * MQ_KB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, kl = wave * 32 + (lane & 31);
    const int b = blockIdx.x / H, h = blockIdx.x % H, dm = H * DH;
    const int64_t tok0 = cu[b];
    const int S = cu[b + 1] - cu[b];
    const int r0 = moff[b], M = moff[b + 1] - moff[b];
    if (M <= 0) {     // no query reads this sequence's keys in this layer: their gradient is zero
        for (int c = tid; c < S * (DH / 8); c += MQ_THREADS) {
            const int row = c / (DH / 8), part = c % (DH / 8);
            float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            Vec8<T>::store(dkv + (tok0 + row) * ld_dkv + h * DH + part * 8, z);
            Vec8<T>::store(dkv + (tok0 + row) * ld_dkv + dm + h * DH + part * 8, z);
        }
        return;
    }
    const int nchunk = (M + MQ - 1) / MQ;
    for (int k0 = 0; k0 < S; k0 += MQ_KB) {
        const int j = k0 + kl;
        const bool live = j < S;
        float kr[HD], vr[HD], dk[HD], dv[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { kr[d] = 0.f; vr[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
        if (live) {
            mq_load<T, HD>(kv + (tok0 + j) * ld_kv + h * DH + half * HD, kr);
            mq_load<T, HD>(kv + (tok0 + j) * ld_kv + dm + h * DH + half * HD, vr);
        }
        const bool pad = live && key_pad && key_pad[tok0 + j];
        __syncthreads();                                      // the previous key block's dq phase is done with sK
        mq_store<T, HD>(sK + kl * DH + half * HD, kr);          // (zeros for the slots past the sequence)
        for (int ci = 0; ci < nchunk; ++ci) {
            const int mc = ci * MQ, mq = min(MQ, M - mc);
            __syncthreads();                                  // the previous chunk's dq phase is done with sQ / sG / sDS
            for (int c = tid; c < MQ * (DH / 8); c += MQ_THREADS) {   // rows past the chunk: zeros (q, dO) -> dS = dV = dK terms 0
                const int m = c / (DH / 8), part = c % (DH / 8);
                float t[8], g8[8], o8[8];
                if (m >= mq) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { t[k] = 0.f; g8[k] = 0.f; o8[k] = 0.f; }
                } else {
                    Vec8<T>::load(q + (int64_t)(r0 + mc + m) * ld_q + h * DH + part * 8, t);
                    Vec8<T>::load(d_o + (int64_t)(r0 + mc + m) * ld_do + h * DH + part * 8, g8);
                    Vec8<T>::load(o + (int64_t)(r0 + mc + m) * ld_o + h * DH + part * 8, o8);
                }
                float pd = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    sQ[m * DH + part * 8 + k] = t[k];
                    sG[m * DH + part * 8 + k] = g8[k];
                    pd += g8[k] * o8[k];
                }
                pd = group_sum<DH / 8>(pd);                   // the DH / 8 consecutive lanes of a row
                if (part == 0) {
                    sDelta[m] = pd;
                    sLse[m] = m < mq ? lse[(int64_t)(r0 + mc + m) * H + h] : INFINITY;     // p = exp(s - inf) = 0 for the zero rows
                }
            }
            __syncthreads();
            // phase 1: this thread's half key against the chunk's queries
            for (int m = 0; m < mq; ++m) {
                float sc = mq_dot<HD>(kr, sQ + m * DH + half * HD);
                float dp = mq_dot<HD>(vr, sG + m * DH + half * HD);
                sc += __shfl_xor(sc, 32);
                dp += __shfl_xor(dp, 32);
                float p = 0.f, ds = 0.f;
                if (live && !pad) {                           // a padded key has p == 0 exactly (exp(-1e9 - lse))
                    p = expf(sc / sqrt_dk - sLse[m]);
                    ds = p * (dp - sDelta[m]);
                }
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4 *>(sG + m * DH + half * HD + d);
                    const f32x4 q4 = *reinterpret_cast<const f32x4 *>(sQ + m * DH + half * HD + d);
                    dv[d] += p * g4[0]; dv[d + 1] += p * g4[1]; dv[d + 2] += p * g4[2]; dv[d + 3] += p * g4[3];
                    dk[d] += ds * q4[0]; dk[d + 1] += ds * q4[1]; dk[d + 2] += ds * q4[2]; dk[d + 3] += ds * q4[3];
                }
                if (half == 0) sDS[m * MQ_KB + kl] = ds;
            }
            for (int m = mq + (tid >> 7); m < MQ; m += 2)     // rows past the chunk: zeros (phase 2 reads whole groups)
                sDS[m * MQ_KB + (tid & 127)] = 0.f;
            __syncthreads();
            // phase 2: dq_m (+)= sum over this block's keys of dS[m][key] K[key] / sqrt(dk): a read-modify-write of the
            // output row when the sequence has more than one key block (S > 128)
            const int nk = min(MQ_KB, S - k0);
            if (lane < DH) {
                constexpr int QW = (MQ + 3) / 4;
                float acc[QW];
#pragma unroll
                for (int i = 0; i < QW; ++i) acc[i] = 0.f;
                // every key slot of sDS is written each chunk (0 for slots past the sequence) and rows past the chunk are 0
                const int nk4 = (nk + 3) & ~3;
                for (int jj = 0; jj < nk4; jj += 4) {
                    f32x4 d4[QW];
#pragma unroll
                    for (int i = 0; i < QW; ++i) d4[i] = *reinterpret_cast<const f32x4 *>(sDS + (wave + 4 * i) * MQ_KB + jj);
                    float c4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) c4[u] = (float)sK[(jj + u) * DH + lane];
#pragma unroll
                    for (int i = 0; i < QW; ++i)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[i] += d4[i][u] * c4[u];
                }
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    const int m = wave + 4 * i;
                    if (m < mq) {
                        float a = acc[i] / sqrt_dk;
                        T *dst = dq + (int64_t)(r0 + mc + m) * ld_dq + h * DH + lane;
                        if (k0 > 0) a += (float)*dst;
                        *dst = (T)a;
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int d = 0; d < HD; ++d) dk[d] /= sqrt_dk;
            mq_store<T, HD>(dkv + (tok0 + j) * ld_dkv + h * DH + half * HD, dk);
            mq_store<T, HD>(dkv + (tok0 + j) * ld_dkv + dm + h * DH + half * HD, dv);
        }
    }
}

static size_t mq_fwd_lds(int SP, int dh, int esz) { return (size_t)MQ * dh * 4 + (size_t)MQ * SP * 4 + (size_t)MQ_KB * dh * esz; }
static size_t mq_bwd_lds(int dh, int esz) { return (size_t)2 * MQ * dh * 4 + 2 * MQ * 4 + (size_t)MQ * MQ_KB * 4 + (size_t)MQ_KB * dh * esz; }

template <typename Kern> static void mq_allow_lds(Kern k, size_t bytes) {
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static int mq_check(const char *who, const void *q, const void *kv, const int32_t *cu, const int32_t *moff, int ld_q, int ld_kv, int B,
                    int max_len, int H, int dh, int dtype) {
    B4C_REQUIRE(q && kv && cu && moff, "%s: null pointer", who);
    B4C_REQUIRE(B > 0 && H > 0 && max_len > 0, "%s: B=%d H=%d max_len=%d", who, B, H, max_len);
    B4C_REQUIRE(dh == 32 || dh == 64, "%s: head depth %d unsupported (32 or 64)", who, dh);
    B4C_REQUIRE(dtype == B4C_F32 || dtype == B4C_BF16, "%s: dtype %d", who, dtype);
    B4C_REQUIRE(ld_q >= H * dh && ld_kv >= 2 * H * dh && ld_q % 8 == 0 && ld_kv % 8 == 0, "%s: pitches (ld_q=%d ld_kv=%d)", who, ld_q, ld_kv);
    B4C_REQUIRE((((uintptr_t)q | (uintptr_t)kv) & 15) == 0, "%s: operands must be 16-byte aligned", who);
    return B4C_OK;
}

extern "C" int b4c_attn_mq_fwd(const void *q, int ld_q, const void *kv, int ld_kv, const uint8_t *key_pad, const int32_t *cu_seqlens,
                               const int32_t *q_offsets, void *o, int ld_o, float *lse, int B, int max_len, int H, int dh, int dtype,
                               void *stream) {
    const int rc = mq_check("attn_mq_fwd", q, kv, cu_seqlens, q_offsets, ld_q, ld_kv, B, max_len, H, dh, dtype);
    if (rc != B4C_OK) return rc;
    B4C_REQUIRE(o && lse && ld_o >= H * dh && ld_o % 8 == 0, "attn_mq_fwd: output");
    const int SP = (max_len + 63) / 64 * 64;
    const float sq = sqrtf((float)dh);
    hipStream_t st = (hipStream_t)stream;
    const size_t shm = mq_fwd_lds(SP, dh, dtype == B4C_BF16 ? 2 : 4);
    B4C_REQUIRE(shm <= 160 * 1024, "attn_mq_fwd: max_len %d needs %zu bytes of LDS", max_len, shm);
#define MQ_FWD(TT, DHH)                                                                                                              \
    do {                                                                                                                             \
        mq_allow_lds(attn_mq_fwd_kernel<TT, DHH>, shm);                                                                              \
        attn_mq_fwd_kernel<TT, DHH><<<B * H, MQ_THREADS, shm, st>>>((const TT *)q, ld_q, (const TT *)kv, ld_kv, key_pad, cu_seqlens, q_offsets, \
                                                             (TT *)o, ld_o, lse, H, SP, sq);                                        \
    } while (0)
    if (dtype == B4C_BF16) { if (dh == 64) MQ_FWD(bf16_t, 64); else MQ_FWD(bf16_t, 32); }
    else { if (dh == 64) MQ_FWD(float, 64); else MQ_FWD(float, 32); }
#undef MQ_FWD
    return b4c_check_launch("attn_mq_fwd");
}

extern "C" int b4c_attn_mq_bwd(const void *q, int ld_q, const void *kv, int ld_kv, const uint8_t *key_pad, const int32_t *cu_seqlens,
                               const int32_t *q_offsets, const void *o, int ld_o, const void *d_o, int ld_do, const float *lse,
                               void *dq, int ld_dq, void *dkv, int ld_dkv, int B, int max_len, int H, int dh, int dtype, void *stream) {
    const int rc = mq_check("attn_mq_bwd", q, kv, cu_seqlens, q_offsets, ld_q, ld_kv, B, max_len, H, dh, dtype);
    if (rc != B4C_OK) return rc;
    B4C_REQUIRE(o && d_o && lse && dq && dkv, "attn_mq_bwd: null pointer");
    B4C_REQUIRE(ld_o >= H * dh && ld_do >= H * dh && ld_dq >= H * dh && ld_dkv >= 2 * H * dh && ld_o % 8 == 0 && ld_do % 8 == 0 &&
                    ld_dq % 8 == 0 && ld_dkv % 8 == 0, "attn_mq_bwd: pitches");
    const float sq = sqrtf((float)dh);
    hipStream_t st = (hipStream_t)stream;
    const size_t shm = mq_bwd_lds(dh, dtype == B4C_BF16 ? 2 : 4);
#define MQ_BWD(TT, DHH)                                                                                                              \
    do {                                                                                                                             \
        mq_allow_lds(attn_mq_bwd_kernel<TT, DHH>, shm);                                                                              \
        attn_mq_bwd_kernel<TT, DHH><<<B * H, MQ_THREADS, shm, st>>>((const TT *)q, ld_q, (const TT *)kv, ld_kv, key_pad, cu_seqlens, q_offsets, \
                                                             (const TT *)o, ld_o, (const TT *)d_o, ld_do, lse, (TT *)dq, ld_dq,      \
                                                             (TT *)dkv, ld_dkv, H, sq);                                              \
    } while (0)
    if (dtype == B4C_BF16) { if (dh == 64) MQ_BWD(bf16_t, 64); else MQ_BWD(bf16_t, 32); }
    else { if (dh == 64) MQ_BWD(float, 64); else MQ_BWD(float, 32); }
#undef MQ_BWD
    return b4c_check_launch("attn_mq_bwd");
}
