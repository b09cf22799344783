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

// ------------------------------------------------------------------------------------------------------------------
// bf16 forward on the matrix cores, ONE WAVE per (sequence, head): the item's work (a 32-row query tile against S keys)
// is too small for a workgroup's barriers, and what it needs is many items in flight per CU (16 waves = 16 items).
//   S^T = K Q^T per 32-key tile, key on the accumulator rows, query on the lane (the maps of attn_mfma.hip): K and Q
//   fragments come straight from global memory (16 B per lane and k-step), the softmax statistics are lane-local (one
//   xor-32 exchange), P^T feeds O^T = V^T P^T from the accumulator registers; only V passes through LDS (a wave-private
//   double buffer, for the transposed fragment reads).  The next tile's K and V loads are in flight during a tile.
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned mq_u32x4;
typedef __attribute__((ext_vector_type(4))) short mq_s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 mq_bf16x4;
__device__ __forceinline__ int mq_rowmap(int t, int hf) { return (t & 3) + 8 * (t >> 2) + 4 * hf; }
__device__ __forceinline__ bf16x8 mq_pack8(const float *p) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)p[j];
    return v;
}
__device__ __forceinline__ bf16x8 mq_frag_tr(const char *p, int second_off) {
    const mq_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mq_s16x4 __attribute__((address_space(3))) *)(p));
    const mq_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mq_s16x4 __attribute__((address_space(3))) *)(p + second_off));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 w = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, w);
}

#define MQ_WAVES 4
template <int DH, bool HAS_PAD>
__global__ void __launch_bounds__(64 * MQ_WAVES, 3) attn_mq_fwd_mfma_kernel(const bf16_t *__restrict__ q, int ld_q, const bf16_t *__restrict__ kv,
                                                                         int ld_kv, const uint8_t *__restrict__ key_pad,
                                                                         const int32_t *__restrict__ cu, const int32_t *__restrict__ moff,
                                                                         bf16_t *__restrict__ o, int ld_o, float *__restrict__ lse, int H,
                                                                         int n_items, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSTR = DH * 2 + 16;
    constexpr int NKS = DH / 16, NDT = DH / 32, CH = DH / 8, VC = 32 * CH / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hf = lane >> 5, li = lane & 15, g = lane >> 4;
    char *sV = smem + wave * (2 * 32 * KSTR);                 // this wave's two V tiles
    const int item = blockIdx.x * MQ_WAVES + wave;
    if (item >= n_items) return;                              // (no workgroup barrier anywhere below)
    const int b = item / H, hh = item % H, dm = H * DH;
    const int64_t tok0 = cu[b];
    const int S = cu[b + 1] - cu[b];
    const int r0 = moff[b], M = moff[b + 1] - moff[b];
    if (M <= 0 || S <= 0) return;
    const int nkt = (S + 31) >> 5;
    const float scale2 = scale * 1.4426950408889634f;
    const bf16_t *kbase = kv + tok0 * ld_kv + hh * DH;
    const bf16_t *vbase = kbase + dm;
    for (int q0 = 0; q0 < M; q0 += 32) {
        const int qrow = q0 + r;
        const bool qvalid = qrow < M;
        bf16x8 qf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            mq_u32x4 v = {0u, 0u, 0u, 0u};
            if (qvalid) v = *reinterpret_cast<const mq_u32x4 *>(q + (int64_t)(r0 + qrow) * ld_q + hh * DH + ks * 16 + hf * 8);
            qf[ks] = __builtin_bit_cast(bf16x8, v);
        }
        // K / V rows through buffer loads: one descriptor for the sequence's rows (reads past its end return zeros), 32-bit
        // per-lane offsets -- no 64-bit address pair per load in the register file
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(kbase), 0, (unsigned)((int64_t)S * ld_kv * 2 - (int64_t)hh * DH * 2), 0x00020000);
        const int koff = (r * ld_kv + hf * 8) * 2;                        // this lane's K row, its half of a k-step
        const int voff0 = ((lane / CH) * ld_kv + dm + (lane % CH) * 8) * 2;     // V chunk i: rows (lane + 64 i) / CH
        const int tile_bytes = 32 * ld_kv * 2;
        mq_u32x4 kf[NKS], rv[VC];
        auto fetch_k = [&](int kt, mq_u32x4 (&fk)[NKS]) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) fk[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, koff + kt * tile_bytes + ks * 32, 0, 0);
        };
        auto fetch_v = [&](int kt, mq_u32x4 (&fv)[VC]) {
#pragma unroll
            for (int i = 0; i < VC; ++i) fv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff0 + kt * tile_bytes + i * (64 / CH) * ld_kv * 2, 0, 0);
        };
        fetch_k(0, kf);
        fetch_v(0, rv);
        float m = -INFINITY, l = 0.f;
        f32x16 oacc[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int t = 0; t < 16; ++t) oacc[dt][t] = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            char *vt = sV + (kt & 1) * (32 * KSTR);
#pragma unroll
            for (int i = 0; i < VC; ++i) {
                const int c = lane + 64 * i, row = c / CH, part = c % CH;
                *reinterpret_cast<mq_u32x4 *>(vt + row * KSTR + part * 16) = rv[i];
            }
            f32x16 acc;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[ks]), qf[ks], acc, 0, 0, 0);
            if (kt + 1 < nkt) {                                // in flight during this tile's softmax and P V
                fetch_k(kt + 1, kf);
                fetch_v(kt + 1, rv);
            }
            float tm = -INFINITY;
            const bool edge = (kt + 1) * 32 > S;              // keys past the sequence in this tile (wave-uniform)
            if (HAS_PAD || edge) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int key = kt * 32 + mq_rowmap(t, hf);
                    float madd = 0.f;
                    if (key >= S) madd = -INFINITY;
                    else if (HAS_PAD && key_pad[tok0 + key]) madd = -1e9f * 1.4426950408889634f;
                    acc[t] = __builtin_fmaf(acc[t], scale2, madd);
                    tm = fmaxf(tm, acc[t]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    acc[t] *= scale2;
                    tm = fmaxf(tm, acc[t]);
                }
            }
            tm = fmaxf(tm, __shfl_xor(tm, 32));               // the two lanes of a query (key halves) share the reference
            const bool raise = tm > m + 12.0f;
            if (__any(raise)) {
                if (raise) {
                    const float al = __builtin_amdgcn_exp2f(m - tm);      // 0 at the first tile (m = -inf)
                    m = tm;
                    l *= al;
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                        for (int t = 0; t < 16; ++t) oacc[dt][t] *= al;
                }
            }
            const float mref = (m == -INFINITY) ? 0.f : m;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc[t] = __builtin_amdgcn_exp2f(acc[t] - mref);
                l += acc[t];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this tile's V rows are in LDS (written by the wave's lanes)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = acc[8 * s2 + j];
                const bf16x8 pf = mq_pack8(pv);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    // V^T[dh = dt*32 + r][keys 16 s2 + 4 hf + {0..3, 8..11} of the tile] from the row-major V tile
                    const char *vb = vt + (16 * s2 + 4 * hf + (li >> 2)) * KSTR + (dt * 32 + 16 * (g & 1) + 4 * (li & 3)) * 2;
                    const bf16x8 vf = mq_frag_tr(vb, 8 * KSTR);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
                }
            }
        }
        l += __shfl_xor(l, 32);
        if (qvalid) {
            const float inv = 1.0f / l;
            bf16_t *orow = o + (int64_t)(r0 + qrow) * ld_o + hh * DH;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    mq_bf16x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (bf16_t)(oacc[dt][4 * tq + j] * inv);
                    *reinterpret_cast<mq_bf16x4 *>(orow + dt * 32 + 8 * tq + 4 * hf) = w;
                }
            if (hf == 0) lse[(int64_t)(r0 + qrow) * H + hh] = (m + __log2f(l)) * 0.6931471805599453f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 backward on the matrix cores, one wave per (sequence, head), key tiles of 32 (the maps of attn_mfma.hip's backward):
//   S = Q K^T and dP = dO V^T with the KEY on the lane (Q / dO rows and the K tile from LDS, V rows straight from global),
//   P = 2^(S scale log2e + mask - lse_q), dS = P (dP - delta_q); P and dS feed dV^T = dO^T P and dK^T = Q^T dS from the
//   accumulator registers; dS crosses LDS once for dQ^T += K^T dS^T (query on the lane: dQ stays in registers over the key
//   tiles).  dK / dV of a tile leave as 8-byte pieces of their rows.  More than 32 query rows: further passes that add to
//   the dK / dV rows already written.
// LDS per wave: sQ | sG [32][KSTR] (the query tile and its dO), sK [32][KSTR] (current key tile), sDS [32][TSTR], lse2 / delta.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 mq_frag_2x8B(const char *p0, const char *p1) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2v;
    const u32x2v lo = *reinterpret_cast<const u32x2v *>(p0);
    const u32x2v hi = *reinterpret_cast<const u32x2v *>(p1);
    const mq_u32x4 w = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, w);
}

template <int DH, bool HAS_PAD>
__global__ void __launch_bounds__(64 * MQ_WAVES, 2) attn_mq_bwd_mfma_kernel(const bf16_t *__restrict__ q, int ld_q, const bf16_t *__restrict__ kv,
                                                                         int ld_kv, const uint8_t *__restrict__ key_pad,
                                                                         const int32_t *__restrict__ cu, const int32_t *__restrict__ moff,
                                                                         const bf16_t *__restrict__ o, int ld_o, const bf16_t *__restrict__ d_o,
                                                                         int ld_do, const float *__restrict__ lse, bf16_t *__restrict__ dq,
                                                                         int ld_dq, bf16_t *__restrict__ dkv, int ld_dkv, int H, int n_items,
                                                                         float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSTR = DH * 2 + 16, TSTR = 32 * 2 + 16;
    constexpr int NKS = DH / 16, NDT = DH / 32, CH = DH / 8, VC = 32 * CH / 64;
    constexpr int WBYTES = 3 * 32 * KSTR + 32 * TSTR + 2 * 32 * 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hf = lane >> 5, li = lane & 15, g = lane >> 4;
    char *sQ = smem + wave * WBYTES;
    char *sG = sQ + 32 * KSTR;
    char *sK = sG + 32 * KSTR;
    char *sDS = sK + 32 * KSTR;
    float *sLse = reinterpret_cast<float *>(sDS + 32 * TSTR);
    float *sDelta = sLse + 32;
    const int item = blockIdx.x * MQ_WAVES + wave;
    if (item >= n_items) return;                              // (no workgroup barrier anywhere below)
    const int b = item / H, hh = item % H, dm = H * DH;
    const int64_t tok0 = cu[b];
    const int S = cu[b + 1] - cu[b];
    const int r0 = moff[b], M = moff[b + 1] - moff[b];
    if (S <= 0) return;
    bf16_t *dkbase = dkv + tok0 * ld_dkv + hh * DH;
    if (M <= 0) {                                             // no query reads this sequence's keys in this layer
        for (int c = lane; c < S * CH; c += 64) {
            const int row = c / CH, part = c % CH;
            const mq_u32x4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<mq_u32x4 *>(dkbase + (int64_t)row * ld_dkv + part * 8) = z;
            *reinterpret_cast<mq_u32x4 *>(dkbase + (int64_t)row * ld_dkv + dm + part * 8) = z;
        }
        return;
    }
    const int nkt = (S + 31) >> 5;
    const float scale2 = scale * 1.4426950408889634f;
    const bf16_t *kbase = kv + tok0 * ld_kv + hh * DH;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(kbase), 0, (unsigned)((int64_t)S * ld_kv * 2 - (int64_t)hh * DH * 2), 0x00020000);
    const int koff0 = ((lane / CH) * ld_kv + (lane % CH) * 8) * 2;            // K chunk i: rows (lane + 64 i) / CH
    const int voff = (r * ld_kv + dm + hf * 8) * 2;                            // this lane's V row (as the B operand of dP)
    const int tile_bytes = 32 * ld_kv * 2;
    for (int q0 = 0; q0 < M; q0 += 32) {
        const int qrow = q0 + r;
        const bool qvalid = qrow < M;
        // ---- the query tile: Q and dO rows -> LDS (row-major, zero rows past M), delta = rowsum(dO * O), lse in log2 units
        {
            float pd = 0.f;
#pragma unroll
            for (int i = 0; i < VC; ++i) {
                const int c = lane + 64 * i, row = c / CH, part = c % CH;
                mq_u32x4 vq = {0u, 0u, 0u, 0u}, vg = {0u, 0u, 0u, 0u}, vo = {0u, 0u, 0u, 0u};
                if (q0 + row < M) {
                    vq = *reinterpret_cast<const mq_u32x4 *>(q + (int64_t)(r0 + q0 + row) * ld_q + hh * DH + part * 8);
                    vg = *reinterpret_cast<const mq_u32x4 *>(d_o + (int64_t)(r0 + q0 + row) * ld_do + hh * DH + part * 8);
                    vo = *reinterpret_cast<const mq_u32x4 *>(o + (int64_t)(r0 + q0 + row) * ld_o + hh * DH + part * 8);
                }
                *reinterpret_cast<mq_u32x4 *>(sQ + row * KSTR + part * 16) = vq;
                *reinterpret_cast<mq_u32x4 *>(sG + row * KSTR + part * 16) = vg;
                const bf16x8 g8 = __builtin_bit_cast(bf16x8, vg), o8 = __builtin_bit_cast(bf16x8, vo);
                pd = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) pd += (float)g8[k] * (float)o8[k];
                pd = group_sum<CH>(pd);                       // CH consecutive lanes share a row
                if (part == 0) sDelta[row] = pd;
            }
            if (lane < 32) sLse[lane] = (q0 + lane < M) ? lse[(int64_t)(r0 + q0 + lane) * H + hh] * 1.4426950408889634f : INFINITY;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        bf16x8 qa[NKS], ga[NKS];                              // A operands: this lane's query row (half a k-step each)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qa[ks] = *reinterpret_cast<const bf16x8 *>(sQ + r * KSTR + ks * 32 + hf * 16);
            ga[ks] = *reinterpret_cast<const bf16x8 *>(sG + r * KSTR + ks * 32 + hf * 16);
        }
        float lq[16], dl[16];                                 // lse2 / delta of the accumulator rows (queries) of this lane
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const f32x4 a4 = *reinterpret_cast<const f32x4 *>(sLse + 8 * tq + 4 * hf);
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(sDelta + 8 * tq + 4 * hf);
#pragma unroll
            for (int j = 0; j < 4; ++j) { lq[4 * tq + j] = a4[j]; dl[4 * tq + j] = b4[j]; }
        }
        f32x16 dqa[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int t = 0; t < 16; ++t) dqa[dt][t] = 0.f;
        mq_u32x4 rk[VC], vb[NKS];
        auto fetch = [&](int kt) {
#pragma unroll
            for (int i = 0; i < VC; ++i) rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, koff0 + kt * tile_bytes + i * (64 / CH) * ld_kv * 2, 0, 0);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) vb[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + kt * tile_bytes + ks * 32, 0, 0);
        };
        fetch(0);
        for (int kt = 0; kt < nkt; ++kt) {
            const int key = kt * 32 + r;
#pragma unroll
            for (int i = 0; i < VC; ++i) {
                const int c = lane + 64 * i, row = c / CH, part = c % CH;
                *reinterpret_cast<mq_u32x4 *>(sK + row * KSTR + part * 16) = rk[i];
            }
            bf16x8 vcur[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) vcur[ks] = __builtin_bit_cast(bf16x8, vb[ks]);
            float madd = (key < S) ? 0.f : -INFINITY;
            if (HAS_PAD) { if (key < S && key_pad[tok0 + key]) madd = -1e9f * 1.4426950408889634f; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (kt + 1 < nkt) fetch(kt + 1);                  // in flight during this tile
            f32x16 sa, pa;
#pragma unroll
            for (int t = 0; t < 16; ++t) { sa[t] = 0.f; pa[t] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kb = *reinterpret_cast<const bf16x8 *>(sK + r * KSTR + ks * 32 + hf * 16);
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kb, sa, 0, 0, 0);
                pa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[ks], vcur[ks], pa, 0, 0, 0);
            }
            float pv[16], dsv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[t], scale2, madd - lq[t]));
                pv[t] = p;
                dsv[t] = p * (pa[t] - dl[t]);
                *reinterpret_cast<bf16_t *>(sDS + mq_rowmap(t, hf) * TSTR + r * 2) = (bf16_t)dsv[t];
            }
            // dV^T = dO^T P, dK^T = Q^T dS (key on the lane), from the accumulator registers
            f32x16 dv[NDT], dk[NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int t = 0; t < 16; ++t) { dv[dt][t] = 0.f; dk[dt][t] = 0.f; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = mq_pack8(pv + 8 * s2);
                const bf16x8 df = mq_pack8(dsv + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const int off = (16 * s2 + 4 * hf + (li >> 2)) * KSTR + (dt * 32 + 16 * (g & 1) + 4 * (li & 3)) * 2;
                    const bf16x8 fgt = mq_frag_tr(sG + off, 8 * KSTR);
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fgt, pf, dv[dt], 0, 0, 0);
                    const bf16x8 fqt = mq_frag_tr(sQ + off, 8 * KSTR);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fqt, df, dk[dt], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the dS tile is in LDS
            __builtin_amdgcn_wave_barrier();
            // dQ^T += K^T dS^T (query on the lane)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 fs = mq_frag_2x8B(sDS + r * TSTR + (16 * s2 + 4 * hf) * 2, sDS + r * TSTR + (16 * s2 + 8 + 4 * hf) * 2);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const int off = (16 * s2 + 4 * hf + (li >> 2)) * KSTR + (dt * 32 + 16 * (g & 1) + 4 * (li & 3)) * 2;
                    const bf16x8 fkt = mq_frag_tr(sK + off, 8 * KSTR);
                    dqa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fkt, fs, dqa[dt], 0, 0, 0);
                }
            }
            // this tile's dK / dV rows (key = lane): 8-byte pieces, added to what an earlier query pass wrote
            if (key < S) {
                bf16_t *krow = dkbase + (int64_t)key * ld_dkv;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int tq = 0; tq < 4; ++tq) {
                        mq_bf16x4 wk, wv;
                        bf16_t *pk = krow + dt * 32 + 8 * tq + 4 * hf;
                        float ak[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
                        if (q0 > 0) {
                            const mq_bf16x4 ek = *reinterpret_cast<const mq_bf16x4 *>(pk), ev = *reinterpret_cast<const mq_bf16x4 *>(pk + dm);
#pragma unroll
                            for (int j = 0; j < 4; ++j) { ak[j] = (float)ek[j]; av[j] = (float)ev[j]; }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            wk[j] = (bf16_t)(dk[dt][4 * tq + j] * scale + ak[j]);
                            wv[j] = (bf16_t)(dv[dt][4 * tq + j] + av[j]);
                        }
                        *reinterpret_cast<mq_bf16x4 *>(pk) = wk;
                        *reinterpret_cast<mq_bf16x4 *>(pk + dm) = wv;
                    }
            }
            __builtin_amdgcn_wave_barrier();                  // (sK / sDS are rewritten by the next tile: LDS is in order per wave)
        }
        if (qvalid) {
            bf16_t *qo = dq + (int64_t)(r0 + qrow) * ld_dq + hh * DH;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    mq_bf16x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (bf16_t)(dqa[dt][4 * tq + j] * scale);
                    *reinterpret_cast<mq_bf16x4 *>(qo + dt * 32 + 8 * tq + 4 * hf) = w;
                }
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
    static const bool valu_only = getenv("B4C_MQ_VALU") && atoi(getenv("B4C_MQ_VALU")) != 0;
    if (dtype == B4C_BF16 && !valu_only) {      // matrix-core form, one wave per (sequence, head)
        const int n_items = B * H;
        const size_t shm_m = (size_t)MQ_WAVES * 2 * 32 * (dh * 2 + 16);
        const int grid = (n_items + MQ_WAVES - 1) / MQ_WAVES;
#define MQ_MFMA(DHH, PP)                                                                                                             \
    attn_mq_fwd_mfma_kernel<DHH, PP><<<grid, 64 * MQ_WAVES, shm_m, st>>>((const bf16_t *)q, ld_q, (const bf16_t *)kv, ld_kv, key_pad, cu_seqlens, \
                                                                       q_offsets, (bf16_t *)o, ld_o, lse, H, n_items, 1.0f / sq)
        if (dh == 64) { if (key_pad) MQ_MFMA(64, true); else MQ_MFMA(64, false); }
        else { if (key_pad) MQ_MFMA(32, true); else MQ_MFMA(32, false); }
#undef MQ_MFMA
        return b4c_check_launch("attn_mq_fwd (mfma)");
    }
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
    static const bool valu_only = getenv("B4C_MQ_VALU") && atoi(getenv("B4C_MQ_VALU")) != 0;
    if (dtype == B4C_BF16 && !valu_only) {      // matrix-core form, one wave per (sequence, head)
        const int n_items = B * H, kstr = dh * 2 + 16;
        const size_t shm_m = (size_t)MQ_WAVES * (3 * 32 * kstr + 32 * (32 * 2 + 16) + 2 * 32 * 4);
        const int grid = (n_items + MQ_WAVES - 1) / MQ_WAVES;
#define MQ_MFMA_B(DHH, PP)                                                                                                           \
    do {                                                                                                                             \
        mq_allow_lds(attn_mq_bwd_mfma_kernel<DHH, PP>, shm_m);                                                                       \
        attn_mq_bwd_mfma_kernel<DHH, PP><<<grid, 64 * MQ_WAVES, shm_m, st>>>((const bf16_t *)q, ld_q, (const bf16_t *)kv, ld_kv, key_pad, cu_seqlens, \
            q_offsets, (const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, lse, (bf16_t *)dq, ld_dq, (bf16_t *)dkv, ld_dkv, H, n_items, 1.0f / sq); \
    } while (0)
        if (dh == 64) { if (key_pad) MQ_MFMA_B(64, true); else MQ_MFMA_B(64, false); }
        else { if (key_pad) MQ_MFMA_B(32, true); else MQ_MFMA_B(32, false); }
#undef MQ_MFMA_B
        return b4c_check_launch("attn_mq_bwd (mfma)");
    }
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
