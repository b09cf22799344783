// bf16 MFMA attention for gfx950 (throughput path).  One 256-thread workgroup per (sequence, head);
// the whole K / V of that head lives in LDS (S <= 256), scores stay in MFMA accumulators.
//
// forward : S^T = K Q^T with the KEY on the accumulator rows and the QUERY on the lane, so the softmax
//           row statistics are lane-local (one xor-32 exchange), and P^T feeds O^T = V^T P^T straight
//           from the accumulator registers (no LDS round trip).  V is staged transposed.
// backward: S = Q K^T and dP = dO V^T with the KEY on the lane; P and dS feed dV^T = dO^T P and
//           dK^T = Q^T dS from registers; only dS crosses LDS once, for dQ = dS K.
//           Waves own key tiles (dK^T / dV^T never leave registers), queries stream in 32-row tiles.
//
// v_mfma_f32_32x32x16_bf16 operand maps (lane l: r = l & 31, hf = l >> 5):
//   A[row r][k = 8 hf + j], B[k = 8 hf + j][col r], j = 0..7;  D reg t: row (t&3) + 8 (t>>2) + 4 hf, col r.
//   An accumulator tile X used as the B operand of the next MFMA (summing over X's rows): k-step s takes
//   regs 8s..8s+7, whose rows are 16 s + 8 (j>>2) + 4 hf + (j&3) -- the A operand must use the same k order.
#include <math.h>

#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define ATT_MAX_KT 8   // key tiles of 32 -> S <= 256

template <typename K> static void allow_lds_attn(K kernel, size_t bytes) {
    static thread_local const void *done[8];
    static thread_local size_t done_bytes[8];
    static thread_local int ndone = 0;
    for (int i = 0; i < ndone; ++i)
        if (done[i] == (const void *)kernel && done_bytes[i] >= bytes) return;
    (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (ndone < 8) { done[ndone] = (const void *)kernel; done_bytes[ndone++] = bytes; }
}

__device__ __forceinline__ bf16x8 pack8(const float *p) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)p[j];
    return v;
}
__device__ __forceinline__ bf16x8 frag_from_2x8B(const char *p0, const char *p1) {
    const u32x2 lo = *reinterpret_cast<const u32x2 *>(p0);
    const u32x2 hi = *reinterpret_cast<const u32x2 *>(p1);
    u32x4 w = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ int rowmap(int t, int hf) { return (t & 3) + 8 * (t >> 2) + 4 * hf; }

// Stage the transpose of a [rows][DH] bf16 block (global, row pitch ld) into LDS as [DH][rows] with row
// stride `str` bytes (multiple of 8).  Each participating lane owns one dword (2 columns) of 8 consecutive
// rows: 8 coalesced dword loads, 4 ds_write_b64.  `slot` = which 8-row group this lane-group handles.
template <int DH>
__device__ __forceinline__ void stage_transposed8(const bf16_t *__restrict__ src, int ld, int row0, int nrows_valid,
                                                  char *dst, int str, int grp, int dl) {
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = row0 + grp * 8 + j;
        v[j] = (row < nrows_valid) ? *reinterpret_cast<const unsigned *>(src + (int64_t)row * ld + 2 * dl) : 0u;
    }
    u32x2 lo0 = {(v[0] & 0xFFFFu) | (v[1] << 16), (v[2] & 0xFFFFu) | (v[3] << 16)};
    u32x2 lo1 = {(v[4] & 0xFFFFu) | (v[5] << 16), (v[6] & 0xFFFFu) | (v[7] << 16)};
    u32x2 hi0 = {(v[0] >> 16) | (v[1] & 0xFFFF0000u), (v[2] >> 16) | (v[3] & 0xFFFF0000u)};
    u32x2 hi1 = {(v[4] >> 16) | (v[5] & 0xFFFF0000u), (v[6] >> 16) | (v[7] & 0xFFFF0000u)};
    char *b = dst + (2 * dl) * str + (row0 + grp * 8) * 2;
    *reinterpret_cast<u32x2 *>(b) = lo0;
    *reinterpret_cast<u32x2 *>(b + 8) = lo1;
    *reinterpret_cast<u32x2 *>(b + str) = hi0;
    *reinterpret_cast<u32x2 *>(b + str + 8) = hi1;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ void __launch_bounds__(256) attn_fwd_mfma_kernel(const bf16_t *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                            bf16_t *__restrict__ o, int ld_o, float *__restrict__ lse, int S, int H,
                                                            float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSTR = DH * 2 + 16;
    constexpr int NKS = DH / 16, NDT = DH / 32, DWR = DH / 2;
    const int nkt = (S + 31) >> 5, S_pad = nkt * 32;
    const int VSTR = S_pad * 2 + 8;
    char *sK = smem;
    char *sVt = sK + S_pad * KSTR;
    float *sMask = reinterpret_cast<float *>(sVt + ((DH * VSTR + 15) & ~15));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int b = blockIdx.x / H, hh = blockIdx.x % H, dm = H * DH;
    const int64_t tok0 = (int64_t)b * S;
    const bf16_t *kbase = qkv + tok0 * ld + dm + hh * DH;
    const bf16_t *vbase = qkv + tok0 * ld + 2 * dm + hh * DH;

    for (int c = tid; c < S_pad * (DH / 8); c += 256) {
        const int row = c / (DH / 8), part = c % (DH / 8);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < S) v = *reinterpret_cast<const u32x4 *>(kbase + (int64_t)row * ld + part * 8);
        *reinterpret_cast<u32x4 *>(sK + row * KSTR + part * 16) = v;
    }
    {
        constexpr int G = 64 / DWR;
        const int dl = lane % DWR;
        for (int grp = wave * G + lane / DWR; grp * 8 < S_pad; grp += 4 * G)
            stage_transposed8<DH>(vbase, ld, 0, S, sVt, VSTR, grp, dl);
    }
    for (int k = tid; k < S_pad; k += 256) sMask[k] = (k >= S) ? -INFINITY : (key_pad[tok0 + k] ? -1e9f : 0.f);
    __syncthreads();

    for (int qt = wave; qt < nkt; qt += 4) {
        const int qrow = qt * 32 + r;
        const bool qvalid = qrow < S;
        bf16x8 qf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (qvalid) v = *reinterpret_cast<const u32x4 *>(qkv + (tok0 + qrow) * ld + hh * DH + ks * 16 + hf * 8);
            qf[ks] = __builtin_bit_cast(bf16x8, v);
        }
        f32x16 acc[ATT_MAX_KT];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < ATT_MAX_KT; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[kt][t] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(sK + (kt * 32 + r) * KSTR + ks * 32 + hf * 16);
                    acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc[kt], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float s = acc[kt][t] * scale + sMask[kt * 32 + rowmap(t, hf)];
                    acc[kt][t] = s;
                    m = fmaxf(m, s);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < ATT_MAX_KT; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float p = __expf(acc[kt][t] - m);
                    acc[kt][t] = p;
                    l += p;
                }
            }
        }
        l += __shfl_xor(l, 32);
        f32x16 oacc[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int t = 0; t < 16; ++t) oacc[dt][t] = 0.f;
#pragma unroll
        for (int kt = 0; kt < ATT_MAX_KT; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float pv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pv[j] = acc[kt][8 * s2 + j];
                    const bf16x8 pf = pack8(pv);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        const char *vb = sVt + (dt * 32 + r) * VSTR + (kt * 32 + 16 * s2 + 4 * hf) * 2;
                        const bf16x8 vf = frag_from_2x8B(vb, vb + 16);
                        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
                    }
                }
            }
        }
        if (qvalid) {
            const float inv = 1.0f / l;
            bf16_t *orow = o + (tok0 + qrow) * ld_o + hh * DH;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    bf16x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (bf16_t)(oacc[dt][4 * tq + j] * inv);
                    *reinterpret_cast<bf16x4 *>(orow + dt * 32 + 8 * tq + 4 * hf) = w;
                }
            if (hf == 0 && lse) lse[((int64_t)b * H + hh) * S + qrow] = m + logf(l);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ void __launch_bounds__(256) attn_delta_kernel(const bf16_t *__restrict__ o, int ld_o, const bf16_t *__restrict__ d_o, int ld_do,
                                                         float *__restrict__ delta, int B, int S, int H) {
    constexpr int LPR = DH / 8;  // lanes per (token, head)
    const int64_t gid = (blockIdx.x * 256ll + threadIdx.x) / LPR;
    const int part = threadIdx.x % LPR;
    const int64_t total = (int64_t)B * S * H;
    float s = 0.f;
    if (gid < total) {
        const int64_t tok = gid / H;
        const int hh = (int)(gid % H);
        float a[8], g[8];
        Vec8<bf16_t>::load(o + tok * ld_o + hh * DH + part * 8, a);
        Vec8<bf16_t>::load(d_o + tok * ld_do + hh * DH + part * 8, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a[k] * g[k];
    }
    s = group_sum<LPR>(s);
    if (gid < total && part == 0) {
        const int64_t tok = gid / H;
        const int hh = (int)(gid % H);
        delta[((tok / S) * H + hh) * S + tok % S] = s;
    }
}

template <int DH>
__global__ void __launch_bounds__(256) attn_bwd_mfma_kernel(const bf16_t *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                            const bf16_t *__restrict__ d_o, int ld_do, const float *__restrict__ lse,
                                                            const float *__restrict__ delta, bf16_t *__restrict__ dqkv, int ld_dq,
                                                            int S, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSTR = DH * 2 + 16;   // row-major [row][DH] images
    constexpr int QSTR = 32 * 2 + 8;    // transposed 32-query tiles [DH][32]
    constexpr int NKS = DH / 16, NDT = DH / 32, DWR = DH / 2;
    const int nkt = (S + 31) >> 5, S_pad = nkt * 32;
    const int TSTR = S_pad * 2 + 16;    // [DH][S_pad] (K^T) and [32][S_pad] (dS) images, 16-B aligned rows
    char *sK = smem;
    char *sV = sK + S_pad * KSTR;
    char *sKt = sV + S_pad * KSTR;
    char *sQ = sKt + DH * TSTR;
    char *sdO = sQ + 32 * KSTR;
    char *sQt = sdO + 32 * KSTR;
    char *sdOt = sQt + ((DH * QSTR + 15) & ~15);
    char *sDS = sdOt + ((DH * QSTR + 15) & ~15);
    float *sLse = reinterpret_cast<float *>(sDS + 32 * TSTR);
    float *sDelta = sLse + 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int b = blockIdx.x / H, hh = blockIdx.x % H, dm = H * DH;
    const int64_t tok0 = (int64_t)b * S;
    const bf16_t *qbase = qkv + tok0 * ld + hh * DH;
    const bf16_t *kbase = qbase + dm;
    const bf16_t *vbase = qbase + 2 * dm;
    const bf16_t *gbase = d_o + tok0 * ld_do + hh * DH;

    for (int c = tid; c < S_pad * (DH / 8); c += 256) {
        const int row = c / (DH / 8), part = c % (DH / 8);
        u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
        if (row < S) {
            kv = *reinterpret_cast<const u32x4 *>(kbase + (int64_t)row * ld + part * 8);
            vv = *reinterpret_cast<const u32x4 *>(vbase + (int64_t)row * ld + part * 8);
        }
        *reinterpret_cast<u32x4 *>(sK + row * KSTR + part * 16) = kv;
        *reinterpret_cast<u32x4 *>(sV + row * KSTR + part * 16) = vv;
    }
    constexpr int G = 64 / DWR;
    const int dl = lane % DWR, slot = wave * G + lane / DWR;
    for (int grp = slot; grp * 8 < S_pad; grp += 4 * G) stage_transposed8<DH>(kbase, ld, 0, S, sKt, TSTR, grp, dl);

    // this wave's key tiles: kt = wave and wave + 4; key of this lane inside a tile = r
    float madd[2];
    f32x16 dk[2][NDT], dv[2][NDT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = (wave + 4 * i) * 32 + r;
        madd[i] = (key >= S) ? -INFINITY : (key_pad[tok0 + key] ? -1e9f : 0.f);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int t = 0; t < 16; ++t) { dk[i][dt][t] = 0.f; dv[i][dt][t] = 0.f; }
    }

    for (int qt = 0; qt < nkt; ++qt) {
        const int q0 = qt * 32;
        __syncthreads();   // previous tile's dQ phase has finished with sDS / sQ / sdO
        for (int c = tid; c < 32 * (DH / 8); c += 256) {
            const int row = c / (DH / 8), part = c % (DH / 8);
            u32x4 qv = {0u, 0u, 0u, 0u}, gv = {0u, 0u, 0u, 0u};
            if (q0 + row < S) {
                qv = *reinterpret_cast<const u32x4 *>(qbase + (int64_t)(q0 + row) * ld + part * 8);
                gv = *reinterpret_cast<const u32x4 *>(gbase + (int64_t)(q0 + row) * ld_do + part * 8);
            }
            *reinterpret_cast<u32x4 *>(sQ + row * KSTR + part * 16) = qv;
            *reinterpret_cast<u32x4 *>(sdO + row * KSTR + part * 16) = gv;
        }
        if (slot < 4) stage_transposed8<DH>(qbase + (int64_t)q0 * ld, ld, 0, S - q0, sQt, QSTR, slot, dl);
        else if (slot < 8) stage_transposed8<DH>(gbase + (int64_t)q0 * ld_do, ld_do, 0, S - q0, sdOt, QSTR, slot - 4, dl);
        if (tid < 32) {
            const bool ok = q0 + tid < S;
            sLse[tid] = ok ? lse[((int64_t)b * H + hh) * S + q0 + tid] : INFINITY;
            sDelta[tid] = ok ? delta[((int64_t)b * H + hh) * S + q0 + tid] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kt = wave + 4 * i;
            if (kt < nkt) {
                f32x16 sa, pa;
#pragma unroll
                for (int t = 0; t < 16; ++t) { sa[t] = 0.f; pa[t] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const bf16x8 fq = *reinterpret_cast<const bf16x8 *>(sQ + r * KSTR + ks * 32 + hf * 16);
                    const bf16x8 fk = *reinterpret_cast<const bf16x8 *>(sK + (kt * 32 + r) * KSTR + ks * 32 + hf * 16);
                    sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq, fk, sa, 0, 0, 0);
                    const bf16x8 fg = *reinterpret_cast<const bf16x8 *>(sdO + r * KSTR + ks * 32 + hf * 16);
                    const bf16x8 fv = *reinterpret_cast<const bf16x8 *>(sV + (kt * 32 + r) * KSTR + ks * 32 + hf * 16);
                    pa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg, fv, pa, 0, 0, 0);
                }
                float pv[16], dsv[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int q = rowmap(t, hf);
                    const float p = __expf(sa[t] * scale + madd[i] - sLse[q]);
                    pv[t] = p;
                    dsv[t] = p * (pa[t] - sDelta[q]);
                    *reinterpret_cast<bf16_t *>(sDS + q * TSTR + (kt * 32 + r) * 2) = (bf16_t)dsv[t];
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 pf = pack8(pv + 8 * s2);
                    const bf16x8 df = pack8(dsv + 8 * s2);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        const int off = (dt * 32 + r) * QSTR + (16 * s2 + 4 * hf) * 2;
                        const bf16x8 fgt = frag_from_2x8B(sdOt + off, sdOt + off + 16);
                        dv[i][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fgt, pf, dv[i][dt], 0, 0, 0);
                        const bf16x8 fqt = frag_from_2x8B(sQt + off, sQt + off + 16);
                        dk[i][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fqt, df, dk[i][dt], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        if (wave < NDT) {   // dQ tile: [32 q][32 dh] per wave, summed over all keys
            f32x16 qa;
#pragma unroll
            for (int t = 0; t < 16; ++t) qa[t] = 0.f;
            for (int ks = 0; ks < S_pad / 16; ++ks) {
                const bf16x8 fs = *reinterpret_cast<const bf16x8 *>(sDS + r * TSTR + ks * 32 + hf * 16);
                const bf16x8 fk = *reinterpret_cast<const bf16x8 *>(sKt + (wave * 32 + r) * TSTR + ks * 32 + hf * 16);
                qa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs, fk, qa, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int q = q0 + rowmap(t, hf);
                if (q < S) dqkv[(tok0 + q) * ld_dq + hh * DH + wave * 32 + r] = (bf16_t)(qa[t] * scale);
            }
        }
    }
    // dK^T / dV^T accumulators: rows = dh (registers), col = key (lane)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = (wave + 4 * i) * 32 + r;
        if (wave + 4 * i < nkt && key < S) {
            bf16_t *krow = dqkv + (tok0 + key) * ld_dq + dm + hh * DH;
            bf16_t *vrow = krow + dm;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    bf16x4 wk, wv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        wk[j] = (bf16_t)(dk[i][dt][4 * tq + j] * scale);
                        wv[j] = (bf16_t)dv[i][dt][4 * tq + j];
                    }
                    *reinterpret_cast<bf16x4 *>(krow + dt * 32 + 8 * tq + 4 * hf) = wk;
                    *reinterpret_cast<bf16x4 *>(vrow + dt * 32 + 8 * tq + 4 * hf) = wv;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
static bool mfma_shape_ok(int S, int dh) { return (dh == 32 || dh == 64) && S <= 32 * ATT_MAX_KT; }

int b4c_attn_fwd_mfma(const void *qkv, int ld_qkv, const uint8_t *key_pad, void *o, int ld_o, float *lse, int B, int S,
                      int H, int dh, hipStream_t st) {
    if (!mfma_shape_ok(S, dh)) return B4C_EUNSUPPORTED;
    const int S_pad = (S + 31) / 32 * 32;
    const size_t shm = (size_t)S_pad * (dh * 2 + 16) + (((size_t)dh * (S_pad * 2 + 8) + 15) & ~(size_t)15) + (size_t)S_pad * 4;
    const float scale = 1.0f / sqrtf((float)dh);
    if (dh == 64) {
        allow_lds_attn(attn_fwd_mfma_kernel<64>, shm);
        attn_fwd_mfma_kernel<64><<<B * H, 256, shm, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (bf16_t *)o, ld_o, lse, S, H, scale);
    } else {
        allow_lds_attn(attn_fwd_mfma_kernel<32>, shm);
        attn_fwd_mfma_kernel<32><<<B * H, 256, shm, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (bf16_t *)o, ld_o, lse, S, H, scale);
    }
    return b4c_check_launch("attn_fwd_mfma");
}

int b4c_attn_bwd_mfma(const void *qkv, int ld_qkv, const uint8_t *key_pad, const void *o, int ld_o, const void *d_o,
                      int ld_do, const float *lse, float *delta, void *dqkv, int ld_dqkv, int B, int S, int H, int dh,
                      hipStream_t st) {
    if (!mfma_shape_ok(S, dh)) return B4C_EUNSUPPORTED;
    const int S_pad = (S + 31) / 32 * 32;
    const size_t kstr = dh * 2 + 16, tstr = S_pad * 2 + 16, qstr = 72;
    const size_t shm = 2 * S_pad * kstr + dh * tstr + 2 * 32 * kstr + 2 * ((dh * qstr + 15) & ~(size_t)15) + 32 * tstr + 64 * 4;
    const float scale = 1.0f / sqrtf((float)dh);
    const int64_t groups = (int64_t)B * S * H;
    if (dh == 64) {
        attn_delta_kernel<64><<<(int)ceil_div64(groups * 8, 256), 256, 0, st>>>((const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, delta, B, S, H);
        allow_lds_attn(attn_bwd_mfma_kernel<64>, shm);
        attn_bwd_mfma_kernel<64><<<B * H, 256, shm, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)d_o, ld_do, lse, delta, (bf16_t *)dqkv, ld_dqkv, S, H, scale);
    } else {
        attn_delta_kernel<32><<<(int)ceil_div64(groups * 4, 256), 256, 0, st>>>((const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, delta, B, S, H);
        allow_lds_attn(attn_bwd_mfma_kernel<32>, shm);
        attn_bwd_mfma_kernel<32><<<B * H, 256, shm, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)d_o, ld_do, lse, delta, (bf16_t *)dqkv, ld_dqkv, S, H, scale);
    }
    return b4c_check_launch("attn_bwd_mfma");
}
