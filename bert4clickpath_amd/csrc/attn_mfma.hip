// bf16 MFMA attention for gfx950 (throughput path).  One 512-thread workgroup per (sequence, head);
// the whole K / V of that head lives in LDS (S <= 256), scores stay in MFMA accumulators.
//
// forward : S^T = K Q^T with the KEY on the accumulator rows and the QUERY on the lane, so the softmax
//           row statistics are lane-local (one xor-32 exchange), and P^T feeds O^T = V^T P^T straight
//           from the accumulator registers (no LDS round trip).  V^T fragments come from the row-major V image
//           through ds_read_b64_tr_b16 (no transposed staging).
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

// -DB4C_ATTN_PHASES: per-workgroup timestamps (100 MHz) of the resident backward: start, images in LDS, tile loop done,
// stores issued -> g_attn_phases[blockIdx][4] once b4c_attn_phases_set() was given a buffer (scratch/attn_phases.py).
#ifdef B4C_ATTN_PHASES
__device__ unsigned long long *g_attn_phases = nullptr;
extern "C" void b4c_attn_phases_set(void *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_phases), &p, sizeof(p)); }
#define ATT_STAMP(i) do { if (tid == 0) att_t[i] = wall_clock64(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif

template <typename K> static void allow_lds_attn(K kernel, size_t bytes) {
    static thread_local const void *done[8];
    static thread_local size_t done_bytes[8];
    static thread_local int ndone = 0;
    for (int i = 0; i < ndone; ++i)
        if (done[i] == (const void *)kernel && done_bytes[i] >= bytes) return;
    (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (ndone < 8) { done[ndone] = (const void *)kernel; done_bytes[ndone++] = bytes; }
}

template <bool NT> __device__ __forceinline__ void att_store16(bf16_t *p, const u32x4 &v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
    else *reinterpret_cast<u32x4 *>(p) = v;
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

// 4 rows x 16 columns of bf16 read transposed (ds_read_b64_tr_b16): lane i of each 16-lane group gets
// column (col0 + i) of rows row0..row0+3; the lane supplies the address of row (i>>2), columns 4(i&3)...
// Checked on MI355X with integer data (scratch/trtest.hip).  EXEC must be full.
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ s16x4 tr_read(const char *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(p));
}
__device__ __forceinline__ bf16x8 frag_tr(const char *p, int second_off) {
    const s16x4 a = tr_read(p), b = tr_read(p + second_off);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 w = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, w);
}
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// ------------------------------------------------------------------------------------------
// forward: 512 threads, wave w owns query tiles w, w + 8, ... (QPW of them: S <= 256 QPW; QPW = 2 covers S <= 512, where
// the K / V images of one (sequence, head) take 147 KB of LDS and one workgroup runs per CU)
// ------------------------------------------------------------------------------------------
template <int DH, int QPW>
__global__ void __launch_bounds__(512, (QPW == 1 ? 4 : 2)) attn_fwd_mfma_kernel(const bf16_t *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                            bf16_t *__restrict__ o, int ld_o, float *__restrict__ lse, int S_arg, int H,
                                                            float scale, const int32_t *__restrict__ cu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSTR = DH * 2 + 16;
    constexpr int NKS = DH / 16, NDT = DH / 32, CH = DH / 8;
    const int b = blockIdx.x / H, hh = blockIdx.x % H, dm = H * DH;
    // packed (padding-free) layout: sequence b owns rows cu[b] .. cu[b+1] of qkv / o; S_arg is the longest sequence
    // (LDS is sized for it, lse rows keep the pitch S_arg).  cu == NULL: dense, every sequence S_arg rows.
    const int S = cu ? cu[b + 1] - cu[b] : S_arg;
    const int64_t tok0 = cu ? (int64_t)cu[b] : (int64_t)b * S_arg;
    const int nkt = (S + 31) >> 5, S_pad = nkt * 32;
    char *sK = smem;
    char *sV = sK + S_pad * KSTR;
    float *sMask = reinterpret_cast<float *>(sV + S_pad * KSTR);
    int *sLive = reinterpret_cast<int *>(sMask + S_pad);          // per key tile: any unmasked key?
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const float scale2 = scale * 1.4426950408889634f;
    const int li = lane & 15, g = lane >> 4;
    if (S <= 0) return;
    const bf16_t *kbase = qkv + tok0 * ld + dm + hh * DH;
    const bf16_t *vbase = kbase + dm;

    // every global load of the workgroup is issued up front: the key-padding byte (unconditional on a clamped index: a
    // conditional load behind the staging barrier was an exposed round trip), Q fragments, then K / V rows
    const uint8_t r_padk = key_pad[tok0 + (tid < S ? tid : S - 1)];       // S_pad <= 512 = threads (S > 0 here)
    bf16x8 qf[QPW][NKS];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int qrow = (wave + 8 * qi) * 32 + r;
        const bool qvalid = wave + 8 * qi < nkt && qrow < S;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (qvalid) v = *reinterpret_cast<const u32x4 *>(qkv + (tok0 + qrow) * ld + hh * DH + ks * 16 + hf * 8);
            qf[qi][ks] = __builtin_bit_cast(bf16x8, v);
        }
    }
    if (tid < ATT_MAX_KT * QPW) sLive[tid] = 0;
    {
        constexpr int NIT = (256 * QPW * CH + 511) / 512;
        u32x4 rk[NIT], rv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tid + it * 512;
            const int row = c / CH, part = c % CH;
            rk[it] = rv[it] = (u32x4){0u, 0u, 0u, 0u};
            if (c < S_pad * CH && row < S) {
                rk[it] = *reinterpret_cast<const u32x4 *>(kbase + (int64_t)row * ld + part * 8);
                rv[it] = *reinterpret_cast<const u32x4 *>(vbase + (int64_t)row * ld + part * 8);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tid + it * 512;
            const int row = c / CH, part = c % CH;
            if (c < S_pad * CH) {
                *reinterpret_cast<u32x4 *>(sK + row * KSTR + part * 16) = rk[it];
                *reinterpret_cast<u32x4 *>(sV + row * KSTR + part * 16) = rv[it];
            }
        }
    }
    __syncthreads();
    if (tid < S_pad) {
        const int k = tid;
        const bool live = k < S && !r_padk;
        sMask[k] = (k >= S) ? -INFINITY : (live ? 0.f : -1e9f * 1.4426950408889634f);
        if (live) sLive[k >> 5] = 1;     // benign race: every writer stores 1
    }
    __syncthreads();
    // Fully padded key tiles contribute exp(-1e9 - m) == 0 and are skipped -- unless the sequence has no
    // unmasked key at all (never the case for chained inputs: [CLS]/[SEP] are real tokens), where the
    // reference's softmax degenerates to uniform weights: then every tile is processed.
    int any_live = 0;
#pragma unroll
    for (int kt = 0; kt < ATT_MAX_KT * QPW; ++kt) any_live |= (kt < nkt) ? sLive[kt] : 0;
    const int force = !any_live;

    // Online softmax over the key tiles (one 32 x 32 score tile live at a time instead of all of them: ~100 registers,
    // two workgroups per CU).  Scores are in log2 units; p = 2^(s - m) against a per-query reference m that is raised
    // -- with l and O rescaled -- only when a score exceeds it by 2^12 (after the first live tile that is rare), so
    // the usual per-tile rescale of O disappears.  The two lanes of a query (key halves) share m: their P values
    // meet in the same PV MFMA.
    float m[QPW], l[QPW];
    f32x16 oacc[QPW][NDT];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        m[qi] = -INFINITY;
        l[qi] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int t = 0; t < 16; ++t) oacc[qi][dt][t] = 0.f;
    }
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const bool active = wave + 8 * qi < nkt;     // wave-uniform
        for (int kt = 0; kt < (active ? nkt : 0); ++kt) {
            if (!(sLive[kt] | force)) continue;       // wave-uniform
            f32x16 acc;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(sK + (kt * 32 + r) * KSTR + ks * 32 + hf * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qi][ks], acc, 0, 0, 0);
            }
            float tm = -INFINITY;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc[t] = __builtin_fmaf(acc[t], scale2, sMask[kt * 32 + rowmap(t, hf)]);
                tm = fmaxf(tm, acc[t]);
            }
            tm = fmaxf(tm, __shfl_xor(tm, 32));
            const bool raise = tm > m[qi] + 12.0f;
            if (__any(raise)) {
                if (raise) {
                    const float al = __builtin_amdgcn_exp2f(m[qi] - tm);      // 0 at the first live tile (m = -inf)
                    m[qi] = tm;
                    l[qi] *= al;
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                        for (int t = 0; t < 16; ++t) oacc[qi][dt][t] *= al;
                }
            }
            const float mref = (m[qi] == -INFINITY) ? 0.f : m[qi];                 // a tile of -inf scores only: p = 0
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                acc[t] = __builtin_amdgcn_exp2f(acc[t] - mref);
                l[qi] += acc[t];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = acc[8 * s2 + j];
                const bf16x8 pf = pack8(pv);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    // V^T[dh = dt*32 + r][keys kt*32 + 16 s2 + 4 hf + {0..3, 8..11}] from row-major V
                    const char *vb = sV + (kt * 32 + 16 * s2 + 4 * hf + (li >> 2)) * KSTR + (dt * 32 + 16 * (g & 1) + 4 * (li & 3)) * 2;
                    const bf16x8 vf = frag_tr(vb, 8 * KSTR);
                    oacc[qi][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[qi][dt], 0, 0, 0);
                }
            }
        }
        l[qi] += __shfl_xor(l[qi], 32);
    }
    // O^T (dh on the registers, query on the lane) -> row-major bf16 through this query tile's slice of the K image (dead
    // once every wave has left the key loops), then whole 16-B chunks: full rows per store instruction instead of 8 B per
    // lane scattered over 32 rows.
    B4C_LDS_BARRIER();
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
        const int qt = wave + 8 * qi;
        if (qt < nkt) {
            const float inv = 1.0f / l[qi];
            char *st = smem + qt * (32 * KSTR);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    bf16x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (bf16_t)(oacc[qi][dt][4 * tq + j] * inv);
                    *reinterpret_cast<bf16x4 *>(st + r * KSTR + (dt * 32 + 8 * tq + 4 * hf) * 2) = w;
                }
#pragma unroll
            for (int i = 0; i < 32 * CH / 64; ++i) {
                const int c = lane + 64 * i, row = c / CH, part = c % CH;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(st + row * KSTR + part * 16);
                if (qt * 32 + row < S) att_store16<B4C_NT(B4C_NT_ATTN_O)>(o + (tok0 + qt * 32 + row) * ld_o + hh * DH + part * 8, v);
            }
            const int qrow = qt * 32 + r;
            if (qrow < S && hf == 0 && lse) lse[((int64_t)b * H + hh) * S_arg + qrow] = (m[qi] + __log2f(l[qi])) * 0.6931471805599453f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward: 512 threads; wave w owns key tile w (dK^T / dV^T stay in its registers); 32-query tiles
// stream through a double-buffered LDS image with register prefetch; delta = rowsum(dO * O) is
// computed while staging; dQ tile = 8 (16 x 16) MFMA tiles, one per wave.
// ------------------------------------------------------------------------------------------
template <int DH>
__device__ __forceinline__ void attn_bwd_key_block(const bf16_t *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                   const bf16_t *__restrict__ o, int ld_o, const bf16_t *__restrict__ d_o,
                                                   int ld_do, const float *__restrict__ lse, bf16_t *__restrict__ dqkv,
                                                   int ld_dq, int S_arg, int H, float scale, int key0, float *__restrict__ dq_acc,
                                                   int acc_mode, const int32_t *__restrict__ cu) {
    // One block of up to 256 keys of one (sequence, head): the workgroup owns the dK / dV rows of those keys and the part of
    // dQ that sums over them.  acc_mode 0: dQ is complete, written as bf16; 1: first block of several, the partial dQ goes
    // to dq_acc (fp32 [B*S][H*DH]); 2: middle block, dq_acc += partial; 3: last block, dQ = bf16(dq_acc + partial).  Every
    // block streams all the query tiles.  (A dQ element is written and read back by the SAME thread in every block.)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSTR = DH * 2 + 16;
    constexpr int NKS = DH / 16, NDT = DH / 32, CH = DH / 8;
    constexpr int QBUF = 2 * 32 * KSTR + 256;          // sQ | sdO | lse[32] | delta[32]
    const int b = blockIdx.x / H, hh = blockIdx.x % H, dm = H * DH;
    const int S = cu ? cu[b + 1] - cu[b] : S_arg;                    // packed layout: this sequence's own length
    const int64_t tok0 = cu ? (int64_t)cu[b] : (int64_t)b * S_arg;
    const int nqt = (S + 31) >> 5;                                   // query tiles: the whole sequence
    const int nkt_all = (S - key0 + 31) >> 5;
    const int nkt = nkt_all > ATT_MAX_KT ? ATT_MAX_KT : (nkt_all < 0 ? 0 : nkt_all), S_pad = nkt * 32;   // key tiles of this launch
    if (nkt == 0) {
        // a sequence that ends before this key block (packed layout, mixed lengths): nothing to add; the last block
        // still has to turn the accumulated dQ into bf16
        if (acc_mode == 3) {
            for (int c = threadIdx.x; c < S * DH; c += 512) {
                const int q = c / DH, e = c % DH;
                dqkv[(tok0 + q) * ld_dq + hh * DH + e] = (bf16_t)dq_acc[(tok0 + q) * (int64_t)dm + hh * DH + e];
            }
        }
        return;
    }
    const int TSTR = S_pad * 2 + 16;
    char *sK = smem;
    char *sV = sK + S_pad * KSTR;
    char *sQB = sV + S_pad * KSTR;                       // 2 query-tile buffers
    char *sDS = sQB + 2 * QBUF;                          // [32][TSTR] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int li = lane & 15, g = lane >> 4;
    const bf16_t *qbase = qkv + tok0 * ld + hh * DH;
    const bf16_t *kbase = qbase + dm;
    const bf16_t *vbase = qbase + 2 * dm;
    const bf16_t *gbase = d_o + tok0 * ld_do + hh * DH;
    const bf16_t *obase = o + tok0 * ld_o + hh * DH;
    const float *lbase = lse + ((int64_t)b * H + hh) * S_arg;

    // staging roles for a 32-query tile: threads [0, 32 CH) carry Q chunks, [256, 256 + 32 CH) carry dO (+ O for delta)
    const bool roleQ = tid < 32 * CH, roleG = tid >= 256 && tid < 256 + 32 * CH;
    const int srow = (roleG ? tid - 256 : tid) / CH, spart = (roleG ? tid - 256 : tid) % CH;
    u32x4 pre = {0u, 0u, 0u, 0u};
    float pre_delta = 0.f, pre_lse = INFINITY;
    auto prefetch = [&](int q0) {
        pre = (u32x4){0u, 0u, 0u, 0u};
        pre_delta = 0.f;
        const bool ok = q0 + srow < S;
        if (roleQ && ok) pre = *reinterpret_cast<const u32x4 *>(qbase + (int64_t)(q0 + srow) * ld + spart * 8);
        if (roleG) {
            float part = 0.f;
            if (ok) {
                pre = *reinterpret_cast<const u32x4 *>(gbase + (int64_t)(q0 + srow) * ld_do + spart * 8);
                const bf16x8 gv = __builtin_bit_cast(bf16x8, pre);
                const bf16x8 ov = *reinterpret_cast<const bf16x8 *>(obase + (int64_t)(q0 + srow) * ld_o + spart * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) part += (float)gv[k] * (float)ov[k];
            }
            pre_delta = group_sum<CH>(part);
        }
        if (tid < 32) pre_lse = (q0 + tid < S) ? lbase[q0 + tid] : INFINITY;
    };
    auto commit = [&](char *buf) {
        if (roleQ) *reinterpret_cast<u32x4 *>(buf + srow * KSTR + spart * 16) = pre;
        if (roleG) {
            *reinterpret_cast<u32x4 *>(buf + 32 * KSTR + srow * KSTR + spart * 16) = pre;
            if (spart == 0) reinterpret_cast<float *>(buf + 2 * 32 * KSTR)[32 + srow] = pre_delta;
        }
        if (tid < 32) reinterpret_cast<float *>(buf + 2 * 32 * KSTR)[tid] = pre_lse;
    };

    prefetch(0);
    for (int c = tid; c < S_pad * CH; c += 512) {
        const int row = c / CH, part = c % CH;
        u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
        if (key0 + row < S) {
            kv = *reinterpret_cast<const u32x4 *>(kbase + (int64_t)(key0 + row) * ld + part * 8);
            vv = *reinterpret_cast<const u32x4 *>(vbase + (int64_t)(key0 + row) * ld + part * 8);
        }
        *reinterpret_cast<u32x4 *>(sK + row * KSTR + part * 16) = kv;
        *reinterpret_cast<u32x4 *>(sV + row * KSTR + part * 16) = vv;
    }
    commit(sQB);

    // this wave's key tile; the lane's key inside it is r
    const int kt = wave;
    const int key = key0 + kt * 32 + r;
    const bool key_live = kt < nkt && key < S && !key_pad[tok0 + key];
    const float madd = (kt >= nkt || key >= S) ? -INFINITY : (key_live ? 0.f : -1e9f);
    const bool tile_live = __any(key_live);           // ballot over the wave's 64 lanes (both halves hold the same keys)
    f32x16 dk[NDT], dv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int t = 0; t < 16; ++t) { dk[dt][t] = 0.f; dv[dt][t] = 0.f; }
    __syncthreads();

    for (int qt = 0; qt < nqt; ++qt) {
        const int q0 = qt * 32;
        char *cur = sQB + (qt & 1) * QBUF;
        const char *sQ = cur, *sdO = cur + 32 * KSTR;
        const float *sLse = reinterpret_cast<const float *>(cur + 2 * 32 * KSTR), *sDelta = sLse + 32;
        const bool more = qt + 1 < nqt;
        if (more) prefetch(q0 + 32);
        if (kt < nkt) {
            if (tile_live) {
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
                    const float p = __expf(sa[t] * scale + madd - sLse[q]);
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
                        // dO^T / Q^T [dh = dt*32 + r][queries 16 s2 + 4 hf + {0..3, 8..11}] from the row-major tiles
                        const int off = (16 * s2 + 4 * hf + (li >> 2)) * KSTR + (dt * 32 + 16 * (g & 1) + 4 * (li & 3)) * 2;
                        const bf16x8 fgt = frag_tr(sdO + off, 8 * KSTR);
                        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fgt, pf, dv[dt], 0, 0, 0);
                        const bf16x8 fqt = frag_tr(sQ + off, 8 * KSTR);
                        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fqt, df, dk[dt], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    *reinterpret_cast<bf16_t *>(sDS + rowmap(t, hf) * TSTR + (kt * 32 + r) * 2) = (bf16_t)0.f;
            }
        }
        __syncthreads();                                  // dS tile complete
        if (wave < 2 * (DH / 16)) {                       // dQ[32 q][DH] as (16 q) x (16 dh) tiles, one per wave
            const int qi = wave / (DH / 16), di = wave % (DH / 16);
            f32x4 qa = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < nkt; ++ks) {
                const bf16x8 fs = *reinterpret_cast<const bf16x8 *>(sDS + (qi * 16 + li) * TSTR + (ks * 32 + 8 * g) * 2);
                // K^T: B[k = key ks*32 + 8 g + j][col = dh di*16 + li] via two transposed 4-key reads
                const char *kb = sK + (ks * 32 + 8 * g + (li >> 2)) * KSTR + (di * 16 + 4 * (li & 3)) * 2;
                const bf16x8 fk = frag_tr(kb, 4 * KSTR);
                qa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fs, fk, qa, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int q = q0 + qi * 16 + 4 * g + t;
                if (q < S) {
                    float v = qa[t] * scale;
                    if (acc_mode != 0) {
                        float *ap = dq_acc + (tok0 + q) * (int64_t)dm + hh * DH + di * 16 + li;
                        if (acc_mode >= 2) v += *ap;
                        if (acc_mode <= 2) *ap = v;
                    }
                    if (acc_mode == 0 || acc_mode == 3) dqkv[(tok0 + q) * ld_dq + hh * DH + di * 16 + li] = (bf16_t)v;
                }
            }
        }
        if (more) commit(sQB + ((qt + 1) & 1) * QBUF);
        __syncthreads();                                  // dS consumed; next query tile visible
    }
    if (kt < nkt && key < S) {
        bf16_t *krow = dqkv + (tok0 + key) * ld_dq + dm + hh * DH;
        bf16_t *vrow = krow + dm;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                bf16x4 wk, wv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wk[j] = (bf16_t)(dk[dt][4 * tq + j] * scale);
                    wv[j] = (bf16_t)dv[dt][4 * tq + j];
                }
                *reinterpret_cast<bf16x4 *>(krow + dt * 32 + 8 * tq + 4 * hf) = wk;
                *reinterpret_cast<bf16x4 *>(vrow + dt * 32 + 8 * tq + 4 * hf) = wv;
            }
    }
}

// Sequences of up to 256 keys: one block (acc_mode 0).  Longer ones (S <= 512): the workgroup walks the blocks of 256 keys
// one after the other in ONE launch (round 2 launched the kernel once per block): Q, dO, O and the fp32 partial dQ of the
// second pass come out of L2 right behind the first instead of from HBM behind a whole launch.
template <int DH>
__global__ void __launch_bounds__(512) attn_bwd_mfma_kernel(const bf16_t *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                            const bf16_t *__restrict__ o, int ld_o, const bf16_t *__restrict__ d_o,
                                                            int ld_do, const float *__restrict__ lse, bf16_t *__restrict__ dqkv,
                                                            int ld_dq, int S_arg, int H, float scale, int nblk, float *__restrict__ dq_acc,
                                                            const int32_t *__restrict__ cu) {
    for (int kb = 0; kb < nblk; ++kb) {
        const int mode = nblk == 1 ? 0 : (kb == 0 ? 1 : (kb == nblk - 1 ? 3 : 2));
        attn_bwd_key_block<DH>(qkv, ld, key_pad, o, ld_o, d_o, ld_do, lse, dqkv, ld_dq, S_arg, H, scale, kb * 256, dq_acc, mode, cu);
    }
}

// ------------------------------------------------------------------------------------------
// backward, fully resident (S_pad <= 224 at dh = 64): K, V, Q, dO of the (sequence, head) are all staged
// once -- one burst of ~130 KB of loads per workgroup instead of a dependent load per query tile (the
// streaming kernel above spends ~60 % of its wave-cycles parked on those, SQ_WAIT_ANY) -- then the 32-query
// tiles run back to back out of LDS.
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ void __launch_bounds__(512) attn_bwd_resident_kernel(const bf16_t *__restrict__ qkv, int ld, const uint8_t *__restrict__ key_pad,
                                                                const bf16_t *__restrict__ o, int ld_o, const bf16_t *__restrict__ d_o,
                                                                int ld_do, const float *__restrict__ lse, bf16_t *__restrict__ dqkv,
                                                                int ld_dq, int S_arg, int H, float scale, int n_items, int *__restrict__ work_counter,
                                                                const int32_t *__restrict__ cu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSTR = DH * 2 + 16;
    constexpr int NKS = DH / 16, NDT = DH / 32, CH = DH / 8;
    // the LDS layout is that of the longest sequence S_arg; with the packed layout (cu != NULL) every item brings its own
    // length S <= S_arg and only walks its own tiles
    const int S_pad_max = ((S_arg + 31) >> 5) * 32;
    const int TSTR = S_pad_max * 2 + 16;
    char *sK = smem;
    char *sV = sK + S_pad_max * KSTR;
    char *sQa = sV + S_pad_max * KSTR;
    char *sGa = sQa + S_pad_max * KSTR;
    char *sDS = sGa + S_pad_max * KSTR;                       // [32][TSTR] bf16
    float *sLseA = reinterpret_cast<float *>(sDS + 32 * TSTR);
    float *sDeltaA = sLseA + S_pad_max;
    int *sNZ = reinterpret_cast<int *>(sDeltaA + S_pad_max);  // [ATT_MAX_KT][SPT]: does this slice of the query tile hold a nonzero dO?
    int *sNext = sNZ + ATT_MAX_KT * 4;                    // the item this workgroup takes next
    char *sDQ = reinterpret_cast<char *>(sNext + 4);      // [32][KSTR] bf16: the dQ tile on its way out as row chunks
    constexpr int RPW = 64 / CH, SPT = 32 / RPW;          // rows one wave stages per pass, such slices per query tile
    const int dm = H * DH;
    // One workgroup per CU walks the (sequence, head) items: no dispatch gap between items, and the stores of one item
    // drain under the loads of the next.  After its first item a workgroup takes items from a global counter (ragged
    // sequences and skipped query tiles make item times differ by several x; a static split leaves CUs idle at the end).
    // (Prefetching the next item's loads into registers across the loop was measured slower: 256 VGPRs, and the only
    // window is the 1 us epilogue.)
    for (int item = blockIdx.x; item < n_items;) {
    // the lane coordinates are re-derived per item behind an opaque zero: hoisted out of this loop, the per-lane LDS
    // addresses of every access pattern stay live across it and push the kernel into spills
    int opaque0;
    asm volatile("s_mov_b32 %0, 0" : "=s"(opaque0));
    const int tid = threadIdx.x + opaque0, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int li = lane & 15, g = lane >> 4;
    const int b = item / H, hh = item % H;
    const int S = cu ? cu[b + 1] - cu[b] : S_arg;
    const int64_t tok0 = cu ? (int64_t)cu[b] : (int64_t)b * S_arg;
    const int nkt = (S + 31) >> 5, S_pad = nkt * 32;
    const bf16_t *qbase = qkv + tok0 * ld + hh * DH;
    const bf16_t *kbase = qbase + dm;
    const bf16_t *vbase = qbase + 2 * dm;
    const bf16_t *gbase = d_o + tok0 * ld_do + hh * DH;
    const bf16_t *obase = o + tok0 * ld_o + hh * DH;
    const float *lbase = lse + ((int64_t)b * H + hh) * S_arg;

#ifdef B4C_ATTN_PHASES
    unsigned long long att_t[4] = {0, 0, 0, 0};
#endif
    ATT_STAMP(0);
    // the two small loads go out first, unconditional on clamped indices (a conditional scalar load is waited for on
    // the spot; behind the staging they were two exposed round trips per item)
    const float r_lse = lbase[tid < S ? tid : (S > 0 ? S - 1 : 0)];      // S_pad <= 256 < 512 threads
    const int key_c = wave * 32 + r;
    const uint8_t r_pad = key_pad[S > 0 ? tok0 + (key_c < S ? key_c : S - 1) : 0];      // (an empty sequence owns no row)
    {   // all global loads of the workgroup are issued before the first LDS write (one latency, not one per pass)
        constexpr int NIT = (256 * CH + 511) / 512;
        u32x4 rk[NIT], rv[NIT], rq[NIT], rg[NIT], ro[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tid + it * 512;
            const int row = c / CH, part = c % CH;
            rk[it] = rv[it] = rq[it] = rg[it] = ro[it] = (u32x4){0u, 0u, 0u, 0u};
            if (c < S_pad * CH && row < S) {
                rk[it] = *reinterpret_cast<const u32x4 *>(kbase + (int64_t)row * ld + part * 8);
                rv[it] = *reinterpret_cast<const u32x4 *>(vbase + (int64_t)row * ld + part * 8);
                rq[it] = *reinterpret_cast<const u32x4 *>(qbase + (int64_t)row * ld + part * 8);
                rg[it] = *reinterpret_cast<const u32x4 *>(gbase + (int64_t)row * ld_do + part * 8);
                ro[it] = *reinterpret_cast<const u32x4 *>(obase + (int64_t)row * ld_o + part * 8);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tid + it * 512;
            const int row = c / CH, part = c % CH;
            if (c < S_pad * CH) {          // wave-uniform: S_pad * CH is a multiple of 128
                *reinterpret_cast<u32x4 *>(sK + row * KSTR + part * 16) = rk[it];
                *reinterpret_cast<u32x4 *>(sV + row * KSTR + part * 16) = rv[it];
                *reinterpret_cast<u32x4 *>(sQa + row * KSTR + part * 16) = rq[it];
                *reinterpret_cast<u32x4 *>(sGa + row * KSTR + part * 16) = rg[it];
                const bf16x8 g8 = __builtin_bit_cast(bf16x8, rg[it]), o8 = __builtin_bit_cast(bf16x8, ro[it]);
                float pd = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) pd += (float)g8[k] * (float)o8[k];
                pd = group_sum<CH>(pd);     // CH consecutive lanes share a row
                if (part == 0) sDeltaA[row] = pd;
                // rows past S were loaded as zeros; +0 only (a -0 row takes the dense path, so the result bits match)
                const int nz = __any((rg[it][0] | rg[it][1] | rg[it][2] | rg[it][3]) != 0u);
                if (lane == 0) sNZ[(row >> 5) * SPT + (wave % SPT)] = nz;
            }
        }
    }
    if (tid < S_pad) sLseA[tid] = (tid < S) ? r_lse * 1.4426950408889634f : INFINITY;   // log2 units

    const int kt = wave;
    const int key = kt * 32 + r;
    const bool key_live = kt < nkt && key < S && !r_pad;
    const float madd = (kt >= nkt || key >= S) ? -INFINITY : (key_live ? 0.f : -1e9f * 1.4426950408889634f);
    const float scale2 = scale * 1.4426950408889634f;
    const bool tile_live = __any(key_live);
    f32x16 dk[NDT], dv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int t = 0; t < 16; ++t) { dk[dt][t] = 0.f; dv[dt][t] = 0.f; }
    if (kt < nkt && !tile_live) {     // a fully padded key tile never changes: its dS columns are zero for every query tile
#pragma unroll
        for (int t = 0; t < 16; ++t)
            *reinterpret_cast<bf16_t *>(sDS + rowmap(t, hf) * TSTR + (kt * 32 + r) * 2) = (bf16_t)0.f;
    }
    B4C_LDS_BARRIER();            // LDS only: the previous item's stores keep draining
    ATT_STAMP(1);
    // The next item is taken now and needed at the end of this one.  atomicAdd() reads its result back on the spot (the
    // compiler's atomic optimizer), a global round trip: the last wave does it -- with S <= 224 it owns no key tile and
    // only waits at the next barrier anyway -- and leaves the answer in LDS.
    if (tid == 448) *sNext = (int)gridDim.x + atomicAdd(work_counter, 1);
    // query tiles that hold a nonzero dO, as a wave-uniform bit mask (lane i < nkt * SPT reads one slice flag)
    const unsigned long long nzb = __ballot(lane < nkt * SPT && sNZ[lane < nkt * SPT ? lane : 0] != 0);
    unsigned q_live_mask = 0;
#pragma unroll
    for (int t = 0; t < ATT_MAX_KT; ++t)
        if ((nzb >> (t * SPT)) & ((1ull << SPT) - 1)) q_live_mask |= 1u << t;

    // A query tile whose dO is all +0 (padded positions under a [MASK]-only loss) has P o (dP - delta) = 0: nothing
    // reaches dK / dV and its dQ rows are +0 -- the bits the dense path would write.  Those rows are zeroed here and
    // the tile loop visits the other tiles only (the mask is uniform over the workgroup).
    for (int qt = 0; qt < nkt; ++qt) {
        if (!((q_live_mask >> qt) & 1u) && tid < 32 * CH) {
            const int q = qt * 32 + tid / CH;
            if (q < S) *reinterpret_cast<u32x4 *>(dqkv + (tok0 + q) * ld_dq + hh * DH + (tid % CH) * 8) = (u32x4){0u, 0u, 0u, 0u};
        }
    }
    // this wave's K / V fragments do not change over the query tiles
    bf16x8 fkr[NKS], fvr[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int krow = (kt < nkt ? kt : 0) * 32 + r;
        fkr[ks] = *reinterpret_cast<const bf16x8 *>(sK + krow * KSTR + ks * 32 + hf * 16);
        fvr[ks] = *reinterpret_cast<const bf16x8 *>(sV + krow * KSTR + ks * 32 + hf * 16);
    }
    for (unsigned todo = q_live_mask & ((1u << nkt) - 1u); todo; todo &= todo - 1u) {
        const int qt = __builtin_ctz(todo);
        const int q0 = qt * 32;
        const char *sQ = sQa + q0 * KSTR, *sdO = sGa + q0 * KSTR;
        const float *sLse = sLseA + q0, *sDelta = sDeltaA + q0;
        if (kt < nkt && tile_live) {
            f32x16 sa, pa;
#pragma unroll
            for (int t = 0; t < 16; ++t) { sa[t] = 0.f; pa[t] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 fq = *reinterpret_cast<const bf16x8 *>(sQ + r * KSTR + ks * 32 + hf * 16);
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq, fkr[ks], sa, 0, 0, 0);
                const bf16x8 fg = *reinterpret_cast<const bf16x8 *>(sdO + r * KSTR + ks * 32 + hf * 16);
                pa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg, fvr[ks], pa, 0, 0, 0);
            }
            float pv[16], dsv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int q = rowmap(t, hf);
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[t], scale2, madd - sLse[q]));
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
                    const int off = (16 * s2 + 4 * hf + (li >> 2)) * KSTR + (dt * 32 + 16 * (g & 1) + 4 * (li & 3)) * 2;
                    const bf16x8 fgt = frag_tr(sdO + off, 8 * KSTR);
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fgt, pf, dv[dt], 0, 0, 0);
                    const bf16x8 fqt = frag_tr(sQ + off, 8 * KSTR);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fqt, df, dk[dt], 0, 0, 0);
                }
            }
        }
        B4C_LDS_BARRIER();                                // dS tile complete (LDS only: dQ stores of the previous tile keep draining)
        if (wave < 2 * (DH / 16)) {
            const int qi = wave / (DH / 16), di = wave % (DH / 16);
            f32x4 qa = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < nkt; ++ks) {
                const bf16x8 fs = *reinterpret_cast<const bf16x8 *>(sDS + (qi * 16 + li) * TSTR + (ks * 32 + 8 * g) * 2);
                const char *kb = sK + (ks * 32 + 8 * g + (li >> 2)) * KSTR + (di * 16 + 4 * (li & 3)) * 2;
                const bf16x8 fk = frag_tr(kb, 4 * KSTR);
                qa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fs, fk, qa, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *reinterpret_cast<bf16_t *>(sDQ + (qi * 16 + 4 * g + t) * KSTR + (di * 16 + li) * 2) = (bf16_t)(qa[t] * scale);
        }
        B4C_LDS_BARRIER();                                // dS consumed, dQ tile staged
        if (tid < 32 * CH) {                              // one 16-B chunk per lane: whole rows per store instruction
            const int row = tid / CH, part = tid % CH;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(sDQ + row * KSTR + part * 16);
            if (q0 + row < S) att_store16<B4C_NT(B4C_NT_ATTN_DQKV)>(dqkv + (tok0 + q0 + row) * ld_dq + hh * DH + part * 8, v);
        }
    }
    ATT_STAMP(2);
    // dK / dV: each wave turns its accumulators (dh on the registers, key on the lane) into row-major bf16 through its
    // own slice of LDS (every image is dead after the last barrier), then writes whole 16-B chunks: full rows per
    // store instruction instead of 8 B per lane scattered over 64 rows.
    if (kt < nkt) {
        char *stK = smem + wave * (2 * 32 * KSTR), *stV = stK + 32 * KSTR;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                bf16x4 wk, wv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wk[j] = (bf16_t)(dk[dt][4 * tq + j] * scale);
                    wv[j] = (bf16_t)dv[dt][4 * tq + j];
                }
                *reinterpret_cast<bf16x4 *>(stK + r * KSTR + (dt * 32 + 8 * tq + 4 * hf) * 2) = wk;
                *reinterpret_cast<bf16x4 *>(stV + r * KSTR + (dt * 32 + 8 * tq + 4 * hf) * 2) = wv;
            }
#pragma unroll
        for (int i = 0; i < 32 * CH / 64; ++i) {
            const int c = lane + 64 * i, row = c / CH, part = c % CH;
            const u32x4 k4 = *reinterpret_cast<const u32x4 *>(stK + row * KSTR + part * 16);
            const u32x4 v4 = *reinterpret_cast<const u32x4 *>(stV + row * KSTR + part * 16);
            if (kt * 32 + row < S) {
                bf16_t *krow = dqkv + (tok0 + kt * 32 + row) * ld_dq + dm + hh * DH + part * 8;
                att_store16<B4C_NT(B4C_NT_ATTN_DQKV)>(krow, k4);
                att_store16<B4C_NT(B4C_NT_ATTN_DQKV)>(krow + dm, v4);
            }
        }
    }
#ifdef B4C_ATTN_PHASES
    if (tid == 0 && g_attn_phases) {
        att_t[3] = wall_clock64();
        for (int i = 0; i < 4; ++i) g_attn_phases[(size_t)item * 4 + i] = att_t[i];
    }
#endif
    B4C_LDS_BARRIER();            // the staging slices are read back before the next item's images land
    item = __builtin_amdgcn_readfirstlane(*sNext);
    }
}

// ------------------------------------------------------------------------------------------
static int att_num_cus() {
    static thread_local int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
#define ATT_MAX_S 512      // two query tiles per wave in the forward; key blocks of 256 in the backward
static bool mfma_shape_ok(int S, int dh) { return (dh == 32 || dh == 64) && S <= ATT_MAX_S; }

int b4c_attn_fwd_mfma(const void *qkv, int ld_qkv, const uint8_t *key_pad, void *o, int ld_o, float *lse, int B, int S,
                      int H, int dh, const int32_t *cu, hipStream_t st) {
    if (!mfma_shape_ok(S, dh)) return B4C_EUNSUPPORTED;
    const int S_pad = (S + 31) / 32 * 32;
    const int qpw = S_pad > 32 * ATT_MAX_KT ? 2 : 1;
    const size_t shm = 2 * (size_t)S_pad * (dh * 2 + 16) + (size_t)S_pad * 4 + ATT_MAX_KT * qpw * 4;
    const float scale = 1.0f / sqrtf((float)dh);
#define ATT_FWD_LAUNCH(DHH, QQ)                                                                                          \
    do {                                                                                                                 \
        allow_lds_attn(attn_fwd_mfma_kernel<DHH, QQ>, shm);                                                              \
        attn_fwd_mfma_kernel<DHH, QQ><<<B * H, 512, shm, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (bf16_t *)o, ld_o, lse, S, H, scale, cu); \
    } while (0)
    if (dh == 64) { if (qpw == 1) ATT_FWD_LAUNCH(64, 1); else ATT_FWD_LAUNCH(64, 2); }
    else { if (qpw == 1) ATT_FWD_LAUNCH(32, 1); else ATT_FWD_LAUNCH(32, 2); }
#undef ATT_FWD_LAUNCH
    return b4c_check_launch("attn_fwd_mfma");
}

int64_t b4c_attn_bwd_mfma_workspace_bytes(int B, int S, int H, int dh) {
    if (!mfma_shape_ok(S, dh) || S <= 32 * ATT_MAX_KT) return 0;
    return (int64_t)B * S * H * dh * (int64_t)sizeof(float);       // fp32 dQ accumulator across the key blocks
}

int b4c_attn_bwd_mfma(const void *qkv, int ld_qkv, const uint8_t *key_pad, const void *o, int ld_o, const void *d_o,
                      int ld_do, const float *lse, float *delta, void *dqkv, int ld_dqkv, int B, int S, int H, int dh,
                      void *workspace, int64_t workspace_bytes, const int32_t *cu, hipStream_t st) {
    if (!mfma_shape_ok(S, dh)) return B4C_EUNSUPPORTED;
    const float scale_s = 1.0f / sqrtf((float)dh);
    if (S > 32 * ATT_MAX_KT) {
        // the blocks of 256 keys one after the other inside one launch; their partial dQ sums meet in an fp32 accumulator
        // (caller's workspace)
        if (!workspace || workspace_bytes < b4c_attn_bwd_mfma_workspace_bytes(B, S, H, dh)) return B4C_EUNSUPPORTED;
        const size_t kstr2 = dh * 2 + 16, tstr2 = 256 * 2 + 16;
        const size_t shm2 = 2 * 256 * kstr2 + 2 * (2 * 32 * kstr2 + 256) + 32 * tstr2;
        const int nblk = (S + 255) / 256;
        if (dh == 64) {
            allow_lds_attn(attn_bwd_mfma_kernel<64>, shm2);
            attn_bwd_mfma_kernel<64><<<B * H, 512, shm2, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, lse, (bf16_t *)dqkv, ld_dqkv, S, H, scale_s, nblk, (float *)workspace, cu);
        } else {
            allow_lds_attn(attn_bwd_mfma_kernel<32>, shm2);
            attn_bwd_mfma_kernel<32><<<B * H, 512, shm2, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, lse, (bf16_t *)dqkv, ld_dqkv, S, H, scale_s, nblk, (float *)workspace, cu);
        }
        return b4c_check_launch("attn_bwd_mfma (key blocks)");
    }
    const int S_pad = (S + 31) / 32 * 32;
    const size_t kstr = dh * 2 + 16, tstr = S_pad * 2 + 16;
    const size_t shm = 2 * S_pad * kstr + 2 * (2 * 32 * kstr + 256) + 32 * tstr;
    const size_t shm_res = 4 * S_pad * kstr + 32 * tstr + 2 * (size_t)S_pad * 4 + ATT_MAX_KT * 4 * 4 + 16 + 32 * kstr;
    const float scale = 1.0f / sqrtf((float)dh);
    if (shm_res <= 160 * 1024) {
        const int grid_res = B * H < att_num_cus() ? B * H : att_num_cus();   // > 80 KB of LDS: one workgroup per CU
        // delta itself is computed while staging dO / O; its first word is the work counter of the persistent grid
        if (hipMemsetAsync(delta, 0, sizeof(int), st) != hipSuccess) return B4C_ELAUNCH;
        if (dh == 64) {
            allow_lds_attn(attn_bwd_resident_kernel<64>, shm_res);
            attn_bwd_resident_kernel<64><<<grid_res, 512, shm_res, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, lse, (bf16_t *)dqkv, ld_dqkv, S, H, scale, B * H, (int *)delta, cu);
        } else {
            allow_lds_attn(attn_bwd_resident_kernel<32>, shm_res);
            attn_bwd_resident_kernel<32><<<grid_res, 512, shm_res, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, lse, (bf16_t *)dqkv, ld_dqkv, S, H, scale, B * H, (int *)delta, cu);
        }
        return b4c_check_launch("attn_bwd_resident");
    }
    if (dh == 64) {
        allow_lds_attn(attn_bwd_mfma_kernel<64>, shm);
        attn_bwd_mfma_kernel<64><<<B * H, 512, shm, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, lse, (bf16_t *)dqkv, ld_dqkv, S, H, scale, 1, nullptr, cu);
    } else {
        allow_lds_attn(attn_bwd_mfma_kernel<32>, shm);
        attn_bwd_mfma_kernel<32><<<B * H, 512, shm, st>>>((const bf16_t *)qkv, ld_qkv, key_pad, (const bf16_t *)o, ld_o, (const bf16_t *)d_o, ld_do, lse, (bf16_t *)dqkv, ld_dqkv, S, H, scale, 1, nullptr, cu);
    }
    return b4c_check_launch("attn_bwd_mfma");
}
