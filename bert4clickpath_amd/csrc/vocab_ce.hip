// Vocabulary projection + softmax + masked sparse cross-entropy WITHOUT the (rows x V) logits in HBM
// (bf16 throughput path of R12-R14: head.py:36 Dense(V, softmax), utils.py:56-134, losses.py:31-98).
//
// At C2 the materialised form moves the 4.1 GB logits tensor through HBM five times per step (projection
// write, CE read + write, dX read, dW read).  The MI355X has ~300 FLOP per HBM byte to spare, so the logits
// are recomputed instead: 128 x 128 tiles of  x = h W^T + b  live only in MFMA accumulators, twice
// (three times for rows that TF's clip touches):
//
//   sweep 1  (token-owned, vce_token_kernel<K,1>)  online softmax against a lazily raised reference (flash
//            attention with V = W): p' = 2^(x log2e - m2) feeds  U = P' W  from the accumulator registers (the
//            P'^T tile is the B operand of the next MFMA); per row: m2, l = sum p', min x, max x -> lse, clip flag
//   sweep 1b (vce_token_kernel<K,2>)  only for rows whose probabilities leave [1e-7, 1-1e-7] (TF's clip,
//            backend.py sparse_categorical_crossentropy), against the row's final lse: Ud = P (1-u) W, Pc = sum (1-u) p,
//            nu = sum u  (u = 1 inside the clip range);  Pu = 1 - Pc,  S = sum clip(p) = Pu + 1e-7 n_low + (1 - 1e-7) n_high
//   combine  (one wave per row)  loss, dh = gs ((U - Ud) / S - G U - yd W_y), row scalars for sweep 2
//   sweep 2  (vocabulary-owned, vce_dw_kernel)  dlogit = p (u a - b) feeds  dW^T = h^T dlogit  from registers;
//            db = column sums
//   label term  dW[:, y] -= yd h_row, db[y] -= yd  (coalesced scatter into a vocabulary-major scratch, added transposed)
//
// HBM traffic: h, W (L2 / MALL resident, 12.8 MB), per-row partial sums.  Work: 4 GEMM units + 2 R V exps.
// W / h tiles arrive by LDS-DMA (buffer_load ... lds) into an XOR-swizzled image that both the direct
// (ds_read_b128) and the transposed (ds_read_b64_tr_b16) fragment reads hit without bank conflicts.
// B4C_VCE_TIMING=1 prints per-kernel HIP-event times of the previous call (no synchronisation).
//
// MFMA 32x32x16 bf16 maps (lane l: r = l & 31, hf = l >> 5): A[row r][k = 8 hf + j], B[k = 8 hf + j][col r],
// D reg t: row (t&3) + 8 (t>>2) + 4 hf, col r.  An accumulator tile used as B operand of the next MFMA sums
// over its rows in the order 16 s + 8 (j>>2) + 4 hf + (j&3); the A operand reads the same order through
// ds_read_b64_tr_b16.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned vu32x4;
typedef __attribute__((ext_vector_type(4))) short vs16x4;
typedef __attribute__((ext_vector_type(8))) short vs16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define VCE_EPS 1e-7f
#define VCE_LOG2E 1.4426950408889634f
#define VCE_LN2 0.6931471805599453f

__device__ __forceinline__ int vce_rowmap(int t, int hf) { return (t & 3) + 8 * (t >> 2) + 4 * hf; }
__device__ __forceinline__ bf16x8 vce_pack8(const float *p) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)p[j];
    return v;
}
__device__ __forceinline__ bf16x8 vce_frag_tr(const char *p, int second_off) {
    const vs16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vs16x4 __attribute__((address_space(3))) *)(p));
    const vs16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vs16x4 __attribute__((address_space(3))) *)(p + second_off));
    const vs16x8 w = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vce_rsrc(const void *base, int64_t rows, int64_t row_bytes) {
    int64_t bytes = (rows < 0 ? 0 : rows) * row_bytes;
    if (bytes > 0x3FFFFFF0ll) bytes = 0x3FFFFFF0ll;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (unsigned)bytes, 0x00020000);
}

// A [128 rows][KD] bf16 tile in LDS, filled by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write).
// Rows are unpadded (KD * 2 bytes); the 16-B chunk c of row j sits at chunk c ^ f(j):
//   KD = 128 (a row = one 256-B bank row):   f(j) = ((j & 3) << 2) | ((j >> 2) & 3)
//   KD =  64 (two rows per bank row):        f(j) = (((j >> 1) & 1) << 2) | ((j >> 2) & 3)
// so that a transposed read (32-lane group: 4 consecutive rows x the same 64 B) and a direct fragment read
// (ds_read_b128 16-lane groups: rows {0-3,12-15,20-27} / {4-11,16-19,28-31}, same chunk) both land on distinct
// 16-B slots of the 64 banks (MI355X_MICROARCH.md, LDS).  An LDS-DMA wave instruction writes 64 x 16 B
// contiguously (lane l -> base + 16 l), so the swizzle is applied to the SOURCE address of each lane.
template <int KD> struct VTile {
    static constexpr int CH = KD / 8;            // 16-B chunks per row
    static constexpr int NIT = 128 * CH / 512;   // DMA instructions per thread: 4 (KD = 128) or 2 (KD = 64)
    static constexpr int STR = KD * 2;
    static constexpr int BYTES = 128 * STR;
    static __device__ __forceinline__ int swz(int row) {
        return KD == 128 ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
    }
    static __device__ __forceinline__ int chunk_off(int row, int chunk) { return row * STR + ((chunk ^ swz(row)) << 4); }
    // rows [row0, row0 + 128) of P (row pitch ld elements) -> LDS tile at `dst`; rows >= nrows arrive as zeros.
    // Completion is on the VM counter: s_waitcnt vmcnt(0) + a barrier before any wave reads the tile.
    template <int NT = 512>
    static __device__ __forceinline__ void dma(const bf16_t *__restrict__ P, int ld, int64_t row0, int64_t nrows, char *dst, int tid) {
        const int64_t left = nrows - row0;
        const __amdgpu_buffer_rsrc_t rs = vce_rsrc(P + row0 * ld, left < 128 ? left : 128, (int64_t)ld * 2);
#pragma unroll
        for (int i = 0; i < NIT * 512 / NT; ++i) {
            const int c = tid + i * NT, row = c / CH, slot = c % CH;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + ((c & ~63) << 4)), 16,
                                                     (row * ld + ((slot ^ swz(row)) << 3)) * 2, 0, 0, 0);
        }
    }
    // one DMA instruction of the same transfer (piece i of NIT * 512 / NT; a 256-thread workgroup spreads them over its loop)
    template <int NT>
    static __device__ __forceinline__ void dma_piece(const bf16_t *__restrict__ P, int ld, int64_t row0, int64_t nrows, char *dst, int tid, int i) {
        const int64_t left = nrows - row0;
        const __amdgpu_buffer_rsrc_t rs = vce_rsrc(P + row0 * ld, left < 128 ? left : 128, (int64_t)ld * 2);
        const int c = tid + i * NT, row = c / CH, slot = c % CH;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + ((c & ~63) << 4)), 16,
                                                 (row * ld + ((slot ^ swz(row)) << 3)) * 2, 0, 0, 0);
    }
    // The same piece as inline assembly, for a loop whose LDS slots are run-time values: through the builtin the compiler
    // cannot tell the DMA's destination from the slots the loop's ds_reads address and parks every wave on vmcnt(0) after
    // each piece (400 cycles per piece measured).  Here it sees no LDS write at all: the caller orders the tile's arrival
    // against its first read itself (s_waitcnt vmcnt(0) + barrier), as every sweep of this file does anyway.
    // lds_base: LDS byte address of the slot (wave-uniform); wave: the wave's index (wave-uniform).
    template <int NT>
    static __device__ __forceinline__ void dma_piece_asm(const bf16_t *__restrict__ P, int ld, int64_t row0, int64_t nrows, unsigned lds_base,
                                                         int wave, int lane, int i) {
        const int64_t left = nrows - row0;
        int64_t bytes = (left < 0 ? 0 : (left < 128 ? left : 128)) * (int64_t)ld * 2;
        if (bytes > 0x3FFFFFF0ll) bytes = 0x3FFFFFF0ll;
        const uint64_t base = (uint64_t)(P + row0 * ld);
        vu32x4 rs;
        rs[0] = __builtin_amdgcn_readfirstlane((unsigned)base);
        rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xFFFFu);
        rs[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
        rs[3] = 0x00020000u;
        const int c = wave * 64 + lane + i * NT, row = c / CH, slot = c % CH;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((wave * 64 + i * NT) << 4));
        const unsigned voff = (unsigned)((row * ld + ((slot ^ swz(row)) << 3)) * 2);
        // (s_nop 0: one wait state between a SALU write of M0 and an LDS-DMA that reads it -- csrc/dxdw_common.h dd_dma)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(voff), "s"(rs) : "m0");
    }
    // per-lane offsets, relative to a row base that is a multiple of 16 rows:
    //   direct fragment (row r, k-step ks, half hf): 16 B
    static __device__ __forceinline__ int frag_off(int r, int ks, int hf) { return chunk_off(r, 2 * ks + hf); }
    //   transposed fragment piece: rows 4 hf + (li >> 2) (+ 8 for the second piece), columns dt*32 + 16 (g&1) + 4 (li&3)
    static __device__ __forceinline__ int tr_off(int hf, int li, int g, int dt, int second) {
        const int row = 4 * hf + (li >> 2) + 8 * second;
        const int e = dt * 32 + 16 * (g & 1) + 4 * (li & 3);
        return chunk_off(row, e >> 3) + (e & 7) * 2;
    }
};
#define VCE_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
__device__ __forceinline__ bf16x8 vce_frag_tr2(const char *p0, const char *p1) {
    const vs16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vs16x4 __attribute__((address_space(3))) *)(p0));
    const vs16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vs16x4 __attribute__((address_space(3))) *)(p1));
    const vs16x8 w = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, w);
}

// The lane's token row of h as MFMA B fragments (KD / 16 k-steps)
template <int KD>
__device__ __forceinline__ void vce_load_hfrag(const bf16_t *__restrict__ h, int ld_h, int64_t tok, int64_t R, int hf, bf16x8 (&f)[KD / 16]) {
#pragma unroll
    for (int ks = 0; ks < KD / 16; ++ks) {
        vu32x4 v = {0u, 0u, 0u, 0u};
        if (tok < R) v = *reinterpret_cast<const vu32x4 *>(h + tok * ld_h + ks * 16 + hf * 8);
        f[ks] = __builtin_bit_cast(bf16x8, v);
    }
}

struct VceArgs {
    const bf16_t *h;      // [R][ld_h]
    const bf16_t *wt;     // [V][ld_w]  (vocab-major rows of KD)
    const float *bias;    // [V] or NULL
    const int32_t *labels;
    const float *grad_scale;   // device scalar: d(total loss) / d(row loss)
    float *st1;           // [parts][R][4]: m2 (log2-domain reference), l = sum 2^(x log2e - m2), min x, max x
    float *u;             // [parts][R][KD]: sum 2^(x log2e - m2) W   (un-normalised P W)
    float *ud;            // [parts][R][KD]: Ud = sum over the p outside the clip range of p W (normalised)
    float *sp;            // [parts][R][4]: nu (entries inside the clip range, the row's dominant one apart), Pc (sum of the p below it), ntop (entries above 1/2: 0 or 1), -
    float *rowscal;       // [R][8]: lse2, c = a - b, nb = -b, lo (lower clip bound, -inf on rows that stay inside), gs yd, 1 = every probability outside the range (c = nb = 0), -, -
    float *item_loss;     // [R]
    bf16_t *dh;           // [R][ld_dh]
    int ld_h, ld_w, ld_dh;
    int64_t R;
    int V, parts, variant;
    int ntt;              // 128-token tiles
    // "lse first" form of the forward (vce_exact_kernel): the row's lse and largest logit are known before the one sweep that
    // accumulates U and Ud, both against that final lse
    const float *rowstat; // [R][2]: lse2, max x  (NULL: the online form, statistics per part in st1)
    int parts_st;         // vocabulary parts of the sweep that filled st1 (the lse sweep's own split)
};

// merge the per-part statistics of one row: lse2 = log2 sum_j 2^(x_j log2e), clipped flag, and (optionally) the
// factor f_p = 2^(m2_p - M2) / l that turns part p's un-normalised sums into probabilities
__device__ __forceinline__ void vce_row_stats(const VceArgs &a, int64_t row, float &lse2, bool &clipped, float &pmax) {
    if (a.rowstat) {          // the lse-first form: one merged record per row; whether the row is clipped is read off the sweep's counts
        const f32x2 s = *reinterpret_cast<const f32x2 *>(a.rowstat + row * 2);
        lse2 = s[0];
        pmax = __builtin_amdgcn_exp2f(s[1] * VCE_LOG2E - lse2);
        clipped = a.variant == B4C_CE_TF;
        return;
    }
    float M = -INFINITY, l = 0.f, mn = INFINITY, mx = -INFINITY;
    for (int p = 0; p < a.parts; ++p) {
        const f32x4 s = *reinterpret_cast<const f32x4 *>(a.st1 + ((int64_t)p * a.R + row) * 4);
        const float M2 = fmaxf(M, s[0]);
        l = l * __builtin_amdgcn_exp2f(M - M2) + s[1] * __builtin_amdgcn_exp2f(s[0] - M2);
        M = M2;
        mn = fminf(mn, s[2]);
        mx = fmaxf(mx, s[3]);
    }
    lse2 = M + __log2f(l);
    const float pmin = __builtin_amdgcn_exp2f(mn * VCE_LOG2E - lse2);
    pmax = __builtin_amdgcn_exp2f(mx * VCE_LOG2E - lse2);
    clipped = (a.variant == B4C_CE_TF) && (pmin < VCE_EPS || pmax > 1.0f - VCE_EPS);
}

// ------------------------------------------------------------------------------------------
// K2: one workgroup = 128 tokens x one part of the vocabulary; 8 waves = 4 token groups (32 tokens on the
// lanes) x 2 vocabulary halves of each 128-row W tile.
//   MODE 1: online softmax (as flash attention with V = W): p' = 2^(x log2e - m2) against a lazily updated
//           per-token reference m2 (raised, with U and l rescaled, only when a logit exceeds it by 2^12), P'^T
//           feeds U += W^T P'^T from the accumulator registers.  One sweep gives lse and P W.
//   MODE 2: second sweep for the tokens whose probabilities leave [1e-7, 1 - 1e-7], against the row's final lse:
//           Ud = P (1 - u) W (the part OUTSIDE the clip range), Pc = sum (1 - u) p, nu = sum u.  The clipped sum follows
//           from the counts (Pu = 1 - Pc, S = Pu + eps n_low + (1 - eps) n_high: vce_combine_kernel) -- per entry: exp,
//           compare, select, add, count (accumulating S, Pu and the clipped part took eight VALU instructions per entry,
//           this takes six).  Ud and not Uu: in dh = ((U - Ud) / S - G U - ...) every product with U keeps ONE bf16
//           rounding of P (sweep 1's), so the O(1) terms cancel exactly where the true gradient is ~0 (a confident wrong
//           row: G ~ 1); with Uu from this sweep's own rounding of P they would leave 2^-9 |W| behind.
//   MODE 0: the statistics of MODE 1 alone (m2, l -> lse; no U, no min / max): the lse sweep in front of the
//           materialised softmax projection (b4c_vocab_lse).
// ------------------------------------------------------------------------------------------
#define VCE_LAZY 12.0f
#ifndef VCE_FRAG_BATCH
#define VCE_FRAG_BATCH 1
#endif

#ifdef VCE_SCAN_STAMPS
// diagnostic build only (scratch/scan_stamps.py): cycles per phase of the tile loop, summed per wave
__device__ unsigned long long g_vce_stamps[2048 * 8 * 6];
extern "C" int b4c_debug_vce_stamps(void *dst, size_t nbytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_vce_stamps), nbytes < sizeof(g_vce_stamps) ? nbytes : sizeof(g_vce_stamps));
}
#define VCE_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[k] += t_ - t0_; t0_ = t_; } while (0)
__device__ unsigned long long g_vce_xstamps[2048 * 4 * 8];       // vce_exact_kernel: per wave, per pipeline step
extern "C" int b4c_debug_vce_xstamps(void *dst, size_t nbytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_vce_xstamps), nbytes < sizeof(g_vce_xstamps) ? nbytes : sizeof(g_vce_xstamps));
}
#else
#define VCE_STAMP(k) do { } while (0)
#endif

// NH = 1: 128 tokens per workgroup, 8 waves = 4 token groups x the 2 halves of each 128-row W tile.
// NH = 2: 256 tokens per workgroup, 8 token groups, every wave takes both halves in turn.  A 32 KB W tile takes a CU about
// 2,900 cycles to pull in by LDS-DMA (~11 B / clock / CU, whatever the source: the per-CU load path, MI355X_MICROARCH.md
// 'ldsdma-fill'), and at 128 tokens the two waves of a SIMD have 2 x 32 MFMAs = 2,048 cycles of matrix work per tile: the
// sweep waits for its tiles (per-phase stamps, scratch/token_stamps.py: 20 - 34 % of a wave's cycles at the end-of-tile
// wait).  At 256 tokens the tile's matrix work (4,096 cycles) covers its arrival.
template <int KD, int MODE, int NH>
__global__ void __launch_bounds__(512, 2) vce_token_kernel(VceArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NKS = KD / 16, NDT = KD / 32, STR = VTile<KD>::STR;
    constexpr int TILE_B = VTile<KD>::BYTES;
    float *sBias = reinterpret_cast<float *>(smem + 2 * TILE_B);     // [2][128]
    const int unit = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int li = lane & 15, g = lane >> 4;
    const int tg = NH == 2 ? wave : (wave & 3), vh = NH == 2 ? 0 : (wave >> 2);
    const int64_t tok0 = (int64_t)(unit % a.ntt) * (128 * NH);
    const int64_t tok = tok0 + tg * 32 + r;
    const int part = unit / a.ntt;
    const int nvt = (a.V + 127) >> 7;
    const int vt0 = (int)((int64_t)nvt * part / a.parts), vt1 = (int)((int64_t)nvt * (part + 1) / a.parts);

    float lse2 = INFINITY;   // MODE 2: log2-domain lse of the lane's token
    bool anyhi = false;      // MODE 2: some token of this tile has a DOMINANT entry (p > 1/2), the only kind that can pass 1 - 1e-7
    if (MODE == 2) {
        bool clipped = false;
        float pmax = 0.f;
        if (tok < a.R) vce_row_stats(a, tok, lse2, clipped, pmax);
        if (!__syncthreads_or(clipped)) return;          // no clipped row in these 128 tokens (block-uniform)
        // (pmax is 2^(x_max log2e - lse2): good to a few 1e-6 at logits of +-50 -- enough to see a dominant entry, useless to
        // tell 1 - 1e-7 from 1: that decision is taken from the OTHER entries' mass, vce_combine_kernel)
        anyhi = __syncthreads_or(pmax > 0.4f);
    }
    bf16x8 hfr[NKS];
    vce_load_hfrag<KD>(a.h, a.ld_h, tok, a.R, hf, hfr);
    int foff[NKS], toff[NDT][2];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) foff[ks] = VTile<KD>::frag_off(r, ks, hf) + vh * 64 * STR;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
        toff[dt][0] = VTile<KD>::tr_off(hf, li, g, dt, 0) + vh * 64 * STR;
        toff[dt][1] = VTile<KD>::tr_off(hf, li, g, dt, 1) + vh * 64 * STR;
    }

    // tile vt -> LDS buffer `buf` (DMA), its bias -> a register (stored to LDS after the current tile's reads)
    float breg = 0.f;
    auto fetch = [&](int vt, int buf) {
        VTile<KD>::template dma<512>(a.wt, a.ld_w, (int64_t)vt * 128, vt < vt1 ? a.V : 0, smem + buf * TILE_B, tid);
        if (tid < 128) {
            const int v = vt * 128 + tid;
            breg = (vt < vt1 && v < a.V) ? (a.bias ? a.bias[v] : 0.f) : -INFINITY;   // rows past V: logit = -inf
        }
    };
    fetch(vt0, 0);
    if (tid < 128) sBias[tid] = breg;
    VCE_DMA_WAIT();
    __syncthreads();

    // running state of the lane's token over its (hf, vh) share of the vocabulary
    float m2 = -INFINITY, l = 0.f, mn = INFINITY, mx = -INFINITY;   // MODE 1
    float Pc = 0.f;                                                 // MODE 2: sum of the probabilities outside the clip range
    unsigned nu = 0, nhi = 0;                                       // MODE 2: entries inside / above the clip range
    f32x16 U[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int t = 0; t < 16; ++t) U[dt][t] = 0.f;

#ifdef VCE_SCAN_STAMPS
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#endif
    // one vocabulary tile; the LDS buffer index is a compile-time constant (the loop is unrolled by two) so that every
    // LDS address is a per-lane VGPR + an immediate
    auto tile = [&](auto BUF, int vt) {
        constexpr int buf = decltype(BUF)::value;
        VCE_STAMP(5);
        fetch(vt + 1, buf ^ 1);        // the other buffer was last read one tile ago (behind the previous barrier)
        VCE_STAMP(0);
#pragma unroll
        for (int hv = 0; hv < NH; ++hv) {            // NH = 2: the wave takes the tile's two 64-row halves in turn
        const int vhe = NH == 2 ? hv : vh;           // which half
        const char *w = smem + buf * TILE_B + (NH == 2 ? hv * 64 * STR : 0);
        const float *bs = sBias + buf * 128;
        f32x16 acc[2];
#if VCE_FRAG_BATCH
        // The W fragments of a 32-row tile are fetched as a batch in front of its MFMA chain, and the second tile's while
        // the first chain runs (left to itself the compiler sends every fragment through one register quad: ds_read_b128 ->
        // s_waitcnt lgkmcnt(0) -> v_mfma, sixteen times).  Same MFMA order: bit-identical results; 1.5 - 3 % per sweep.
        {
            bf16x8 wfq[NKS];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bs + vhe * 64 + rt * 32 + 8 * tq + 4 * hf);
                    acc[rt][4 * tq] = b4[0]; acc[rt][4 * tq + 1] = b4[1]; acc[rt][4 * tq + 2] = b4[2]; acc[rt][4 * tq + 3] = b4[3];
                }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wfq[ks] = *reinterpret_cast<const bf16x8 *>(w + foff[ks]);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfq[ks], hfr[ks], acc[0], 0, 0, 0);
                wfq[ks] = *reinterpret_cast<const bf16x8 *>(w + 32 * STR + foff[ks]);
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfq[ks], hfr[ks], acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8 + NKS, 0);         // bias quads + the first tile's fragments
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NKS, 0);
        }
#else
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bs + vhe * 64 + rt * 32 + 8 * tq + 4 * hf);
                acc[rt][4 * tq] = b4[0]; acc[rt][4 * tq + 1] = b4[1]; acc[rt][4 * tq + 2] = b4[2]; acc[rt][4 * tq + 3] = b4[3];
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8 *>(w + rt * 32 * STR + foff[ks]);
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, hfr[ks], acc[rt], 0, 0, 0);
            }
        }
#endif
        VCE_STAMP(1);
        float e2 = lse2;                             // the exponent reference of this tile
        if (MODE != 2) {
            const bool tail = (vt + 1) * 128 > a.V;      // some rows of this tile are past V (their logit is -inf)
            float tm = -INFINITY;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int t = 0; t < 16; ++t) tm = fmaxf(tm, acc[rt][t]);
            if (MODE == 0) {
            } else if (!tail) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int t = 0; t < 16; ++t) mn = fminf(mn, acc[rt][t]);
            } else {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (vt * 128 + vhe * 64 + rt * 32 + vce_rowmap(t, hf) < a.V) mn = fminf(mn, acc[rt][t]);
            }
            // the two lanes of a token share the reference (their P mix in U): lanes l and l + 32 exchange through
            // v_permlane32_swap (one VALU instruction; __shfl_xor is a ds_bpermute: an LDS round trip on the critical path
            // of every tile)
            // The instruction is written out: through __builtin_amdgcn_permlane32_swap the compiler (ROCm 7.2) keeps the first
            // result only and drops the max below -- also with an opaque copy as the second operand (round 3 shipped that for a
            // while: both lanes then carried the LOWER lane's maximum; lse stayed right, but the row maximum, the clip flags
            // derived from it and, at logit spreads beyond 88, the running sum itself did not; tests/test_gpu_properties.py
            // found it).
            {
                float lo_half = tm, hi_half = tm;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo_half), "+v"(hi_half));
                tm = fmaxf(lo_half, hi_half);      // lanes l < 32: (own, lane l + 32's); lanes l >= 32: (lane l - 32's, own)
            }
            mx = fmaxf(mx, tm);
            const float tm2 = tm * VCE_LOG2E;
            const bool raise = tm2 > m2 + VCE_LAZY;
            if (__any(raise)) {                       // rare after the first tile
                if (raise) {
                    const float al = __builtin_amdgcn_exp2f(m2 - tm2);     // 0 on the first tile (m2 = -inf)
                    m2 = tm2;
                    l *= al;
                    if (MODE == 1) {
#pragma unroll
                        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                            for (int t = 0; t < 16; ++t) U[dt][t] *= al;
                    }
                }
            }
            e2 = (m2 == -INFINITY) ? 0.f : m2;      // no finite logit seen yet (a tail half-tile past V): p = 2^(-inf) = 0
        }
        VCE_STAMP(2);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            __builtin_amdgcn_sched_barrier(0);       // keep one 32-row tile's temporaries live at a time
            float p[16];
            if (MODE == 2 && anyhi) {
                // (block-uniform) some row here has a dominant entry, the one probability that may exceed 1 - 1e-7.  Whether it
                // does is NOT read off its own value (1 - p is below fp32 resolution exactly when it matters): the entry is
                // counted apart, kept out of Pc / Ud, and the combine kernel decides from the mass of all the others.
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[rt][t], VCE_LOG2E, -e2));
                    const bool low = pv < VCE_EPS, top = pv > 0.5f;
                    p[t] = low ? pv : 0.f;
                    Pc += p[t];
                    nu += (low || top) ? 0u : 1u;
                    nhi += top ? 1u : 0u;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[rt][t], VCE_LOG2E, -e2));
                    if (MODE != 2) {
                        l += pv;
                        p[t] = pv;
                    } else {
                        // rows past V carry logit -inf: p = 0, below the range, adds nothing, not counted -- no validity test
                        const bool un = pv >= VCE_EPS;
                        p[t] = un ? 0.f : pv;              // the part OUTSIDE the clip range feeds Ud
                        Pc += p[t];
                        nu += un ? 1u : 0u;
                    }
                }
            }
            if (MODE != 0)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = vce_pack8(p + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    // W^T[d = dt*32 + r][vocab rows 16 s2 + 4 hf + {0..3, 8..11} of this 32-row tile]
                    const char *wb = w + (rt * 32 + 16 * s2) * STR;
                    const bf16x8 wtf = vce_frag_tr2(wb + toff[dt][0], wb + toff[dt][1]);
                    U[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wtf, pf, U[dt], 0, 0, 0);
                }
            }
        }
        }
        VCE_STAMP(3);
        if (tid < 128) sBias[(buf ^ 1) * 128 + tid] = breg;
        VCE_DMA_WAIT();
        B4C_LDS_BARRIER();
        VCE_STAMP(4);
    };
    for (int vt = vt0; vt < vt1; vt += 2) {
        tile(std::integral_constant<int, 0>{}, vt);
        if (vt + 1 < vt1) tile(std::integral_constant<int, 1>{}, vt + 1);
    }
#ifdef VCE_SCAN_STAMPS
    if (lane == 0 && blockIdx.x < 2048)
        for (int k = 0; k < 6; ++k) g_vce_stamps[(blockIdx.x * 8 + wave) * 6 + k] = st_[k];
#endif
    __syncthreads();   // all tiles consumed: LDS is reused below

    // U^T tiles -> LDS [token][d] per wave; per-lane scalars -> LDS; then the two vocabulary halves (NH = 1) and the two
    // lanes of each token are merged and stored row-major
    constexpr int USTR = KD + 4;                         // floats per token row
    float *sU = reinterpret_cast<float *>(smem) + wave * 32 * USTR;
    if (MODE != 0)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const f32x4 v = {U[dt][4 * tq], U[dt][4 * tq + 1], U[dt][4 * tq + 2], U[dt][4 * tq + 3]};
            *reinterpret_cast<f32x4 *>(sU + r * USTR + dt * 32 + 8 * tq + 4 * hf) = v;
        }
    f32x4 *sS = reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(smem) + 8 * 32 * USTR);   // [wave][lane]
    sS[wave * 64 + lane] = (MODE != 2) ? (f32x4){m2, l, mn, mx} : (f32x4){(float)nu, Pc, (float)nhi, 0.f};
    __syncthreads();
    // token t of the tile: NH = 1: waves (t >> 5) and (t >> 5) + 4; NH = 2: wave t >> 5; lanes (t & 31) and (t & 31) + 32
    float *dst = (MODE == 1 ? a.u : a.ud) + (int64_t)part * a.R * KD;
    if (MODE != 0)
    for (int c = tid; c < 128 * NH * (KD / 4); c += 512) {
        const int t = c / (KD / 4), q = c % (KD / 4);
        if (tok0 + t < a.R) {
            const float *p0 = reinterpret_cast<const float *>(smem) + t * USTR + q * 4;
            f32x4 v0 = *reinterpret_cast<const f32x4 *>(p0);
            if (NH == 1) {
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(p0 + 4 * 32 * USTR);
                if (MODE == 1) {
                    const float ma = sS[(t >> 5) * 64 + (t & 31)][0], mb = sS[((t >> 5) + 4) * 64 + (t & 31)][0];
                    const float M = fmaxf(ma, mb);
                    v0 = v0 * __builtin_amdgcn_exp2f(ma - M) + v1 * __builtin_amdgcn_exp2f(mb - M);
                } else {
                    v0 = v0 + v1;
                }
            }
            *reinterpret_cast<f32x4 *>(dst + (tok0 + t) * KD + q * 4) = v0;
        }
    }
    if (tid < 128 * NH && tok0 + tid < a.R) {
        const int tgi = tid >> 5, ri = tid & 31;
        const f32x4 a0 = sS[tgi * 64 + ri], a1 = sS[tgi * 64 + ri + 32];
        // NH = 2: no second wave for the token: neutral elements
        const f32x4 nb = (MODE != 2) ? (f32x4){-INFINITY, 0.f, INFINITY, -INFINITY} : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 b0 = NH == 1 ? sS[(tgi + 4) * 64 + ri] : nb, b1 = NH == 1 ? sS[(tgi + 4) * 64 + ri + 32] : nb;
        if (MODE != 2) {
            const float M = fmaxf(a0[0], b0[0]);     // the two lanes of a token share m2
            const float fa = __builtin_amdgcn_exp2f(a0[0] - M), fb = __builtin_amdgcn_exp2f(b0[0] - M);
            *reinterpret_cast<f32x4 *>(a.st1 + ((int64_t)part * a.R + tok0 + tid) * 4) =
                (f32x4){M, (a0[1] + a1[1]) * fa + (b0[1] + b1[1]) * fb, fminf(fminf(a0[2], a1[2]), fminf(b0[2], b1[2])),
                        fmaxf(a0[3], b0[3])};
        } else {
            // (counts: exact in fp32)
            *reinterpret_cast<f32x4 *>(a.sp + ((int64_t)part * a.R + tok0 + tid) * 4) = a0 + a1 + b0 + b1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2x (round 4): the "lse first" form of the forward.  The row's lse is known (lse sweep: vce_token_kernel<KD, 0> +
// vce_rowstat_kernel), so ONE sweep forms the final probabilities p = 2^(x log2e - lse2) and accumulates BOTH products from the
// same P and the same W fragments:  U = P W  and  Ud = P (1 - u) W  (the entries below TF's clip range), with Pc, the counts
// and the dominant entry kept apart exactly as vce_token_kernel<KD, 2> keeps them.  No running maximum, no lazily raised
// reference, no rescaling: the max / raise chain of the online sweep (21 % of a tile there) does not exist.
//
// One workgroup = 128 tokens x one part of the vocabulary, 256 threads = ONE WAVE PER SIMD with up to 512 registers per lane:
// both accumulator sets (2 x 64 registers at K = 128), two logits tiles, double-buffered fragment sets.  A wave has no partner
// to overlap with, so the overlap is made inside the wave: the loop over the four 32-row tiles of a W tile is software-
// pipelined -- step k issues the logits MFMAs of tile k + 1 and the 16 P W MFMAs of tile k - 1 between the VALU
// instructions of tile k's probabilities (two thirds of an entry's ~6 instructions per MFMA, at most one exponential per gap:
// the issue costs of MI355X_MICROARCH.md fit the MFMA's 32 cycles), and the LDS fragment reads of step k + 1.
// sched_barrier(0) after every MFMA's group pins the interleave.  Three GEMM units per W tile (3,072 cycles of matrix pipe
// per SIMD) cover the tile's arrival by LDS-DMA (~2,900 cycles per CU) at 128 tokens per workgroup.
// ------------------------------------------------------------------------------------------
template <int KD> static size_t vce_exact_lds() {
    const size_t tiles = 3 * (size_t)VTile<KD>::BYTES + 3 * 128 * 4;          // a ring of three W tiles + their bias
    const size_t outs = (size_t)4 * 32 * (KD + 4) * 4 + 4 * 64 * 16;
    return tiles > outs ? tiles : outs;
}

template <int KD>
__global__ void __launch_bounds__(256, 1) vce_exact_kernel(VceArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NKS = KD / 16, NDT = KD / 32, STR = VTile<KD>::STR;
    constexpr int TILE_B = VTile<KD>::BYTES;
    constexpr int NPF = 2 * NDT;                        // transposed fragments of one 32-row tile
    constexpr int NDMA = VTile<KD>::NIT * 2;              // DMA instructions per thread and W tile (256 threads)
    constexpr int BIAS0 = 3 * TILE_B;                   // bias ring [3][128] floats behind the tile ring
    const int unit = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int64_t tok0 = (int64_t)(unit % a.ntt) * 128;
    const int64_t tok = tok0 + wave * 32 + r;
    const int part = unit / a.ntt;
    const int nvt = (a.V + 127) >> 7;
    const int vt0 = (int)((int64_t)nvt * part / a.parts), vt1 = (int)((int64_t)nvt * (part + 1) / a.parts);

    float lse2 = INFINITY;                              // rows past R: p = 2^(-inf) = 0
    float pmax = 0.f;
    if (tok < a.R) {
        const f32x2 rs = *reinterpret_cast<const f32x2 *>(a.rowstat + tok * 2);
        lse2 = rs[0];
        pmax = __builtin_amdgcn_exp2f(rs[1] * VCE_LOG2E - lse2);
    }
    // (block-uniform) some row here has a dominant entry, the one probability that may exceed 1 - 1e-7: counted apart, see
    // vce_token_kernel<KD, 2>
    const bool anyhi = __syncthreads_or(pmax > 0.4f);

    bf16x8 hfr[NKS];
    vce_load_hfrag<KD>(a.h, a.ld_h, tok, a.R, hf, hfr);

    // ---- W tiles: a ring of three LDS slots.  Tile t is read while the chain of tile t + 1's first rows already runs and
    // tile t + 2 is on its way (LDS-DMA).  Slot bases are run-time values: every per-lane LDS address is a VGPR (fragment
    // offset + slot base) + an immediate (the 32-row tile inside the W tile) ----
    int aLc[NKS], aLn[NKS];          // direct fragments: this W tile / the next one
    int aP[NDT][2];                  // transposed fragments: this W tile
    int aBc, aBn;                    // bias quads
    // (absolute LDS byte addresses: formed from `smem + offset` the compiler re-adds the array's base in front of every read)
    const int lds0i = (int)(size_t)(__attribute__((address_space(3))) char *)smem;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) { aLc[ks] = lds0i + VTile<KD>::frag_off(r, ks, hf); aLn[ks] = aLc[ks] + TILE_B; }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
        aP[dt][0] = lds0i + VTile<KD>::tr_off(hf, li, g, dt, 0);
        aP[dt][1] = lds0i + VTile<KD>::tr_off(hf, li, g, dt, 1);
    }
    aBc = lds0i + BIAS0 + 16 * hf;
    aBn = aBc + 512;
    int slot_c = 0;                  // ring slot of the W tile whose probabilities are being formed

    // the DMA of W tile vt into ring slot `slot`, one piece (1 KiB per wave instruction) at a time, and the tile's bias
    const unsigned lds0 = (unsigned)lds0i;
    float breg = 0.f;
    auto dma_piece = [&](int vt, int slot, int i) __attribute__((always_inline)) {
        VTile<KD>::template dma_piece_asm<256>(a.wt, a.ld_w, (int64_t)vt * 128, vt < vt1 ? a.V : 0, lds0 + (unsigned)(slot * TILE_B), wave, lane, i);
    };
    auto bias_fetch = [&](int vt) __attribute__((always_inline)) {
        if (tid < 128) {
            const int v = vt * 128 + tid;
            breg = (vt < vt1 && v < a.V) ? (a.bias ? a.bias[v] : 0.f) : -INFINITY;   // rows past V: logit = -inf, p = 0
        }
    };
    auto bias_store = [&](int slot) __attribute__((always_inline)) {
        if (tid < 128) *reinterpret_cast<float *>(smem + BIAS0 + slot * 512 + tid * 4) = breg;
    };
    // tiles vt0 (slot 0) and vt0 + 1 (slot 1) before anything else
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma_piece(vt0, 0, i);
    bias_fetch(vt0);
    bias_store(0);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma_piece(vt0 + 1, 1, i);
    bias_fetch(vt0 + 1);
    bias_store(1);
    VCE_DMA_WAIT();
    __syncthreads();

#ifdef VCE_SCAN_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#define XSTAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[k] += t_ - t0_; t0_ = t_; } while (0)
#else
#define XSTAMP(k) do { } while (0)
#endif
    float Pc = 0.f;                 // sum of the probabilities below the clip range
    unsigned nlow = 0, nhi = 0;     // entries below the range (rows past V among them: p = 0) / dominant entries (p > 1/2)
    f32x16 U[NDT], Ud[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int t = 0; t < 16; ++t) { U[dt][t] = 0.f; Ud[dt][t] = 0.f; }

    // The pipeline: global step s (one 32-row tile of the vocabulary) does
    //   VALU : probabilities of tile s (logits in acc[s & 3]) -> pk / pl[s & 1], Pc, counts
    //   MFMA : logits chain of tile s + 1 into acc[(s + 1) & 3];  U / Ud += W^T P of tile s - 1 (pk / pl[(s - 1) & 1])
    //   LDS  : every fragment register is re-loaded right behind the MFMA that consumed it with what the NEXT step's MFMA of
    //          that slot needs: the direct fragments (and the bias, into acc[(s + 2) & 3]) of tile s + 2, the transposed
    //          fragments of tile s.  One set of fragment registers, a full step of latency cover; four logits accumulators.
    // A W tile is four steps (RT = 0 .. 3): the re-loads of RT = 2, 3 reach into the NEXT W tile (aLn / aBn).
    // The accumulators of the logits chain must be ordinary VGPRs: the VALU reads every entry, and with more than 256
    // registers in use the compiler keeps MFMA results in the accumulation half of the file (one v_accvgpr_read per entry).
    // The chain's MFMA is therefore inline assembly with VGPR operands; its results are first read a whole step later
    // (>= 16 MFMAs), so no hazard wait is due -- except after the fill step, which waits explicitly.  The two accumulations
    // (Pc, counts) are inline assembly as well: left to the compiler they sink to the end of the W tile (64 masks and values
    // kept alive: spills), out of the MFMA gaps they are meant to fill.
    f32x16 acc[4];
    bf16x8 Lf[NKS], Pf[NPF], pk[2][2], pl[2][2];
#pragma unroll
    for (int i = 0; i < NPF; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) Pf[i][j] = (bf16_t)0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) { pk[i][k][j] = (bf16_t)0.f; pl[i][k][j] = (bf16_t)0.f; }
    const float l2e = VCE_LOG2E;
    typedef const __attribute__((address_space(3))) bf16x8 *lds_bf8_t;
    typedef const __attribute__((address_space(3))) f32x4 *lds_f4_t;
    typedef vs16x4 __attribute__((address_space(3))) *lds_tr_t;
    auto ldsL = [&](int addr, int imm) __attribute__((always_inline)) { return *(lds_bf8_t)(size_t)(unsigned)(addr + imm); };
    auto ldsT = [&](int a0, int a1, int imm) __attribute__((always_inline)) {
        const vs16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(size_t)(unsigned)(a0 + imm));
        const vs16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(size_t)(unsigned)(a1 + imm));
        const vs16x8 w = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
        return __builtin_bit_cast(bf16x8, w);
    };
    auto ldsB = [&](int addr, int rt, f32x16 &acc_) __attribute__((always_inline)) {
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const f32x4 q = *(lds_f4_t)(size_t)(unsigned)(addr + (rt * 32 + 8 * tq) * 4);
            acc_[4 * tq] = q[0]; acc_[4 * tq + 1] = q[1]; acc_[4 * tq + 2] = q[2]; acc_[4 * tq + 3] = q[3];
        }
    };
    // DOE: probabilities of this step's tile; DOC: chain of the next; DOP: P W of the previous; LDP: re-load the transposed
    // fragments; DMA: the W tile after the next one is requested piece by piece in this step's gaps (vt_dma -> slot_dma)
    auto step = [&](auto RTT, auto DOE, auto DOC, auto DOP, auto LDP, auto DMA, auto HI, int vt_dma, int slot_dma) __attribute__((always_inline)) {
        constexpr int RT = decltype(RTT)::value;
        constexpr bool do_e = decltype(DOE)::value, do_c = decltype(DOC)::value, do_p = decltype(DOP)::value;
        constexpr bool ld_p = decltype(LDP)::value, dma = decltype(DMA)::value, hi = decltype(HI)::value;
        constexpr int cur = RT & 1, prv = (RT + 1) & 1;
        constexpr int ac = RT & 3, an = (RT + 1) & 3, an2 = (RT + 2) & 3;
        constexpr int rt2 = (RT + 2) & 3;                      // the 32-row tile the direct re-loads fetch ...
        constexpr bool nextw = RT >= 2;                          // ... of the next W tile
        constexpr int n_c = do_c ? NKS : 0, n_p = do_p ? 4 * NDT : 0, n_mfma = n_c + n_p;
        constexpr int n_gap = n_mfma > 0 ? n_mfma : 1;
        float pv0 = 0.f, pv1 = 0.f, pl0 = 0.f, pl1 = 0.f;
        auto entry = [&](int t, float &pv, float &plo) __attribute__((always_inline)) {
            pv = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[ac][t], l2e, -lse2));
            const bool low = pv < VCE_EPS;
            plo = low ? pv : 0.f;
            asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(Pc) : "v"(plo));
            asm volatile("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(nlow) : "s"(__builtin_amdgcn_ballot_w64(low)) : "vcc");
            // (tried: compare, select and carry-in through VCC in one assembly block -- v_cmp_e32 / s_nop 1 / v_cndmask_e32 /
            // v_addc_co_e32 -- 1.5 % faster and WRONG on the lanes with bit 2 clear; not understood, not kept)
            unsigned &nhi_ = nhi;          // (named outside the discarded branch: the capture must not depend on `hi`)
            if constexpr (hi) asm volatile("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(nhi_) : "s"(__builtin_amdgcn_ballot_w64(pv > 0.5f)) : "vcc");
        };
        // the 24 chunks of a tile's VALU work: entry 2 j | entry 2 j + 1 | their two packs
        auto chunk = [&](int c) __attribute__((always_inline)) {
            const int j = c / 3, ph = c % 3, t0 = 2 * j, t1 = 2 * j + 1;
            if (ph == 0) entry(t0, pv0, pl0);
            else if (ph == 1) entry(t1, pv1, pl1);
            else {
                // entries t0, t1: elements (t & 7) of the k-step (t >> 3) of the B operand
                pk[cur][t0 >> 3][t0 & 7] = (bf16_t)pv0; pk[cur][t0 >> 3][t1 & 7] = (bf16_t)pv1;
                pl[cur][t0 >> 3][t0 & 7] = (bf16_t)pl0; pl[cur][t0 >> 3][t1 & 7] = (bf16_t)pl1;
            }
        };
        if (do_c) ldsB(nextw ? aBn : aBc, rt2, acc[an2]);
#pragma unroll
        for (int i = 0; i < n_gap; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if (i < n_c) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[an]) : "v"(Lf[i]), "v"(hfr[i]));
                Lf[i] = ldsL(nextw ? aLn[i] : aLc[i], rt2 * 32 * STR);
            } else if (i < n_mfma) {
                const int q = i - n_c, f = q / 2;                 // fragment f = s2 * NDT + dt serves U then Ud
                const int s2 = f / NDT, dt = f % NDT;
                if ((q & 1) == 0) U[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Pf[f], pk[prv][s2], U[dt], 0, 0, 0);
                else {
                    Ud[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Pf[f], pl[prv][s2], Ud[dt], 0, 0, 0);
                    if (ld_p) {
                        const int o = (RT * 32 + 16 * s2) * STR;
                        Pf[f] = ldsT(aP[dt][0], aP[dt][1], o);
                    }
                }
            }
            if (dma) {
#pragma unroll
                for (int d = (i * NDMA) / n_gap; d < ((i + 1) * NDMA) / n_gap; ++d) dma_piece(vt_dma, slot_dma, d);
            }
            if (do_e) {
#pragma unroll
                for (int c = (i * 24) / n_gap; c < ((i + 1) * 24) / n_gap; ++c) chunk(c);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr std::integral_constant<bool, true> Y{};
    constexpr std::integral_constant<bool, false> N{};
    constexpr std::integral_constant<int, 0> R0{};
    constexpr std::integral_constant<int, 1> R1{};
    constexpr std::integral_constant<int, 2> R2{};
    constexpr std::integral_constant<int, 3> R3{};

    auto sweep = [&](auto HI) __attribute__((always_inline)) {
        // fill: fragments and bias of the first 32-row tile, its chain (as "step 3 of the W tile before": the re-loads reach
        // into the NEXT W tile, which is the first one), then a wait: nothing orders the inline-assembly MFMAs' results
        // against the VALU reads that follow at once
        {
            // the first W tile plays "next": aLn / aBn point at slot 0 for the fill step
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) aLn[ks] -= TILE_B;
            aBn -= 512;
            ldsB(aBn, 0, acc[0]);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) Lf[ks] = ldsL(aLn[ks], 0);
            step(R3, N, Y, N, N, N, HI, 0, 0);      // chain(0) into acc[0]; re-loads: L(1), bias(1) -> acc[1]
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) aLn[ks] += TILE_B;
            aBn += 512;
            asm volatile("s_nop 15\n\ts_nop 7");
        }
        for (int vt = vt0; vt < vt1; ++vt) {
            XSTAMP(7);
            step(R0, Y, Y, Y, Y, N, HI, 0, 0);
            XSTAMP(0);
            step(R1, Y, Y, Y, Y, N, HI, 0, 0);
            XSTAMP(1);
            // W tile vt + 1 has landed (requested a tile ago) and every wave is past its last read of tile vt - 1: that slot
            // takes tile vt + 2, requested in the gaps of the next step
            bias_store((slot_c + 1) % 3);      // the bias requested together with tile vt + 1, one W tile ago
            VCE_DMA_WAIT();
            B4C_LDS_BARRIER();
            XSTAMP(6);
            const int slot_n2 = (slot_c + 2) % 3;
            bias_fetch(vt + 2);
            step(R2, Y, Y, Y, Y, Y, HI, vt + 2, slot_n2);
            XSTAMP(2);
            step(R3, Y, Y, Y, Y, N, HI, 0, 0);
            XSTAMP(3);
            // the ring turns: next -> current
            const int d_c = ((slot_c + 1) % 3 - slot_c) * TILE_B, d_n = ((slot_c + 2) % 3 - (slot_c + 1) % 3) * TILE_B;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) { aLc[ks] += d_c; aLn[ks] += d_n; }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) { aP[dt][0] += d_c; aP[dt][1] += d_c; }
            aBc += d_c / TILE_B * 512;
            aBn += d_n / TILE_B * 512;
            slot_c = (slot_c + 1) % 3;
        }
        // drain: P W of the last 32-row tile
        step(R0, N, N, Y, N, N, HI, 0, 0);
        XSTAMP(4);
    };
    if (anyhi) sweep(Y); else sweep(N);
    // the last W tile's request (a tile past this part: zeros) may still be on its way into the ring, and the compiler knows
    // nothing of the inline-assembly DMA: it must have landed before the ring's memory is reused below
    VCE_DMA_WAIT();
#ifdef VCE_SCAN_STAMPS
    if (lane == 0 && blockIdx.x < 2048)
        for (int k = 0; k < 8; ++k) g_vce_xstamps[(blockIdx.x * 4 + wave) * 8 + k] = st_[k];
#endif
    __syncthreads();   // all tiles consumed: LDS is reused below

    // U^T / Ud^T tiles -> LDS [token][d] per wave -> row-major partial sums of this vocabulary part; per-lane scalars summed
    // over the two lanes of a token
    constexpr int USTR = KD + 4;
    float *sU = reinterpret_cast<float *>(smem) + wave * 32 * USTR;
    auto flush = [&](f32x16 (&X)[NDT], float *dst) __attribute__((always_inline)) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const f32x4 v = {X[dt][4 * tq], X[dt][4 * tq + 1], X[dt][4 * tq + 2], X[dt][4 * tq + 3]};
                *reinterpret_cast<f32x4 *>(sU + r * USTR + dt * 32 + 8 * tq + 4 * hf) = v;
            }
        __syncthreads();
        for (int c = tid; c < 128 * (KD / 4); c += 256) {
            const int t = c / (KD / 4), q = c % (KD / 4);
            if (tok0 + t < a.R)
                *reinterpret_cast<f32x4 *>(dst + (tok0 + t) * KD + q * 4) =
                    *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(smem) + t * USTR + q * 4);
        }
        __syncthreads();
    };
    flush(U, a.u + (int64_t)part * a.R * KD);
    flush(Ud, a.ud + (int64_t)part * a.R * KD);
    f32x4 *sS = reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(smem) + 4 * 32 * USTR);   // [wave][lane]
    // (n_low counts the rows past V of the last W tile too -- their p is 0: the combine kernel takes them out again)
    sS[wave * 64 + lane] = (f32x4){(float)nlow, Pc, (float)nhi, 0.f};
    __syncthreads();
    if (tid < 128 && tok0 + tid < a.R) {
        const int tgi = tid >> 5, ri = tid & 31;
        *reinterpret_cast<f32x4 *>(a.sp + ((int64_t)part * a.R + tok0 + tid) * 4) = sS[tgi * 64 + ri] + sS[tgi * 64 + ri + 32];
    }
}

// {lse2, max x} per row from the per-part statistics of the lse sweep (vce_token_kernel<KD, 0>: m2, l, -, max x)
__global__ void __launch_bounds__(256) vce_rowstat_kernel(VceArgs a, float *__restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.R) return;
    float M = -INFINITY, l = 0.f, mx = -INFINITY;
    for (int p = 0; p < a.parts_st; ++p) {
        const f32x4 s = *reinterpret_cast<const f32x4 *>(a.st1 + ((int64_t)p * a.R + row) * 4);
        const float M2 = fmaxf(M, s[0]);
        l = l * __builtin_amdgcn_exp2f(M - M2) + s[1] * __builtin_amdgcn_exp2f(s[0] - M2);
        M = M2;
        mx = fmaxf(mx, s[3]);
    }
    *reinterpret_cast<f32x2 *>(out + row * 2) = (f32x2){M + __log2f(l), mx};
}

// ------------------------------------------------------------------------------------------
// K3: one wave per row.  loss, dh, row scalars for the dW sweep.
//   dL/dx_j = p_j (u_j / S - G) - [j = y] u_y p_y / clip(p_y),  G = Pu / S - u_y p_y / clip(p_y)
//   (u = 1, S = 1, G = 0 on rows that never leave the clip range; plain variant: p_j - [j = y])
// ------------------------------------------------------------------------------------------
template <int KD>
__global__ void __launch_bounds__(256) vce_combine_kernel(VceArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.R) return;
    constexpr int E = KD / 64;      // elements per lane (1 or 2), consecutive
    const int y = a.labels[row];
    const bool valid = y >= 0 && y < a.V;
    float *rs = a.rowscal + row * 8;
    bf16_t *dh = a.dh + row * a.ld_dh;
    if (!valid) {
#pragma unroll
        for (int e = 0; e < E; ++e) dh[lane * E + e] = (bf16_t)0.f;
        if (lane == 0) {
            // the NaN goes out as an integer pattern (it survives any floating-point option the file is compiled with)
            reinterpret_cast<uint32_t *>(a.item_loss)[row] = (y >= a.V) ? 0x7fc00000u : 0u;
            *reinterpret_cast<f32x4 *>(rs) = (f32x4){INFINITY, 0.f, 0.f, -INFINITY};     // lse2 = +inf: p = 0
            *reinterpret_cast<f32x4 *>(rs + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    float lse2, pmax;
    bool clipped;
    vce_row_stats(a, row, lse2, clipped, pmax);
    const float lse = lse2 * VCE_LN2;
    float U[E], Ud[E], hv[E], wy[E];
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int d = lane * E + e;
        U[e] = 0.f;
        Ud[e] = 0.f;
        for (int p = 0; p < a.parts; ++p) {
            // part p's sums are relative to its own reference m2_p: 2^(m2_p - lse2) makes them probabilities (the lse-first
            // sweep accumulated probabilities to begin with)
            const float f = a.rowstat ? 1.f : __builtin_amdgcn_exp2f(a.st1[((int64_t)p * a.R + row) * 4] - lse2);
            U[e] += f * a.u[((int64_t)p * a.R + row) * KD + d];
            if (clipped) Ud[e] += a.ud[((int64_t)p * a.R + row) * KD + d];     // the second sweep ran against lse2 itself
        }
        hv[e] = (float)a.h[row * a.ld_h + d];
        wy[e] = (float)a.wt[(int64_t)y * a.ld_w + d];
        dot += hv[e] * wy[e];
    }
    dot = wave_sum(dot);
    const float xy = dot + (a.bias ? a.bias[y] : 0.f);
    const float py = __expf(xy - lse);
    const float gs = a.grad_scale[0];
    float loss, invS = 1.f, G = 0.f, yd = 1.f;
    bool all_out = false;
    if (clipped) {
        float nu = 0.f, Pc = 0.f, ntop = 0.f;
        for (int p = 0; p < a.parts; ++p) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(a.sp + ((int64_t)p * a.R + row) * 4);
            nu += q[0]; Pc += q[1]; ntop += q[2];
        }
        // (the lse-first sweep counts the entries BELOW the range, the padding rows of the last W tile among them)
        if (a.rowstat) nu = (float)a.V - (nu - (float)(((a.V + 127) >> 7) * 128 - a.V)) - ntop;
        // Does the dominant entry (p > 1/2, if the row has one) exceed 1 - 1e-7?  Exactly when all the OTHER entries together
        // stay below 1e-7: none of them inside the clip range (each of those alone is >= 1e-7) and the mass Pc of the ones
        // below it -- a sum of tiny numbers, exact to fp32 rounding -- under 1e-7.  (1 - p itself is not representable
        // there, and 2^(x log2e - lse2) is off by 1e-6 at logits of +-50: a decision read off p flipped between the kernels.)
        all_out = ntop > 0.5f && nu < 0.5f && Pc < VCE_EPS;
        // (lse-first form: every row went through the clipped sweep; one with every probability inside the range -- all V
        // entries counted -- is an unclipped row and takes the unclipped formulas, to the bit)
        if (a.rowstat && nu + ntop >= (float)a.V) { clipped = false; all_out = false; }
        if (!clipped) {
            loss = lse - xy;
        } else if (all_out) {
            // every probability is outside the clip range: the loss is a constant of the row, its gradient exactly zero
            const float S = (1.0f - VCE_EPS) + VCE_EPS * ((float)a.V - 1.0f);
            loss = logf(S) - logf(py > 0.5f ? 1.0f - VCE_EPS : VCE_EPS);
            invS = 0.f; yd = 0.f; G = 0.f;
        } else {
            const float Pu = 1.0f - Pc;          // the probabilities are normalised by the row's own lse: they sum to 1
            // S = sum_j clip(p_j): the entries inside the range (the dominant one among them) as they are, every other at 1e-7
            const float S = Pu + VCE_EPS * ((float)a.V - nu - ntop);
            const float pyc = fmaxf(py, VCE_EPS);
            invS = 1.0f / S;
            yd = (py >= VCE_EPS) ? 1.f : 0.f;    // u_y p_y / clip(p_y)
            G = Pu * invS - yd;
            loss = logf(S) - logf(pyc);
        }
    } else {
        loss = lse - xy;          // -log p_y
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const float v = all_out ? 0.f : (U[e] - Ud[e]) * invS - G * U[e] - yd * wy[e];
        dh[lane * E + e] = (bf16_t)(v * gs);
    }
    if (lane == 0) {
        a.item_loss[row] = loss;
        // dlogit_j = p_j (u_j a - b): inside the clip range [lo, hi] that is p_j c, outside it p_j nb
        const float ra = gs * invS, rb = gs * G;
        *reinterpret_cast<f32x4 *>(rs) = (f32x4){lse2, ra - rb, -rb, clipped ? VCE_EPS : -INFINITY};
        *reinterpret_cast<f32x4 *>(rs + 4) = (f32x4){gs * yd, all_out ? 1.f : 0.f, 0.f, 0.f};
    }
}

// ------------------------------------------------------------------------------------------
// K4: one workgroup = 128 vocabulary rows x one split of the token tiles; 8 waves = 4 vocabulary groups
// (32 ids on the lanes) x 2 token halves of each 128-token h tile.
//   x[tok][v] = h W^T + b;  dlogit = p (u a - b');  dW^T[d][v] += h^T dlogit;  db[v] += colsum
// ------------------------------------------------------------------------------------------
struct VceDwArgs {
    const bf16_t *h;
    const bf16_t *wt;
    const float *bias;
    const float *rowscal;
    float *dW;           // [KD][ldw] fp32 (Keras kernel layout [in][out])
    float *db;           // [V] or NULL
    int ld_h, ld_w, ldw;
    int64_t R;
    int V, tsplit;
    int vt0, nvt;        // the vocabulary tiles [vt0, vt0 + nvt) this launch sweeps
};

template <int KD, int TH>
__global__ void __launch_bounds__(256 * TH, 2) vce_dw_kernel(VceDwArgs a) {
    // TH = 2: 8 waves = 4 vocabulary groups x the 2 token halves of each 128-token h tile, one (vocabulary tile, token
    // split) unit per workgroup.  TH = 1: 4 waves, one per SIMD, each taking all four 32-token sub-tiles; the grid is at
    // most one workgroup per CU and walks the units -- half of every SIMD's registers and issue slots stay free for the
    // HBM-bound kernels of the encoder backward that run beside it on the main stream (ops.overlap_vocab_dw).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NKS = KD / 16, NDT = KD / 32, STR = VTile<KD>::STR;
    constexpr int TILE_B = VTile<KD>::BYTES;
    constexpr int NT = 256 * TH, RTN = 4 / TH;                        // threads; 32-token sub-tiles per wave and h tile
    f32x4 *sRow = reinterpret_cast<f32x4 *>(smem + 2 * TILE_B);      // [2][128]
    const int nvt = a.nvt;
    const int nunits = nvt * a.tsplit;
    int unit = blockIdx.x;
    do {          // TH = 2: one unit per workgroup, no loop
    // (lane coordinates re-derived per unit behind an opaque zero: hoisted, the per-lane LDS offsets of every access
    // pattern stay live across the loop and the kernel spills)
    int opaque0 = 0;
    if (TH == 1) asm volatile("s_mov_b32 %0, 0" : "=s"(opaque0));
    const int tid = threadIdx.x + opaque0, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int li = lane & 15, g = lane >> 4;
    const int vg = wave & 3, th = wave >> 2;
    const int vt = a.vt0 + unit % nvt, ts = unit / nvt;
    const int v = vt * 128 + vg * 32 + r;          // the lane's vocabulary id
    const int64_t ntt = (a.R + 127) >> 7;
    const int64_t tt0 = ntt * ts / a.tsplit, tt1 = ntt * (ts + 1) / a.tsplit;

    bf16x8 wfr[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        vu32x4 q = {0u, 0u, 0u, 0u};
        if (v < a.V) q = *reinterpret_cast<const vu32x4 *>(a.wt + (int64_t)v * a.ld_w + ks * 16 + hf * 8);
        wfr[ks] = __builtin_bit_cast(bf16x8, q);
    }
    const float bv = (v < a.V) ? (a.bias ? a.bias[v] : 0.f) : -INFINITY;
    int foff[NKS], toff[NDT][2];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) foff[ks] = VTile<KD>::frag_off(r, ks, hf) + th * (RTN * 32) * STR;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
        toff[dt][0] = VTile<KD>::tr_off(hf, li, g, dt, 0) + th * (RTN * 32) * STR;
        toff[dt][1] = VTile<KD>::tr_off(hf, li, g, dt, 1) + th * (RTN * 32) * STR;
    }
    const int roff = (th * (RTN * 32) + 4 * hf) * 16;       // the lane's first row-scalar entry (bytes)

    f32x4 rreg = {INFINITY, 0.f, 0.f, -INFINITY};
    auto fetch = [&](int64_t tt, int buf) {
        VTile<KD>::template dma<NT>(a.h, a.ld_h, tt * 128, tt < tt1 ? a.R : 0, smem + buf * TILE_B, tid);
        if (tid < 128) {
            const int64_t row = tt * 128 + tid;
            rreg = (tt < tt1 && row < a.R) ? *reinterpret_cast<const f32x4 *>(a.rowscal + row * 8) : (f32x4){INFINITY, 0.f, 0.f, -INFINITY};
        }
    };
    fetch(tt0, 0);
    if (tid < 128) sRow[tid] = rreg;
    VCE_DMA_WAIT();
    __syncthreads();

    f32x16 dW[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int t = 0; t < 16; ++t) dW[dt][t] = 0.f;
    float dbv = 0.f;

    auto tile = [&](auto BUF, int64_t tt) {
        constexpr int buf = decltype(BUF)::value;
        const char *hh = smem + buf * TILE_B;
        const f32x4 *rs = sRow + buf * 128;
        const char *rsl = reinterpret_cast<const char *>(rs) + roff;
        fetch(tt + 1, buf ^ 1);        // the other buffer was last read one tile ago (behind the previous barrier)
        // row scalars {lse2, c = a - b, nb = -b, lo}: lo > 0 marks a row whose probabilities leave the clip range (rows that
        // stay inside carry lo = -inf)
        const bool any_clip = TH == 2 ? __any(rs[th * 64 + lane][3] > 0.f)      // the wave's token rows
                                      : __any(rs[lane][3] > 0.f || rs[64 + lane][3] > 0.f);
#pragma unroll
        for (int rt = 0; rt < RTN; ++rt) {
            __builtin_amdgcn_sched_barrier(0);       // keep one 32-token tile's temporaries live at a time
            f32x16 acc;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = bv;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 hfg = *reinterpret_cast<const bf16x8 *>(hh + rt * 32 * STR + foff[ks]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hfg, wfr[ks], acc, 0, 0, 0);
            }
            float gv[16];
            if (!any_clip) {      // wave-uniform
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const f32x2 s = *reinterpret_cast<const f32x2 *>(rsl + (rt * 32 + (t & 3) + 8 * (t >> 2)) * 16);   // broadcast
                    gv[t] = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[t], VCE_LOG2E, -s[0])) * s[1];
                    dbv += gv[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const f32x4 s = *reinterpret_cast<const f32x4 *>(rsl + (rt * 32 + (t & 3) + 8 * (t >> 2)) * 16);
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[t], VCE_LOG2E, -s[0]));
                    // below lo = outside the clip range (lo = -inf: never); the one entry that can be ABOVE the range is the row's
                    // dominant one, and a row where it is has c = nb = 0 (vce_combine_kernel): nothing to tell apart here
                    const bool un = p >= s[3];
                    gv[t] = p * (un ? s[1] : s[2]);
                    dbv += gv[t];
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 gf = vce_pack8(gv + 8 * s2);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    // h^T[d = dt*32 + r][tokens 16 s2 + 4 hf + {0..3, 8..11} of this 32-token tile]
                    const char *hb = hh + (rt * 32 + 16 * s2) * STR;
                    const bf16x8 htf = vce_frag_tr2(hb + toff[dt][0], hb + toff[dt][1]);
                    dW[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(htf, gf, dW[dt], 0, 0, 0);
                }
            }
        }
        if (tid < 128) sRow[(buf ^ 1) * 128 + tid] = rreg;
        VCE_DMA_WAIT();
        B4C_LDS_BARRIER();
    };
    for (int64_t tt = tt0; tt < tt1; tt += 2) {
        tile(std::integral_constant<int, 0>{}, tt);
        if (tt + 1 < tt1) tile(std::integral_constant<int, 1>{}, tt + 1);
    }
    __syncthreads();
    dbv += __shfl_xor(dbv, 32);
    const bool direct = a.tsplit == 1;
    if (TH == 2) {
        // the two token halves are summed through LDS; [d][32 vocab] per vocabulary group
        float *sD = reinterpret_cast<float *>(smem) + vg * (KD * 32 + 32);
        if (th == 1) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int t = 0; t < 16; ++t) sD[(dt * 32 + vce_rowmap(t, hf)) * 32 + r] = dW[dt][t];
            if (hf == 0) sD[KD * 32 + r] = dbv;
        }
        __syncthreads();
        if (th == 0 && v < a.V) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int d = dt * 32 + vce_rowmap(t, hf);
                    const float val = dW[dt][t] + sD[d * 32 + r];
                    float *p = a.dW + (int64_t)d * a.ldw + v;
                    if (direct) *p += val;
                    else atomicAdd(p, val);
                }
            if (hf == 0 && a.db) {
                const float val = dbv + sD[KD * 32 + r];
                if (direct) a.db[v] += val;
                else atomicAdd(a.db + v, val);
            }
        }
    } else if (v < a.V) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float *p = a.dW + (int64_t)(dt * 32 + vce_rowmap(t, hf)) * a.ldw + v;
                if (direct) *p += dW[dt][t];
                else atomicAdd(p, dW[dt][t]);
            }
        if (hf == 0 && a.db) {
            if (direct) a.db[v] += dbv;
            else atomicAdd(a.db + v, dbv);
        }
    }
    if (TH == 1) __syncthreads();        // the next unit's tiles land in the images this one read
    unit += gridDim.x;
    } while (TH == 1 && unit < nunits);
}

// K5: the [j = y] term of dlogit: dW[:, y] -= yd h_row, db[y] -= yd.  dW is [K][V] (a label touches a column),
// so the rows are first added into a vocabulary-major scratch [V][KD] (coalesced atomics, one wave per row), which
// is then added transposed.
#define VCE_HOT 64          // labels < VCE_HOT (the head of a frequency-ranked vocabulary) are pre-summed in LDS
#define VCE_LABEL_ROWS 128  // rows per workgroup
template <int KD>
__global__ void __launch_bounds__(256) vce_label_kernel(VceDwArgs a, const int32_t *__restrict__ labels, float *__restrict__ tmp) {
    __shared__ float hot[VCE_HOT][KD + 1];      // [..][KD] = the bias term
    __shared__ int touched[VCE_HOT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < VCE_HOT * (KD + 1); i += 256) (&hot[0][0])[i] = 0.f;
    if (tid < VCE_HOT) touched[tid] = 0;
    __syncthreads();
    const int64_t row0 = (int64_t)blockIdx.x * VCE_LABEL_ROWS;
    for (int i = wave; i < VCE_LABEL_ROWS; i += 4) {
        const int64_t row = row0 + i;
        if (row >= a.R) break;
        const int y = labels[row];
        if (y < 0 || y >= a.V) continue;
        const float yd = a.rowscal[row * 8 + 4];
        if (yd == 0.f) continue;
        if (y < VCE_HOT) {        // wave-uniform
#pragma unroll
            for (int e = 0; e < KD / 64; ++e) {
                const int d = lane + 64 * e;
                atomicAdd(&hot[y][d], -yd * (float)a.h[row * a.ld_h + d]);
            }
            if (lane == 0) { atomicAdd(&hot[y][KD], -yd); touched[y] = 1; }
        } else {
#pragma unroll
            for (int e = 0; e < KD / 64; ++e) {
                const int d = lane + 64 * e;
                atomicAdd(tmp + (int64_t)y * KD + d, -yd * (float)a.h[row * a.ld_h + d]);
            }
            if (lane == 0 && a.db) atomicAdd(a.db + y, -yd);
        }
    }
    __syncthreads();
    for (int i = tid; i < VCE_HOT * KD; i += 256) {
        const int y = i / KD, d = i % KD;
        if (touched[y]) atomicAdd(tmp + (int64_t)y * KD + d, hot[y][d]);
    }
    if (tid < VCE_HOT && touched[tid] && a.db) atomicAdd(a.db + tid, hot[tid][KD]);
}
template <int KD>
__global__ void __launch_bounds__(256) vce_label_add_kernel(float *__restrict__ dW, int ldw, const float *__restrict__ tmp, int V) {
    __shared__ float t[32][33];
    const int v0 = blockIdx.x * 32, d0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = v0 + ty + 8 * i;
        t[ty + 8 * i][tx] = (v < V) ? tmp[(int64_t)v * KD + d0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = d0 + ty + 8 * i, v = v0 + tx;
        const float x = t[tx][ty + 8 * i];
        if (v < V && x != 0.f) dW[(int64_t)d * ldw + v] += x;
    }
}

// K5, deterministic form: the rows are sorted by label (stable: b4c_sort_ids), ONE wave sums each run of equal labels in row
// order and is the only writer of that label's row of the scratch -- no float atomics, the same bits every time.
__global__ void __launch_bounds__(256) vce_label_keys_kernel(const int32_t *__restrict__ labels, int64_t R, int V, int64_t *__restrict__ keys) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i < R) { const int y = labels[i]; keys[i] = (y >= 0 && y < V) ? y : V; }          // V: ignored rows sort to the end
}
template <int KD>
__global__ void __launch_bounds__(256) vce_label_sorted_kernel(VceDwArgs a, const int64_t *__restrict__ keys, const int32_t *__restrict__ order,
                                                               float *__restrict__ tmp) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.R) return;
    const int64_t y = keys[order[i]];
    if (y >= a.V || (i > 0 && keys[order[i - 1]] == y)) return;           // not the first row of a run of a valid label
    float acc[KD / 64], sb = 0.f;
#pragma unroll
    for (int e = 0; e < KD / 64; ++e) acc[e] = 0.f;
    for (int64_t j = i; j < a.R; ++j) {
        const int64_t row = order[j];
        if (keys[row] != y) break;
        const float yd = a.rowscal[row * 8 + 4];
#pragma unroll
        for (int e = 0; e < KD / 64; ++e) acc[e] += -yd * (float)a.h[row * a.ld_h + lane + 64 * e];
        sb += -yd;
    }
#pragma unroll
    for (int e = 0; e < KD / 64; ++e) tmp[y * KD + lane + 64 * e] = acc[e];
    if (lane == 0 && a.db) a.db[y] += sb;
}

// lse2[row] = log2 sum_j 2^(x_j log2 e) from the per-part statistics of vce_token_kernel<KD, 0>
__global__ void __launch_bounds__(256) vce_lse_kernel(VceArgs a, float *__restrict__ lse2) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.R) return;
    float M = -INFINITY, l = 0.f;
    for (int p = 0; p < a.parts; ++p) {
        const f32x4 s = *reinterpret_cast<const f32x4 *>(a.st1 + ((int64_t)p * a.R + row) * 4);
        const float M2 = fmaxf(M, s[0]);
        l = l * __builtin_amdgcn_exp2f(M - M2) + s[1] * __builtin_amdgcn_exp2f(s[0] - M2);
        M = M2;
    }
    lse2[row] = M + __log2f(l);
}

// ------------------------------------------------------------------------------------------
// Split `units` workgroup-sized pieces of work into units * p workgroups for 256 CUs (one workgroup per CU at a time):
// time ~ rounds(p) / p, plus a cost per extra split (partial results to combine: `penalty`, in units of one
// unsplit workgroup's run time).  C2: 320 token tiles -> 4 parts (1280 workgroups, 5 full rounds);
// 391 vocabulary tiles -> 5 token splits (8 rounds instead of 13 for 8 splits, and 5/8 of the dW atomics).
static int vce_pick_split(int64_t units, int64_t max_split, double penalty) {
    int best = 1;
    double best_t = 1e30;
    for (int p = 1; p <= 8; ++p) {
        if (p > max_split) break;
        const double t = (double)ceil_div64(units * p, 256) / p + penalty * p;
        if (t < best_t - 1e-9) { best_t = t; best = p; }
    }
    return best;
}
static bool vce_shape_ok(int K) { return K == 64 || K == 128; }

template <typename Kern> static void vce_allow_lds(Kern k, size_t bytes) {
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <int KD> static size_t vce_token_lds() {
    const size_t tiles = 2 * (size_t)VTile<KD>::BYTES + 2 * 128 * 4;
    const size_t outs = (size_t)8 * 32 * (KD + 4) * 4 + 8 * 64 * 16;
    return tiles > outs ? tiles : outs;
}
// tokens per workgroup of the token-owned sweeps: 256 once there are enough token tiles to fill the chip
// (B4C_VCE_TOKENS=128|256 overrides: A/B)
static int vce_token_nh(int64_t R) {
    static const char *e = getenv("B4C_VCE_TOKENS");
    if (e) return atoi(e) >= 256 ? 2 : 1;
    return R >= 256 * 64 ? 2 : 1;
}
template <int KD> static size_t vce_dw_lds() {
    const size_t tiles = 2 * (size_t)VTile<KD>::BYTES + 2 * 128 * 32;
    const size_t outs = (size_t)4 * (KD * 32 + 32) * 4;
    return tiles > outs ? tiles : outs;
}

extern "C" int64_t b4c_vocab_ce_workspace_bytes(int64_t R, int V, int K) {
    if (R <= 0 || V <= 0 || !vce_shape_ok(K)) return 0;
    // 8 = the largest vocabulary split; + the lse-first form's own statistics (8 parts x R x 4, and 2 per row merged)
    const int64_t fwd = (int64_t)8 * R * (4 + 2 * (int64_t)K + 4) * 4 + (int64_t)R * (8 * 4 + 2) * 4 + 64;
    // vocabulary-major scratch of the label term; deterministic form: + sort keys, order and the sort's own workspace
    const int64_t dw = (int64_t)V * K * 4 + 64 + R * 12 + b4c_sort_ids_workspace_bytes(R, V + 1);
    return fwd > dw ? fwd : dw;
}

// The "lse first" form (round 4): lse sweep -> per-row {lse2, max x} -> ONE exact sweep at one wave per SIMD -> combine.
// B4C_VCE_FORM=online|exact selects (A/B on one library); TF's clip semantics only (the plain variant needs one sweep as it is).
static bool vce_exact_form() {
    static const char *e = getenv("B4C_VCE_FORM");
    return e ? (e[0] == 'e') : false;
}

template <int KD>
static int vce_fwd_exact_launch(VceArgs a, hipStream_t st) {
    // workspace: [exact sweep: sp | u | ud (its own `parts`)] [lse sweep: st1 (parts_st)] [rowstat]
    const int nh = vce_token_nh(a.R);
    const int64_t ntt_l = ceil_div64(a.R, 128 * nh), ntt = ceil_div64(a.R, 128);
    const int nvt = (a.V + 127) / 128;
    float *ws = a.st1;
    VceArgs l = a;                                     // the lse sweep (MODE 0 of the token kernel)
    l.parts = vce_pick_split(ntt_l, nvt, 0.005);
    l.ntt = (int)ntt_l;
    a.parts = vce_pick_split(ntt, nvt, 0.01);
    a.ntt = (int)ntt;
    a.sp = ws;
    a.u = a.sp + (int64_t)a.parts * a.R * 4;
    a.ud = a.u + (int64_t)a.parts * a.R * KD;
    l.st1 = a.ud + (int64_t)a.parts * a.R * KD;
    float *rowstat = l.st1 + (int64_t)l.parts * a.R * 4;
    a.st1 = l.st1;
    a.parts_st = l.parts_st = l.parts;
    a.rowstat = rowstat;
    const size_t lds_l = vce_token_lds<KD>(), lds_x = vce_exact_lds<KD>();
    static thread_local bool done = false;
    if (!done) {
        vce_allow_lds(vce_token_kernel<KD, 0, 1>, lds_l); vce_allow_lds(vce_token_kernel<KD, 0, 2>, lds_l);
        vce_allow_lds(vce_exact_kernel<KD>, lds_x);
        done = true;
    }
    static const bool dbg = getenv("B4C_VCE_TIMING") != nullptr;
    static hipEvent_t ev[4];
    static bool have = false;
    if (dbg) {
        if (have && hipEventQuery(ev[3]) == hipSuccess) {
            float t1 = 0, t2 = 0, t3 = 0;
            (void)hipEventElapsedTime(&t1, ev[0], ev[1]); (void)hipEventElapsedTime(&t2, ev[1], ev[2]); (void)hipEventElapsedTime(&t3, ev[2], ev[3]);
            fprintf(stderr, "[vce_fwd exact] previous call: lse sweep %.3f ms, exact sweep %.3f ms, combine %.3f ms (parts %d / %d)\n", t1, t2, t3, l.parts, a.parts);
        }
        if (!have) { for (auto &e : ev) (void)hipEventCreate(&e); have = true; }
        (void)hipEventRecord(ev[0], st);
    }
    if (nh == 2) vce_token_kernel<KD, 0, 2><<<(unsigned)(ntt_l * l.parts), 512, lds_l, st>>>(l);
    else vce_token_kernel<KD, 0, 1><<<(unsigned)(ntt_l * l.parts), 512, lds_l, st>>>(l);
    vce_rowstat_kernel<<<(unsigned)ceil_div64(a.R, 256), 256, 0, st>>>(l, rowstat);
    if (dbg) (void)hipEventRecord(ev[1], st);
    vce_exact_kernel<KD><<<(unsigned)(ntt * a.parts), 256, lds_x, st>>>(a);
    if (dbg) (void)hipEventRecord(ev[2], st);
    vce_combine_kernel<KD><<<(unsigned)ceil_div64(a.R, 4), 256, 0, st>>>(a);
    if (dbg) (void)hipEventRecord(ev[3], st);
    return b4c_check_launch("vocab_ce_fwd (lse first)");
}

template <int KD>
static int vce_fwd_launch(VceArgs a, hipStream_t st) {
    if (a.variant == B4C_CE_TF && vce_exact_form()) return vce_fwd_exact_launch<KD>(a, st);
    const int nh = vce_token_nh(a.R);
    const int64_t ntt = ceil_div64(a.R, 128 * nh);
    const int nvt = (a.V + 127) / 128;
    // every vocabulary part costs its partial sums a round trip through memory (R x (2 K + 8) floats written, read by the
    // combine: ~20 us per part at C2 against ~1.1 ms for one workgroup's walk over the whole vocabulary)
    a.parts = vce_pick_split(ntt, nvt, nh == 2 ? 0.018 : 0.005);
    float *ws = a.st1;
    a.u = ws + (int64_t)a.parts * a.R * 4;
    a.ud = a.u + (int64_t)a.parts * a.R * KD;
    a.sp = a.ud + (int64_t)a.parts * a.R * KD;
    const size_t lds = vce_token_lds<KD>();
    static thread_local bool done = false;
    if (!done) {
        vce_allow_lds(vce_token_kernel<KD, 1, 1>, lds); vce_allow_lds(vce_token_kernel<KD, 2, 1>, lds);
        vce_allow_lds(vce_token_kernel<KD, 1, 2>, lds); vce_allow_lds(vce_token_kernel<KD, 2, 2>, lds);
        done = true;
    }
    a.ntt = (int)ntt;
    const unsigned grid = (unsigned)(ntt * a.parts);
    // B4C_VCE_TIMING=1: HIP events around the kernels, read back (without synchronising) at the next call
    static const bool dbg = getenv("B4C_VCE_TIMING") != nullptr;
    static hipEvent_t ev[4];
    static bool have = false;
    if (dbg) {
        if (have && hipEventQuery(ev[3]) == hipSuccess) {
            float t1 = 0, t2 = 0, t3 = 0;
            (void)hipEventElapsedTime(&t1, ev[0], ev[1]); (void)hipEventElapsedTime(&t2, ev[1], ev[2]); (void)hipEventElapsedTime(&t3, ev[2], ev[3]);
            fprintf(stderr, "[vce_fwd] previous call: sweep %.3f ms, clipped sweep %.3f ms, combine %.3f ms (parts %d)\n", t1, t2, t3, a.parts);
        }
        if (!have) { for (auto &e : ev) (void)hipEventCreate(&e); have = true; }
        (void)hipEventRecord(ev[0], st);
    }
    if (nh == 2) vce_token_kernel<KD, 1, 2><<<grid, 512, lds, st>>>(a); else vce_token_kernel<KD, 1, 1><<<grid, 512, lds, st>>>(a);
    if (dbg) (void)hipEventRecord(ev[1], st);
    if (a.variant == B4C_CE_TF) {
        if (nh == 2) vce_token_kernel<KD, 2, 2><<<grid, 512, lds, st>>>(a); else vce_token_kernel<KD, 2, 1><<<grid, 512, lds, st>>>(a);
    }
    if (dbg) (void)hipEventRecord(ev[2], st);
    vce_combine_kernel<KD><<<(unsigned)ceil_div64(a.R, 4), 256, 0, st>>>(a);
    if (dbg) (void)hipEventRecord(ev[3], st);
    return b4c_check_launch("vocab_ce_fwd");
}

extern "C" int b4c_vocab_ce_fwd(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, const int32_t *labels,
                                const float *grad_scale, float *item_loss, void *dh, int ld_dh, float *rowscal,
                                void *workspace, int64_t workspace_bytes, int64_t R, int V, int K, int variant, void *stream) {
    B4C_REQUIRE(h && wt && labels && grad_scale && item_loss && dh && rowscal && workspace, "vocab_ce_fwd: null pointer");
    B4C_REQUIRE(vce_shape_ok(K), "vocab_ce_fwd: K=%d unsupported (64 or 128)", K);
    B4C_REQUIRE(variant == B4C_CE_TF || variant == B4C_CE_PLAIN, "vocab_ce_fwd: variant %d", variant);
    B4C_REQUIRE(R >= 0 && V > 0 && ld_h >= K && ld_w >= K && ld_dh >= K, "vocab_ce_fwd: shape");
    B4C_REQUIRE(ld_h % 8 == 0 && ld_w % 8 == 0 && ((((uintptr_t)h | (uintptr_t)wt | (uintptr_t)workspace | (uintptr_t)rowscal) & 15) == 0),
                "vocab_ce_fwd: operands must be 16-byte aligned with pitches % 8 == 0");
    B4C_REQUIRE(workspace_bytes >= b4c_vocab_ce_workspace_bytes(R, V, K), "vocab_ce_fwd: workspace too small");
    if (R == 0) return B4C_OK;
    VceArgs a = {};
    a.h = (const bf16_t *)h; a.wt = (const bf16_t *)wt; a.bias = bias; a.labels = labels; a.grad_scale = grad_scale;
    a.st1 = (float *)workspace; a.rowscal = rowscal; a.item_loss = item_loss; a.dh = (bf16_t *)dh;
    a.ld_h = ld_h; a.ld_w = ld_w; a.ld_dh = ld_dh; a.R = R; a.V = V; a.variant = variant;
    return K == 128 ? vce_fwd_launch<128>(a, (hipStream_t)stream) : vce_fwd_launch<64>(a, (hipStream_t)stream);
}

template <int KD>
static int vce_lse_launch(VceArgs a, float *lse2, hipStream_t st) {
    const int nh = vce_token_nh(a.R);
    const int64_t ntt = ceil_div64(a.R, 128 * nh);
    const int nvt = (a.V + 127) / 128;
    a.parts = vce_pick_split(ntt, nvt, 0.005);
    const size_t lds = vce_token_lds<KD>();
    static thread_local bool done = false;
    if (!done) { vce_allow_lds(vce_token_kernel<KD, 0, 1>, lds); vce_allow_lds(vce_token_kernel<KD, 0, 2>, lds); done = true; }
    a.ntt = (int)ntt;
    if (nh == 2) vce_token_kernel<KD, 0, 2><<<(unsigned)(ntt * a.parts), 512, lds, st>>>(a);
    else vce_token_kernel<KD, 0, 1><<<(unsigned)(ntt * a.parts), 512, lds, st>>>(a);
    vce_lse_kernel<<<(unsigned)ceil_div64(a.R, 256), 256, 0, st>>>(a, lse2);
    return b4c_check_launch("vocab_lse");
}

extern "C" int b4c_vocab_lse(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, float *lse2, void *workspace,
                             int64_t workspace_bytes, int64_t R, int V, int K, void *stream) {
    B4C_REQUIRE(h && wt && lse2 && workspace, "vocab_lse: null pointer");
    B4C_REQUIRE(vce_shape_ok(K), "vocab_lse: K=%d unsupported (64 or 128)", K);
    B4C_REQUIRE(R >= 0 && V > 0 && ld_h >= K && ld_w >= K, "vocab_lse: shape");
    B4C_REQUIRE(ld_h % 8 == 0 && ld_w % 8 == 0 && ((((uintptr_t)h | (uintptr_t)wt | (uintptr_t)workspace) & 15) == 0),
                "vocab_lse: operands must be 16-byte aligned with pitches % 8 == 0");
    B4C_REQUIRE(workspace_bytes >= b4c_vocab_ce_workspace_bytes(R, V, K), "vocab_lse: workspace too small");
    if (R == 0) return B4C_OK;
    VceArgs a = {};
    a.h = (const bf16_t *)h; a.wt = (const bf16_t *)wt; a.bias = bias; a.st1 = (float *)workspace;
    a.ld_h = ld_h; a.ld_w = ld_w; a.R = R; a.V = V; a.variant = B4C_CE_PLAIN;
    return K == 128 ? vce_lse_launch<128>(a, lse2, (hipStream_t)stream) : vce_lse_launch<64>(a, lse2, (hipStream_t)stream);
}

template <int KD>
static void vce_dw_sweep_launch(VceDwArgs a, int vt0, int nvt, int background_wgs, bool deterministic, hipStream_t st) {
    const int64_t ntt = ceil_div64(a.R, 128);
    const size_t lds = vce_dw_lds<KD>();
    static thread_local bool done = false;
    if (!done) { vce_allow_lds(vce_dw_kernel<KD, 2>, lds); vce_allow_lds(vce_dw_kernel<KD, 1>, lds); done = true; }
    a.vt0 = vt0; a.nvt = nvt;
    if (background_wgs > 0) {
        // beside other kernels: at most background_wgs workgroups of 4 waves (one per SIMD), whole rounds of units
        // the token split that fills whole rounds best (units / (rounds x workgroups)); at least two rounds when the
        // tokens allow it, fewer splits (fewer dW atomics) among near-equals
        int ts = 1;
        double best = -1.0;
        for (int t = 1; t <= 16 && t <= ntt; ++t) {
            const int64_t units = (int64_t)nvt * t, rounds = ceil_div64(units, background_wgs);
            const double fill = (double)units / (double)(rounds * background_wgs) - (rounds < 2 ? 0.5 : 0.0);
            if (fill > best + 0.02) { best = fill; ts = t; }
        }
        if (deterministic) ts = 1;          // one workgroup per vocabulary tile walks every token: plain adds, a fixed order
        a.tsplit = ts;
        const int64_t units = (int64_t)nvt * ts, rounds = ceil_div64(units, background_wgs);
        vce_dw_kernel<KD, 1><<<(unsigned)ceil_div64(units, rounds), 256, lds, st>>>(a);
    } else {
        a.tsplit = deterministic ? 1 : vce_pick_split(nvt, ntt, 0.025);
        vce_dw_kernel<KD, 2><<<(unsigned)(nvt * a.tsplit), 512, lds, st>>>(a);
    }
}
template <int KD>
static int vce_dw_label_launch(VceDwArgs a, const int32_t *labels, float *tmp, int64_t tmp_bytes, bool deterministic, hipStream_t st) {
    (void)hipMemsetAsync(tmp, 0, (size_t)a.V * KD * 4, st);
    if (deterministic) {
        char *p = (char *)tmp + (((size_t)a.V * KD * 4 + 63) & ~(size_t)63);
        int64_t *keys = (int64_t *)p;          p += (size_t)a.R * 8;
        int32_t *order = (int32_t *)p;         p += (size_t)a.R * 4;
        const int64_t sort_bytes = b4c_sort_ids_workspace_bytes(a.R, a.V + 1);
        B4C_REQUIRE(p + sort_bytes <= (char *)tmp + tmp_bytes, "vocab_ce_dw_labels: workspace too small for the deterministic form");
        vce_label_keys_kernel<<<(unsigned)ceil_div64(a.R, 256), 256, 0, st>>>(labels, a.R, a.V, keys);
        if (int rc = b4c_sort_ids(keys, a.R, a.V + 1, order, p, sort_bytes, st)) return rc;
        vce_label_sorted_kernel<KD><<<(unsigned)ceil_div64(a.R, 4), 256, 0, st>>>(a, keys, order, tmp);
    } else {
        vce_label_kernel<KD><<<(unsigned)ceil_div64(a.R, VCE_LABEL_ROWS), 256, 0, st>>>(a, labels, tmp);
    }
    vce_label_add_kernel<KD><<<dim3((unsigned)((a.V + 31) / 32), KD / 32), 256, 0, st>>>(a.dW, a.ldw, tmp, a.V);
    return B4C_OK;
}

static int vce_dw_check(const void *h, int ld_h, const void *wt, int ld_w, const float *rowscal, float *dW, int ldw, int64_t R, int V,
                        int K, const char *who) {
    B4C_REQUIRE(h && wt && rowscal && dW, "%s: null pointer", who);
    B4C_REQUIRE(vce_shape_ok(K), "%s: K=%d unsupported (64 or 128)", who, K);
    B4C_REQUIRE(R >= 0 && V > 0 && ld_h >= K && ld_w >= K && ldw >= V, "%s: shape", who);
    B4C_REQUIRE(ld_h % 8 == 0 && ld_w % 8 == 0 && ((((uintptr_t)h | (uintptr_t)wt | (uintptr_t)rowscal) & 15) == 0),
                "%s: operands must be 16-byte aligned with pitches %% 8 == 0", who);
    return B4C_OK;
}
static VceDwArgs vce_dw_args(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, const float *rowscal, float *dW,
                             int ldw, float *db, int64_t R, int V) {
    VceDwArgs a = {};
    a.h = (const bf16_t *)h; a.wt = (const bf16_t *)wt; a.bias = bias; a.rowscal = rowscal; a.dW = dW; a.db = db;
    a.ld_h = ld_h; a.ld_w = ld_w; a.ldw = ldw; a.R = R; a.V = V;
    return a;
}

extern "C" int b4c_vocab_ce_dw(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, const int32_t *labels,
                               const float *rowscal, float *dW, int ldw, float *db, void *workspace, int64_t workspace_bytes,
                               int64_t R, int V, int K, int deterministic, void *stream) {
    if (int rc = vce_dw_check(h, ld_h, wt, ld_w, rowscal, dW, ldw, R, V, K, "vocab_ce_dw")) return rc;
    B4C_REQUIRE(labels && workspace, "vocab_ce_dw: null pointer");
    B4C_REQUIRE(workspace_bytes >= (int64_t)V * K * 4, "vocab_ce_dw: workspace too small");
    if (R == 0) return B4C_OK;
    const VceDwArgs a = vce_dw_args(h, ld_h, wt, ld_w, bias, rowscal, dW, ldw, db, R, V);
    const int nvt = (V + 127) / 128;
    const bool det = deterministic != 0;
    int rc;
    if (K == 128) { vce_dw_sweep_launch<128>(a, 0, nvt, 0, det, (hipStream_t)stream); rc = vce_dw_label_launch<128>(a, labels, (float *)workspace, workspace_bytes, det, (hipStream_t)stream); }
    else { vce_dw_sweep_launch<64>(a, 0, nvt, 0, det, (hipStream_t)stream); rc = vce_dw_label_launch<64>(a, labels, (float *)workspace, workspace_bytes, det, (hipStream_t)stream); }
    if (rc) return rc;
    return b4c_check_launch("vocab_ce_dw");
}

extern "C" int b4c_vocab_ce_dw_sweep(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, const float *rowscal,
                                     float *dW, int ldw, float *db, int64_t R, int V, int K, int tile_begin, int tile_end,
                                     int background_workgroups, int deterministic, void *stream) {
    if (int rc = vce_dw_check(h, ld_h, wt, ld_w, rowscal, dW, ldw, R, V, K, "vocab_ce_dw_sweep")) return rc;
    const int nvt = (V + 127) / 128;
    B4C_REQUIRE(tile_begin >= 0 && tile_begin <= tile_end && tile_end <= nvt, "vocab_ce_dw_sweep: tiles [%d, %d) of %d", tile_begin, tile_end, nvt);
    B4C_REQUIRE(background_workgroups >= 0, "vocab_ce_dw_sweep: background_workgroups %d", background_workgroups);
    if (R == 0 || tile_begin == tile_end) return B4C_OK;
    const VceDwArgs a = vce_dw_args(h, ld_h, wt, ld_w, bias, rowscal, dW, ldw, db, R, V);
    if (K == 128) vce_dw_sweep_launch<128>(a, tile_begin, tile_end - tile_begin, background_workgroups, deterministic != 0, (hipStream_t)stream);
    else vce_dw_sweep_launch<64>(a, tile_begin, tile_end - tile_begin, background_workgroups, deterministic != 0, (hipStream_t)stream);
    return b4c_check_launch("vocab_ce_dw_sweep");
}

extern "C" int b4c_vocab_ce_dw_labels(const void *h, int ld_h, const int32_t *labels, const float *rowscal, float *dW, int ldw,
                                      float *db, void *workspace, int64_t workspace_bytes, int64_t R, int V, int K, int deterministic,
                                      void *stream) {
    B4C_REQUIRE(h && labels && rowscal && dW && workspace, "vocab_ce_dw_labels: null pointer");
    B4C_REQUIRE(vce_shape_ok(K), "vocab_ce_dw_labels: K=%d unsupported (64 or 128)", K);
    B4C_REQUIRE(R >= 0 && V > 0 && ld_h >= K && ldw >= V, "vocab_ce_dw_labels: shape");
    B4C_REQUIRE(workspace_bytes >= (int64_t)V * K * 4, "vocab_ce_dw_labels: workspace too small");
    if (R == 0) return B4C_OK;
    const VceDwArgs a = vce_dw_args(h, ld_h, nullptr, 0, nullptr, rowscal, dW, ldw, db, R, V);
    const int rc = K == 128 ? vce_dw_label_launch<128>(a, labels, (float *)workspace, workspace_bytes, deterministic != 0, (hipStream_t)stream)
                            : vce_dw_label_launch<64>(a, labels, (float *)workspace, workspace_bytes, deterministic != 0, (hipStream_t)stream);
    if (rc) return rc;
    return b4c_check_launch("vocab_ce_dw_labels");
}

// ==========================================================================================================
// Logits-free RANKING over the vocabulary (R15: tf.math.top_k + Recall / NDCG, utils.py:161-190, 225-255) for the bf16
// scoring path: the (R x V) scores never reach HBM.  Softmax is monotone, so the logits rank as the probabilities do.
//   b4c_vocab_rank   the metrics need one number per row: how many items rank before the true one,
//                    rank = #{j : x_j > x_y} + #{j < y : x_j == x_y}   (tf.math.top_k: ties -> lower index first);
//                    HitRate@k = [rank < k], NDCG@k = [rank < k] / log2(rank + 2) for EVERY k from one sweep.
//                    x_y is produced first by the same MFMA chain on gathered label rows (bit-identical to the sweep's
//                    own value of that entry), then one sweep counts.
//   b4c_vocab_topk   the ids: sweep A keeps, per lane, the running maximum of each of its 16 accumulator slots (16 disjoint
//                    classes of the lane's share of the vocabulary; parts x 4 lanes per token -> 16 x parts x 4 disjoint
//                    classes per row); the k-th largest class maximum tau is a lower bound of the row's k-th largest
//                    score (k distinct entries reach it).  Sweep B recomputes the logits and appends every entry >= tau
//                    to the row's candidate list (about k + 1 of them); one wave per row orders the candidates (score
//                    descending, index ascending).  Rows with more than VCE_CAND candidates (mass ties) are reported in
//                    `overflow` with ids -1: the caller ranks those rows on materialised scores.
// The logits are formed as b4c_gemm_nt forms them (accumulate from zero over k, add the bias last), so that on operands
// whose products are exact the ids are those of b4c_topk_rows on the materialised fp32 logits.
// ==========================================================================================================
#define VCE_CAND 128
enum { SCAN_RANK = 0, SCAN_CLASSMAX = 1, SCAN_COLLECT = 2 };

struct VceScanArgs {
    const bf16_t *h;
    const bf16_t *wt;
    const float *bias;
    const int32_t *labels;    // RANK
    const float *xy;          // RANK: the label's logit per row (+inf: row ignored)
    int32_t *rank;            // RANK: += counts (pre-set by vce_label_logit_kernel)
    float *cm;                // CLASSMAX: [parts * 4][R][16]
    const float *tau;         // COLLECT: [R]
    int32_t *cnt;             // COLLECT: [R] candidates so far
    float *cand_v;            // COLLECT: [R][VCE_CAND]
    int32_t *cand_i;
    int ld_h, ld_w;
    int64_t R;
    int V, parts, ntt;
};

// x_y[row] = h_row . W[y] + b[y] through the MFMA chain of the sweeps (32 rows per wave: A = the 32 label rows of W gathered
// straight from memory, B = the 32 token rows; the diagonal of the 32 x 32 tile).  Rows without a valid label: +inf, rank -1.
template <int KD>
__global__ void __launch_bounds__(64) vce_label_logit_kernel(VceScanArgs a, float *__restrict__ xy, int32_t *__restrict__ rank) {
    constexpr int NKS = KD / 16;
    const int lane = threadIdx.x, r = lane & 31, hf = lane >> 5;
    const int64_t tok = (int64_t)blockIdx.x * 32 + r;
    const int y = tok < a.R ? a.labels[tok] : -1;
    const bool valid = y >= 0 && y < a.V;
    bf16x8 hfr[NKS];
    vce_load_hfrag<KD>(a.h, a.ld_h, tok, a.R, hf, hfr);
    f32x16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        vu32x4 q = {0u, 0u, 0u, 0u};
        if (valid) q = *reinterpret_cast<const vu32x4 *>(a.wt + (int64_t)y * a.ld_w + ks * 16 + hf * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, q), hfr[ks], acc, 0, 0, 0);
    }
    // D[row i][col j] = W[y_i] . h_j: the diagonal element of token r sits in the lane with hf = (r >> 2) & 1, register
    // (r & 3) + 4 (r >> 3)
    const int t_diag = (r & 3) + 4 * (r >> 3);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) v = (t == t_diag) ? acc[t] : v;
    if (hf == ((r >> 2) & 1) && tok < a.R) {
        xy[tok] = valid ? v + (a.bias ? a.bias[y] : 0.f) : INFINITY;
        rank[tok] = valid ? 0 : -(1 << 30);
    }
}

// NH = 1: 128 tokens per workgroup, 8 waves = 4 token groups x the 2 halves of each 128-row W tile;
// NH = 2: 256 tokens per workgroup, 8 token groups, every wave takes both halves in turn -- each W tile (32 KB by LDS-DMA
// from L2 / Infinity Cache: the sweeps stream the whole of W once per token tile, 4.1 GB per sweep at C2, and that stream, not
// the matrix pipe, paces them) serves twice the tokens.
template <int KD, int OP, int NH>
__global__ void __launch_bounds__(512, 2) vce_scan_kernel(VceScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NKS = KD / 16, STR = VTile<KD>::STR;
    constexpr int TILE_B = VTile<KD>::BYTES;
    float *sBias = reinterpret_cast<float *>(smem + 2 * TILE_B);     // [3][128]: a ring -- the scores of a tile's second half are
                                                                     // formed one tile later, while the next bias arrives
    const int unit = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int tg = NH == 2 ? wave : (wave & 3), vh = NH == 2 ? 0 : (wave >> 2);
    const int64_t tok0 = (int64_t)(unit % a.ntt) * (128 * NH);
    const int64_t tok = tok0 + tg * 32 + r;
    const int part = unit / a.ntt;
    const int nvt = (a.V + 127) >> 7;
    const int vt0 = (int)((int64_t)nvt * part / a.parts), vt1 = (int)((int64_t)nvt * (part + 1) / a.parts);
    const bool live = tok < a.R;

    bf16x8 hfr[NKS];
    vce_load_hfrag<KD>(a.h, a.ld_h, tok, a.R, hf, hfr);
    int foff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) foff[ks] = VTile<KD>::frag_off(r, ks, hf) + vh * 64 * STR;

    // per-lane state
    float ref = INFINITY;          // RANK: x_y;  COLLECT: tau  (+inf: nothing counts / nothing is collected)
    int y = -1;
    if (OP == SCAN_RANK && live) { ref = a.xy[tok]; y = a.labels[tok]; }
    if (OP == SCAN_COLLECT && live) ref = a.tau[tok];
    unsigned n_before = 0;         // RANK
    float cm[16];                  // CLASSMAX
#pragma unroll
    for (int t = 0; t < 16; ++t) cm[t] = -INFINITY;

    float breg = 0.f;
    auto fetch = [&](int vt, int buf) {
        VTile<KD>::template dma<512>(a.wt, a.ld_w, (int64_t)vt * 128, vt < vt1 ? a.V : 0, smem + buf * TILE_B, tid);
        if (tid < 128) {
            const int v = vt * 128 + tid;
            breg = (vt < vt1 && v < a.V) ? (a.bias ? a.bias[v] : 0.f) : -INFINITY;   // rows past V: score = -inf
        }
    };
    fetch(vt0, 0);
    if (tid < 128) sBias[tid] = breg;
    VCE_DMA_WAIT();
    __syncthreads();

    // One half-tile (64 vocabulary rows x the wave's 32 tokens) = 16 MFMAs into acc, then ~4 VALU instructions per entry
    // on the result.  A VALU wave-instruction holds the SIMD's issue port for 4 cycles, an MFMA for 8 of its 32: run one
    // after the other the two phases add up (measured: matrix pipe 36 % busy, VALU issue 43 %, sum 79 % of the kernel's
    // cycles); interleaved -- the MFMA chain of one half-tile issued between the VALU instructions of the previous one --
    // they overlap.  So the loop is software-pipelined by half a tile: `scores` of half-tile i runs inside the instruction
    // stream of `chain` of half-tile i + 1 (sched_group_barrier pins the interleave), on two accumulator sets.
    // one entry of a half-tile's scores: x = accumulator + bias (the bias last, as the materialising GEMM adds it; rows past
    // V: -inf); RANK: count it if it beats x_y, note an equal one; CLASSMAX: the running maximum of its accumulator slot;
    // COLLECT: note one that reaches tau
    bool hot = false;
    auto entry = [&](f32x16 (&acc)[2], int rt, int t, float bj) __attribute__((always_inline)) {
        const float x = acc[rt][t] + bj;
        acc[rt][t] = x;
        if (OP == SCAN_RANK) {
            n_before += x > ref ? 1u : 0u;
            hot |= x == ref;
        } else if (OP == SCAN_CLASSMAX) {
            cm[t] = fmaxf(cm[t], x);
        } else {
            hot |= x >= ref;
        }
    };
    // The 16 MFMAs of a half-tile's chain into accN, and -- WITH = true -- between them the scores of the half-tile before
    // it (accP: 32 entries per lane, two per MFMA).  sched_barrier(0) after every MFMA's group pins the interleave (left to
    // itself, or to sched_group_barrier, the compiler issues the sixteen MFMAs first and the VALU after them).
    auto chain = [&](auto WITH, f32x16 (&accN)[2], const char *w, f32x16 (&accP)[2], const float *bs, int vhe) __attribute__((always_inline)) {
        constexpr bool with = decltype(WITH)::value;
        bf16x8 wfq[NKS];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < 16; ++t) accN[rt][t] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) wfq[ks] = *reinterpret_cast<const bf16x8 *>(w + foff[ks]);
        hot = false;
        // the eight bias quads of the previous half-tile are requested up front, with the first fragments: a quad requested
        // where it is used parks the wave for a full LDS round trip (~130 cycles) sixteen times per tile
        f32x4 bq[8];
        if (with) {
#pragma unroll
            for (int q = 0; q < 8; ++q) bq[q] = *reinterpret_cast<const f32x4 *>(bs + vhe * 64 + (q >> 2) * 32 + 8 * (q & 3) + 4 * hf);
        }
#pragma unroll
        for (int i = 0; i < 2 * NKS; ++i) {
            const int rt = i / NKS, ks = i % NKS;
            __builtin_amdgcn_sched_barrier(0);
            accN[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfq[ks], hfr[ks], accN[rt], 0, 0, 0);
            if (rt == 0) wfq[ks] = *reinterpret_cast<const bf16x8 *>(w + 32 * STR + foff[ks]);
            if (with) {
                // entries 2 i, 2 i + 1 of the previous half-tile (NKS = 8: all 32; NKS = 4: the rest follows the chain)
#pragma unroll
                for (int e = 2 * i; e < 2 * i + 2; ++e) entry(accP, e >> 4, e & 15, bq[e >> 2][e & 3]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (with && NKS < 8) {
#pragma unroll
            for (int e = 4 * NKS; e < 32; ++e) entry(accP, e >> 4, e & 15, bq[e >> 2][e & 3]);
        }
    };
    // the scores of a half-tile on their own (NH = 1; the last half-tile of NH = 2)
    auto scores = [&](f32x16 (&acc)[2], const float *bs, int vhe) __attribute__((always_inline)) {
        hot = false;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bs + vhe * 64 + rt * 32 + 8 * tq + 4 * hf);
#pragma unroll
                for (int k = 0; k < 4; ++k) entry(acc, rt, 4 * tq + k, b4[k]);
            }
    };
    auto rare = [&](f32x16 (&acc)[2], int vt, int vhe) __attribute__((always_inline)) {       // the half-tile that holds the label / a candidate
        const int row0 = vt * 128 + vhe * 64 + 4 * hf;
        int slot = 0;
        if (OP == SCAN_COLLECT) {          // the lane's candidates of this half-tile take consecutive slots: one atomic
            int n = 0;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    n += (acc[rt][t] >= ref && row0 + rt * 32 + (t & 3) + 8 * (t >> 2) < a.V) ? 1 : 0;
            if (n) slot = atomicAdd(a.cnt + tok, n);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = row0 + rt * 32 + (t & 3) + 8 * (t >> 2);
                const float x = acc[rt][t];
                if (OP == SCAN_RANK) {
                    n_before += (x == ref && j < y) ? 1u : 0u;        // ties: the lower index ranks first
                } else if (x >= ref && j < a.V) {
                    if (slot < VCE_CAND) {
                        a.cand_v[tok * VCE_CAND + slot] = x;
                        a.cand_i[tok * VCE_CAND + slot] = j;
                    }
                    ++slot;
                }
            }
    };
    f32x16 accA[2], accB[2];
    int vt_prev = vt0;
    bool have_prev = false;
    int bcur = 0, bprev = 2;           // bias ring slots of this tile and of the previous one; the next one's goes to the third
#ifdef VCE_SCAN_STAMPS
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#endif
    constexpr std::integral_constant<bool, true> YES{};
    constexpr std::integral_constant<bool, false> NO{};
    auto tile = [&](auto BUF, int vt) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUF)::value;
        VCE_STAMP(5);
        fetch(vt + 1, buf ^ 1);
        VCE_STAMP(0);
        const char *w = smem + buf * TILE_B;
        const int bnext = 3 - bcur - bprev;
        if (NH == 2) {
            // accA <- half 0 of this tile, beside the scores of the previous tile's half 1 (accB)
            if (have_prev) {
                chain(YES, accA, w, accB, sBias + bprev * 128, 1);
                if (OP != SCAN_CLASSMAX && __any(hot)) rare(accB, vt_prev, 1);
            } else {
                chain(NO, accA, w, accB, sBias, 0);
            }
            VCE_STAMP(1);
            // accB <- half 1, beside the scores of half 0
            chain(YES, accB, w + 64 * STR, accA, sBias + bcur * 128, 0);
            if (OP != SCAN_CLASSMAX && __any(hot)) rare(accA, vt, 0);
            VCE_STAMP(2);
            have_prev = true;
            vt_prev = vt;
        } else {
            chain(NO, accA, w, accB, sBias, 0);
            scores(accA, sBias + bcur * 128, vh);
            if (OP != SCAN_CLASSMAX && __any(hot)) rare(accA, vt, vh);
        }
        if (tid < 128) sBias[bnext * 128 + tid] = breg;
        bprev = bcur;
        bcur = bnext;
        VCE_DMA_WAIT();
        VCE_STAMP(3);
        B4C_LDS_BARRIER();
        VCE_STAMP(4);
    };
    for (int vt = vt0; vt < vt1; vt += 2) {
        tile(std::integral_constant<int, 0>{}, vt);
        if (vt + 1 < vt1) tile(std::integral_constant<int, 1>{}, vt + 1);
    }
    if (NH == 2 && have_prev) {          // the last half-tile's scores
        scores(accB, sBias + bprev * 128, 1);
        if (OP != SCAN_CLASSMAX && __any(hot)) rare(accB, vt_prev, 1);
    }
#ifdef VCE_SCAN_STAMPS
    if (lane == 0 && blockIdx.x < 2048)
        for (int k = 0; k < 6; ++k) g_vce_stamps[(blockIdx.x * 8 + wave) * 6 + k] = st_[k];
#endif
    if (!live) return;
    if (OP == SCAN_RANK) {
        if (n_before) atomicAdd(a.rank + tok, (int)n_before);          // integer adds: any order gives the same count
    } else if (OP == SCAN_CLASSMAX) {
        // sub-list index: (part, half of the tile, lane half) for NH = 1; (part, lane half) for NH = 2 (a.nsub_per_part of them)
        const int sub = NH == 2 ? part * 2 + hf : part * 4 + vh * 2 + hf;
        float *o = a.cm + ((int64_t)sub * a.R + tok) * 16;
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) *reinterpret_cast<f32x4 *>(o + 4 * tq) = (f32x4){cm[4 * tq], cm[4 * tq + 1], cm[4 * tq + 2], cm[4 * tq + 3]};
    }
}

// tau[row] = k-th largest of the row's nsub * 16 class maxima (one wave per row; k rounds of "take the maximum out")
__global__ void __launch_bounds__(256) vce_tau_kernel(const float *__restrict__ cm, int nsub, int64_t R, int k, float *__restrict__ tau,
                                                      int32_t *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    // nsub <= 32 -> at most 512 values, 8 per lane: value e of the row lives in sub-list e / 16, slot e % 16
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i;
        v[i] = e < nsub * 16 ? cm[((int64_t)(e >> 4) * R + row) * 16 + (e & 15)] : -INFINITY;
    }
    float kth = -INFINITY;
    for (int round = 0; round < k; ++round) {
        float m = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) m = fmaxf(m, v[i]);
        const float wm = wave_max(m);
        kth = wm;
        const unsigned long long owners = __ballot(m == wm);
        if (lane == __ffsll((long long)owners) - 1) {          // one instance leaves
            bool done = false;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (!done && v[i] == wm) { v[i] = -INFINITY; done = true; }
        }
    }
    if (lane == 0) { tau[row] = kth; cnt[row] = 0; }
}

// one wave per row: the candidates in order (score descending, index ascending) -> ids [k], optional hit / ndcg of the label
__global__ void __launch_bounds__(256) vce_select_kernel(const float *__restrict__ cand_v, const int32_t *__restrict__ cand_i,
                                                         const int32_t *__restrict__ cnt, int64_t R, int V, int k, int32_t *__restrict__ idx,
                                                         const int32_t *__restrict__ labels, float *__restrict__ hit, float *__restrict__ ndcg,
                                                         int32_t *__restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int n = cnt[row];
    const int y = labels ? labels[row] : -1;
    if (n > VCE_CAND) {            // mass ties: the caller ranks this row on materialised scores
        if (lane < k) idx[row * k + lane] = -1;
        if (lane == 0) {
            atomicAdd(overflow, 1);
            if (hit) { hit[row] = __builtin_nanf(""); ndcg[row] = __builtin_nanf(""); }
        }
        return;
    }
    float v[2];
    int id[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = lane + 64 * i;
        v[i] = e < n ? cand_v[row * VCE_CAND + e] : -INFINITY;
        id[i] = e < n ? cand_i[row * VCE_CAND + e] : 0x7fffffff;
    }
    int rk[2] = {0, 0};
    for (int j = 0; j < n; ++j) {          // wave-uniform trip count
        const float vj = __shfl(j < 64 ? v[0] : v[1], j & 63);
        const int ij = __shfl(j < 64 ? id[0] : id[1], j & 63);
#pragma unroll
        for (int i = 0; i < 2; ++i) rk[i] += (vj > v[i] || (vj == v[i] && ij < id[i])) ? 1 : 0;
    }
    float h = 0.f, g = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = lane + 64 * i;
        if (e < n && rk[i] < k) {
            idx[row * k + rk[i]] = id[i];
            if (id[i] == y) { h = 1.f; g = 1.0f / (logf((float)(rk[i] + 2)) / logf(2.0f)); }
        }
    }
    if (n < k && lane >= n && lane < k) idx[row * k + lane] = -1;          // fewer than k items in all
    if (hit) {
        h = wave_sum(h);
        g = wave_sum(g);
        if (lane == 0) { hit[row] = h; ndcg[row] = g; }
    }
}

// hit@k / ndcg@k from the rank of the true item (utils.py:176-190, 245-255); rows with rank < 0 (no valid label): 0
__global__ void __launch_bounds__(256) vce_rank_metrics_kernel(const int32_t *__restrict__ rank, int64_t R, int k, float *__restrict__ hit,
                                                               float *__restrict__ ndcg) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= R) return;
    const int rk = rank[i];
    const bool in = rk >= 0 && rk < k;
    hit[i] = in ? 1.f : 0.f;
    ndcg[i] = in ? 1.0f / (logf((float)(rk + 2)) / logf(2.0f)) : 0.f;
}

extern "C" int64_t b4c_vocab_rank_workspace_bytes(int64_t R, int V, int K) {
    if (R <= 0 || V <= 0 || !vce_shape_ok(K)) return 0;
    // class maxima [32][R][16] | tau [R] | cnt [R] | candidates [R][VCE_CAND] x (score, id) | x_y [R]
    return R * ((int64_t)32 * 16 * 4 + 4 + 4 + (int64_t)VCE_CAND * 8 + 4) + 256;
}

static int vce_scan_check(const void *h, int ld_h, const void *wt, int ld_w, void *workspace, int64_t workspace_bytes, int64_t R, int V,
                          int K, const char *who) {
    B4C_REQUIRE(h && wt && workspace, "%s: null pointer", who);
    B4C_REQUIRE(vce_shape_ok(K), "%s: K=%d unsupported (64 or 128)", who, K);
    B4C_REQUIRE(R >= 0 && V > 0 && ld_h >= K && ld_w >= K, "%s: shape", who);
    B4C_REQUIRE(ld_h % 8 == 0 && ld_w % 8 == 0 && ((((uintptr_t)h | (uintptr_t)wt | (uintptr_t)workspace) & 15) == 0),
                "%s: operands must be 16-byte aligned with pitches %% 8 == 0", who);
    B4C_REQUIRE(workspace_bytes >= b4c_vocab_rank_workspace_bytes(R, V, K), "%s: workspace too small", who);
    return B4C_OK;
}

// tokens per workgroup of the scans: 256 (each W tile serves twice the tokens) once there are enough token tiles to fill the
// chip; B4C_VCE_SCAN_TOKENS=128|256 overrides (A/B)
static int vce_scan_nh(int64_t R) {
    static const char *e = getenv("B4C_VCE_SCAN_TOKENS");
    if (e) return atoi(e) >= 256 ? 2 : 1;
    return R >= 256 * 64 ? 2 : 1;
}
static void vce_scan_geometry(VceScanArgs &a, int nh) {
    a.ntt = (int)ceil_div64(a.R, 128 * nh);
    a.parts = vce_pick_split(a.ntt, (a.V + 127) / 128, 0.005);
}
template <int KD, int OP>
static void vce_scan_launch(VceScanArgs a, int nh, hipStream_t st) {
    const size_t lds = 2 * (size_t)VTile<KD>::BYTES + 3 * 128 * 4;
    static thread_local bool done = false;
    if (!done) { vce_allow_lds(vce_scan_kernel<KD, OP, 1>, lds); vce_allow_lds(vce_scan_kernel<KD, OP, 2>, lds); done = true; }
    if (nh == 2) vce_scan_kernel<KD, OP, 2><<<(unsigned)(a.ntt * a.parts), 512, lds, st>>>(a);
    else vce_scan_kernel<KD, OP, 1><<<(unsigned)(a.ntt * a.parts), 512, lds, st>>>(a);
}

extern "C" int b4c_vocab_rank(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, const int32_t *labels, int32_t *rank,
                              void *workspace, int64_t workspace_bytes, int64_t R, int V, int K, void *stream) {
    if (int rc = vce_scan_check(h, ld_h, wt, ld_w, workspace, workspace_bytes, R, V, K, "vocab_rank")) return rc;
    B4C_REQUIRE(labels && rank, "vocab_rank: null pointer");
    if (R == 0) return B4C_OK;
    hipStream_t st = (hipStream_t)stream;
    VceScanArgs a = {};
    a.h = (const bf16_t *)h; a.wt = (const bf16_t *)wt; a.bias = bias; a.labels = labels; a.rank = rank;
    a.ld_h = ld_h; a.ld_w = ld_w; a.R = R; a.V = V;
    const int nh = vce_scan_nh(R);
    vce_scan_geometry(a, nh);
    float *xy = (float *)workspace;
    a.xy = xy;
    if (K == 128) {
        vce_label_logit_kernel<128><<<(unsigned)ceil_div64(R, 32), 64, 0, st>>>(a, xy, rank);
        vce_scan_launch<128, SCAN_RANK>(a, nh, st);
    } else {
        vce_label_logit_kernel<64><<<(unsigned)ceil_div64(R, 32), 64, 0, st>>>(a, xy, rank);
        vce_scan_launch<64, SCAN_RANK>(a, nh, st);
    }
    return b4c_check_launch("vocab_rank");
}

extern "C" int b4c_rank_metrics(const int32_t *rank, int64_t R, int k, float *hit, float *ndcg, void *stream) {
    B4C_REQUIRE(rank && hit && ndcg && R >= 0 && k >= 1, "rank_metrics: bad argument");
    if (R == 0) return B4C_OK;
    vce_rank_metrics_kernel<<<(unsigned)ceil_div64(R, 256), 256, 0, (hipStream_t)stream>>>(rank, R, k, hit, ndcg);
    return b4c_check_launch("rank_metrics");
}

extern "C" int b4c_vocab_topk(const void *h, int ld_h, const void *wt, int ld_w, const float *bias, int k, int32_t *idx,
                              const int32_t *labels, float *hit, float *ndcg, int32_t *overflow, void *workspace,
                              int64_t workspace_bytes, int64_t R, int V, int K, void *stream) {
    if (int rc = vce_scan_check(h, ld_h, wt, ld_w, workspace, workspace_bytes, R, V, K, "vocab_topk")) return rc;
    B4C_REQUIRE(idx && overflow && k >= 1 && k <= B4C_MAX_TOPK, "vocab_topk: k = %d (1 .. %d)", k, B4C_MAX_TOPK);
    B4C_REQUIRE(!labels || (hit && ndcg), "vocab_topk: labels need hit and ndcg");
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(overflow, 0, 4, st);
    if (R == 0) return B4C_OK;
    VceScanArgs a = {};
    a.h = (const bf16_t *)h; a.wt = (const bf16_t *)wt; a.bias = bias;
    a.ld_h = ld_h; a.ld_w = ld_w; a.R = R; a.V = V;
    const int nh = vce_scan_nh(R);
    vce_scan_geometry(a, nh);
    char *ws = (char *)workspace;
    a.cm = (float *)ws;                         ws += (size_t)R * 32 * 16 * 4;
    float *tau = (float *)ws;                   ws += (size_t)R * 4;
    a.cnt = (int32_t *)ws;                      ws += (size_t)R * 4;
    a.cand_v = (float *)ws;                     ws += (size_t)R * VCE_CAND * 4;
    a.cand_i = (int32_t *)ws;
    a.tau = tau;
    if (K == 128) vce_scan_launch<128, SCAN_CLASSMAX>(a, nh, st); else vce_scan_launch<64, SCAN_CLASSMAX>(a, nh, st);
    vce_tau_kernel<<<(unsigned)ceil_div64(R, 4), 256, 0, st>>>(a.cm, a.parts * (nh == 2 ? 2 : 4), R, k, tau, a.cnt);
    if (K == 128) vce_scan_launch<128, SCAN_COLLECT>(a, nh, st); else vce_scan_launch<64, SCAN_COLLECT>(a, nh, st);
    vce_select_kernel<<<(unsigned)ceil_div64(R, 4), 256, 0, st>>>(a.cand_v, a.cand_i, a.cnt, R, V, k, idx, labels, hit, ndcg, overflow);
    return b4c_check_launch("vocab_topk");
}
