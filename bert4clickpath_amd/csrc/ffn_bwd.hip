// Backward of the position-wise feed-forward block of one encoder layer -- LayerNorm, dropout, both Dense layers -- in ONE pass
// over its tensors (round 4; VERDICT r3 item 1: the encoder backward's byte diet).
//
// The reference's block (transformer.py:154-170: Dense(dff, relu), Dense(d_model), dropout, residual add, LayerNormalization)
// comes back, in the backward pass, as five kernels here until now:
//     add_ln_bwd      dOut, z, stats -> dz (residual branch), dy (through the dropout mask); dgamma, dbeta      4 [T][128] passes
//     gemm_nt (gate)  dh = (dy W2^T) o [h > 0]                                                                    ~2.6
//     gemm_nt (+res)  dx = dh W1^T + dz                                                                            ~2.8
//     gemm_tn x 2     dW2 = h^T dy, dW1 = x^T dh (+ the bias sums)                                                 ~3.6
// 13 passes of [T][128] bf16 over HBM.  dz, dy and dh exist only between these kernels: here they live in registers and LDS, and
// the block's backward reads dOut, z, h, x and writes dx -- 4.8 passes (1,240 B per token against 3,340).
//
// A persistent 512-thread workgroup per CU walks 32-token tiles (tile i of a workgroup = global tile blockIdx.x + i gridDim.x):
//   LDS-DMA     h tile [32][Fp] and x tile [32][128] into a four-stage ring of XOR-swizzled images (three tiles ahead)
//   rows        dOut / z chunks and the rows' statistics arrive by LDS-DMA too, a tile ahead, 16 B per thread at thread * 16 (every
//               request lands in LDS: a register that a load is still filling gets copied by the compiler -- see ln_load below);
//               16 threads per token row hold 8 columns each: LayerNorm backward with 16-lane sums -> dz (kept, bf16, for the
//               residual add two phases later) and dy -> LDS image; dgamma / dbeta partial sums stay in registers over all tiles
//   phase 2     dW2 += h^T dy (MFMA 32x32x16, fragments by transposed LDS reads), dh = (dy W2^T) o [h > 0] (MFMA 16x16x32, the
//               wave's 16 hidden columns of W2 resident in registers) -> LDS image
//   phase 1'    (with the NEXT tile's LayerNorm phase) dW1 += x^T dh, dx = dh W1^T -> staged fp32 tile
//   store       (with the next tile's phase 2) staged dx + dz -> bf16 -> global, 16-B row chunks
//   end         the workgroup's partial sums -> scratch; ffn_bwd_reduce_kernel adds them in a fixed order into the Keras-layout
//               gradient tensors (bit-repeatable: no float atomics anywhere, also not for dgamma / dbeta).
// Two barriers per tile.  d_model = 128, dff <= 128 (the reference hard-codes 100, transformer.py:113), bf16: every other shape
// keeps the five kernels.
#include <stdlib.h>

#include <type_traits>

#include "dxdw_common.h"

#define FB_OSTR 528                  // bytes per staged dx row: 128 fp32 + 16

struct FfnBwdArgs {
    const bf16_t *dOut;   // [M][128]  gradient of the block's output
    const bf16_t *Z;      // [M][128]  x + dropout(y): the LayerNorm's input, saved by the forward pass
    const float *stats;   // [M][2]    mean, 1 / std of z's rows
    const float *gamma;   // [128]
    const bf16_t *H;      // [M][ldh]  relu(x W1 + b1), Fp columns
    const bf16_t *X;      // [M][ldx]  the block's input
    const bf16_t *W2c;    // [Fp][ldw2]  row = hidden column, 128 entries: the operand of dh = dy W2^T
    const bf16_t *W1c;    // [128][ldw1] row = input feature, Fp entries: the operand of dx = dh W1^T
    bf16_t *dX;           // [M][ldo]
    float *part;          // [workgroups][260][128]: dW2^T (row = output column, 128 hidden), dW1^T (row = hidden column, 128 inputs), db2, db1, dgamma, dbeta
    int ldh, ldx, ldw2, ldw1, ldo, Fp;
    float rate;
    uint64_t seed;
    int64_t M;
#ifdef DD_EXPERIMENT
    int debug;            // scratch builds only: the counted wait of the steady state (3) replaced by this many
#endif
};

#define FB_ROWS 260

#ifdef DD_EXPERIMENT
__device__ unsigned g_fb_dbg[4 * 32 * 4];      // [slot][row][0: h chunk != global, 1: == the tile four back, 2: x chunk != global, 3: checks]
extern "C" int b4c_debug_fb(void *dst, size_t nbytes, int clear) {
    if (clear) { static unsigned z[4 * 32 * 4]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fb_dbg), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fb_dbg), nbytes < sizeof(g_fb_dbg) ? nbytes : sizeof(g_fb_dbg));
}
#endif

__global__ void __launch_bounds__(512, 1) ffn_bwd_kernel(FfnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 2 * DD_SUB;                   // h | x
    char *sDY = smem + DD_RING * STAGE;                 // [32][128] bf16 image of dy
    char *sDH = sDY + DD_SUB;                           // [32][128] bf16 image of dh
    char *sOut = sDH + DD_SUB;                          // [32][FB_OSTR] fp32 dx = dh W1^T (before the residual)
    char *sLN = sOut + DD_TOK * FB_OSTR;                // [3][512 threads][16 B]: this thread's dOut chunk, z chunk, statistics pair
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;            // dW tiles: feature half (64 rows), gradient-column quarter (32 columns)
    const int orow = tid >> 4, opart = tid & 15;        // row phases: token row of the tile, 8-column piece
    const int64_t ntile_all = (a.M + DD_TOK - 1) / DD_TOK;
    const int64_t t1 = (ntile_all - blockIdx.x + gridDim.x - 1) / gridDim.x;      // this workgroup's tile count (>= 1)
    const int64_t gstep = gridDim.x, gfirst = blockIdx.x;

    // resident operands of the two dX-type products: this wave's 16 output columns, B[k = 32 ks + 8 g + j][col = 16 wave + li]
    bf16x8 w2f[4], w1f[4];
    {
        const bf16x8 zero = __builtin_bit_cast(bf16x8, (dd_u32x4){0u, 0u, 0u, 0u});
        const int col = 16 * wave + li;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = ks * 32 + 8 * g;
            w2f[ks] = col < a.Fp ? *reinterpret_cast<const bf16x8 *>(a.W2c + (int64_t)col * a.ldw2 + k) : zero;
            w1f[ks] = k < a.Fp ? *reinterpret_cast<const bf16x8 *>(a.W1c + (int64_t)col * a.ldw1 + k) : zero;
        }
    }
    f32x16 acc1[2], acc2[2];                            // dW1^T, dW2^T tiles of this wave: [feature tiles 2 wm + i] x gradient tile wn
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc1[i][t] = acc2[i][t] = 0.f;
    float bsum1 = 0.f, bsum2 = 0.f;
    float gm[8], pg[8], pb[8];
    Vec8<float>::load(a.gamma + 8 * opart, gm);
#pragma unroll
    for (int k = 0; k < 8; ++k) pg[k] = pb[k] = 0.f;

    int toffA[2][2], toffB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { toffA[i][0] = dd_tr_off(hf, li, g, 2 * wm + i, 0); toffA[i][1] = dd_tr_off(hf, li, g, 2 * wm + i, 1); }
    toffB[0] = dd_tr_off(hf, li, g, wn, 0);
    toffB[1] = dd_tr_off(hf, li, g, wn, 1);
    int xoff[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) xoff[mi][q] = dd_chunk_off(16 * mi + li, 4 * q + g);
    // the lane's 4 consecutive columns 16 wave + 4 g .. + 3 of token 16 mi + li inside an image (8-B piece of a 16-B chunk)
    int poff[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) poff[mi] = dd_chunk_off(16 * mi + li, (16 * wave + 4 * g) >> 3) + ((4 * g) & 7) * 2;
    const int rowoff = dd_chunk_off(orow, opart);       // the row phases' 16-B chunk inside an image

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    auto fetch = [&](int64_t t, int slot) {
        const unsigned st = lds0 + (unsigned)(slot * STAGE);
        const int64_t tok0 = (gfirst + t * gstep) * DD_TOK;
        const int64_t M = t < t1 ? a.M : 0;             // past the last tile: zero rows (still LDS writes, still counted)
#ifdef DD_EXPERIMENT
        if (a.debug == 20) {            // x first
            dd_dma(a.X, a.ldx, 0, tok0, M, st + DD_SUB, wave, lane);
            dd_dma(a.H, a.ldh, 0, tok0, M, st, wave, lane, a.Fp * 2);
            return;
        }
        if (a.debug == 21) {            // 64 more wait states between the two
            dd_dma(a.H, a.ldh, 0, tok0, M, st, wave, lane, a.Fp * 2);
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
            dd_dma(a.X, a.ldx, 0, tok0, M, st + DD_SUB, wave, lane);
            return;
        }
#endif
        dd_dma(a.H, a.ldh, 0, tok0, M, st, wave, lane, a.Fp * 2);
        dd_dma(a.X, a.ldx, 0, tok0, M, st + DD_SUB, wave, lane);
    };
    // dOut / z chunks and the row's statistics land in LDS too (LDS-DMA, 16 B per thread at thread * 16: every thread reads back its
    // own piece, so the issuing wave's counted wait is all the ordering it needs).  They landed in REGISTERS first (inline-assembly
    // global loads a tile ahead): the compiler knows nothing of a load in flight and is free to copy such a register whenever it
    // likes -- hipcc 7.2 hoisted the copy of the statistics pair above `s_waitcnt vmcnt(3)` in one of the four unrolled iterations
    // (stale mean / rstd whenever the load had not landed: single rows of lanes 48-63, or whole tiles of every workgroup at once
    // when a page-table miss delayed them all; ~1 launch in 2 at 456 k rows, none below 100 k), and naming the registers on the
    // wait made it copy all of them in front of it.  Nothing can copy LDS.
    dd_u32x4 ds_do, ds_z, ds_st;                        // descriptors of dOut, z (M x 256 B) and stats (M x 8 B): whole tensors
    {
        const uint64_t bd = (uint64_t)a.dOut, bz = (uint64_t)a.Z, bs = (uint64_t)a.stats;
        ds_do[0] = __builtin_amdgcn_readfirstlane((unsigned)bd); ds_do[1] = __builtin_amdgcn_readfirstlane((unsigned)(bd >> 32) & 0xFFFFu);
        ds_z[0] = __builtin_amdgcn_readfirstlane((unsigned)bz); ds_z[1] = __builtin_amdgcn_readfirstlane((unsigned)(bz >> 32) & 0xFFFFu);
        ds_st[0] = __builtin_amdgcn_readfirstlane((unsigned)bs); ds_st[1] = __builtin_amdgcn_readfirstlane((unsigned)(bs >> 32) & 0xFFFFu);
        ds_do[2] = ds_z[2] = __builtin_amdgcn_readfirstlane((unsigned)(a.M * 256));
        ds_st[2] = __builtin_amdgcn_readfirstlane((unsigned)(a.M * 8));
        ds_do[3] = ds_z[3] = ds_st[3] = 0x00020000u;
    }
    const unsigned ln_lds = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(sLN - smem) + (unsigned)(wave * 1024));
    auto ln_load = [&](int64_t t) {
        const int64_t tk = (gfirst + t * gstep) * DD_TOK + orow;
        const unsigned row = (unsigned)(tk < a.M ? tk : a.M - 1);      // (rows past M: read the last row, contribute nothing)
        const unsigned vo = row * 256u + (unsigned)opart * 16u;          // (M x 256 B < 4 GB: checked by the host)
        // statistics: the 16-B window [mean, rstd, mean, rstd] of two rows that holds this row and lies inside the tensor
        const unsigned pair0 = (row & ~1u) + 2u <= (unsigned)a.M ? (row & ~1u) : (unsigned)a.M - 2u;
        const unsigned vs = pair0 * 8u;
        // lgkmcnt(0): this thread's reads of the previous pieces are done before anything can land on them
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(ln_lds), "v"(vo), "s"(ds_do) : "m0", "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(ln_lds + 8192u), "v"(vo), "s"(ds_z) : "m0", "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(ln_lds + 16384u), "v"(vs), "s"(ds_st) : "m0", "memory");
    };
    // the compiler's own loads (W fragments, gamma) are consumed HERE: its wait for them would otherwise sit at their first use inside
    // the tile loop, where it counts none of the requests below and drains them every iteration
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { asm volatile("" : "+v"(w1f[ks])); asm volatile("" : "+v"(w2f[ks])); }
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(gm[k]));
    fetch(0, 0);
    fetch(1, 1);
    fetch(2, 2);
    ln_load(0);

    const float inv_keep = a.rate > 0.f ? 1.0f / (1.0f - a.rate) : 1.0f;
    const uint32_t thr = b4c_keep_threshold(a.rate);
    dd_u32x4 dz0 = {0u, 0u, 0u, 0u}, dz1 = {0u, 0u, 0u, 0u};        // dz of the even / odd tiles, bf16, until their dx rows leave

    // ---- phase 3 of a tile: dW1 += x^T dh, dx = dh W1^T -> staged (x from ring stage `slot`, dh from its image) ----
    auto phase3 = [&](int slot) {
        const char *sx = smem + slot * STAGE + DD_SUB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const char *bx = sx + kk * 16 * 256, *bg = sDH + kk * 16 * 256;
            bf16x8 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = dd_frag_tr(bx + toffA[i][0], bx + toffA[i][1]);
            const bf16x8 fb = dd_frag_tr(bg + toffB[0], bg + toffB[1]);
            if (wm == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum1 += (float)fb[e];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb, acc1[i], 0, 0, 0);
        }
        f32x4 ax[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        bf16x8 fg[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fg[mi][q] = *reinterpret_cast<const bf16x8 *>(sDH + xoff[mi][q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ax[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[q], fg[mi][q], ax[mi], 0, 0, 0);
        // D: lane holds output columns 16 wave + 4 g + j (j = 0..3) of token 16 mi + li
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) *reinterpret_cast<f32x4 *>(sOut + (16 * mi + li) * FB_OSTR + (16 * wave + 4 * g) * 4) = ax[mi];
    };
    // ---- rows of tile tp: staged dx + dz -> global ----
    auto store_rows = [&](int64_t tp, dd_u32x4 dzp) {
        const int64_t tk = (gfirst + tp * gstep) * DD_TOK + orow;
        if (tk < a.M) {
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(sOut + orow * FB_OSTR + opart * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(sOut + orow * FB_OSTR + opart * 32 + 16);
            const bf16x8 rv = __builtin_bit_cast(bf16x8, dzp);
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = lo[k] + (float)rv[k]; v[4 + k] = hi[k] + (float)rv[4 + k]; }
            Vec8<bf16_t>::template store_sel<B4C_NT(B4C_NT_GEMM)>(a.dX + tk * a.ldo + opart * 8, v);
        }
    };

    // One iteration = two barrier intervals.  Iteration t (0 .. t1; the last one only finishes tile t1 - 1):
    //   interval 1: LayerNorm backward of tile t -> dy image, dz kept; requests tile t + 1's rows; phase 3 of tile t - 1
    //   interval 2: requests tile t + 3's h | x; dx rows of tile t - 1 leave; phase 2 of tile t (dW2, dh image)
    // Vector-memory operations per thread in issue order: ... [3 row loads of t + 1] [2 DMA of t + 3] [1 store of t - 1] ...
    // The only counted wait: tile t's row loads at the top of iteration t -- issued since: 2 DMA + 1 store (none / no store in
    // the first iterations).  Everything older has landed with them: the h | x tiles t and t + 1.
    auto tile = [&](auto ODD, int64_t t) {
        constexpr bool odd_tile = decltype(ODD)::value;  // (dz of a tile waits in one of two register quadruples: the parity is compile time)
        const int slot = (int)(t & 3);
        const bool body = t < t1;
        dd_u32x4 &dz_cur = odd_tile ? dz1 : dz0;
        dd_u32x4 &dz_prev = odd_tile ? dz0 : dz1;
        if (body) {
            if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (t == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            // ---- LayerNorm + dropout backward of this thread's 8 columns of row orow (rowops.hip add_ln_bwd_kernel) ----
            const int64_t tk = (gfirst + t * gstep) * DD_TOK + orow;
            const bool live = tk < a.M;
            const bf16x8 dov = *reinterpret_cast<const bf16x8 *>(sLN + tid * 16), zv = *reinterpret_cast<const bf16x8 *>(sLN + 8192 + tid * 16);
            const f32x4 stp = *reinterpret_cast<const f32x4 *>(sLN + 16384 + tid * 16);
            // which half of the window (see ln_load; a row past M reads row M - 1's window: finite numbers, that is all that matters)
            const unsigned rowc = (unsigned)(live ? tk : a.M - 1);
            const bool odd = (rowc & ~1u) + 2u <= (unsigned)a.M ? (rowc & 1u) != 0 : rowc == (unsigned)a.M - 1u;
            const float mean = odd ? stp[2] : stp[0], rstd = odd ? stp[3] : stp[1];
            float gv[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float go = live ? (float)dov[k] : 0.f;
                xh[k] = ((float)zv[k] - mean) * rstd;
                gv[k] = go * gm[k];
                s1 += gv[k];
                s2 += gv[k] * xh[k];
                pg[k] += go * xh[k];
                pb[k] += go;
            }
            ln_load(t + 1);                             // (every piece of tile t has been read: ln_load waits for the reads itself)
            s1 = group_sum<16>(s1) * (1.0f / 128.0f);
            s2 = group_sum<16>(s2) * (1.0f / 128.0f);
            float o[8];
            bf16x8 dzv, dyv;
#pragma unroll
            for (int k = 0; k < 8; ++k) { o[k] = rstd * (gv[k] - s1 - xh[k] * s2); dzv[k] = (bf16_t)o[k]; }
            if (a.rate > 0.f) {
                const uint32_t km = b4c_keep8(a.seed, (uint64_t)(tk * 128 + opart * 8), thr);
#pragma unroll
                for (int k = 0; k < 8; ++k) dyv[k] = (bf16_t)(((km >> k) & 1u) ? o[k] * inv_keep : 0.f);
            } else {
                dyv = dzv;
            }
            dz_cur = __builtin_bit_cast(dd_u32x4, dzv);
            *reinterpret_cast<bf16x8 *>(sDY + rowoff) = dyv;
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (t > 0) phase3((slot + 3) & 3);
        __syncthreads();
#ifdef DD_EXPERIMENT
        if (body && a.debug >= 10) {
            // is the h | x image of tile t what global memory holds?  (every request of tile t was waited for above, by every wave)
            const int64_t tk = (gfirst + t * gstep) * DD_TOK + orow;
            if (tk < a.M && opart * 8 < a.Fp) {
                const dd_u32x4 l = *reinterpret_cast<const dd_u32x4 *>(smem + slot * STAGE + rowoff);
                const dd_u32x4 gq = *reinterpret_cast<const dd_u32x4 *>(a.H + tk * a.ldh + opart * 8);
                const bool ne = l[0] != gq[0] || l[1] != gq[1] || l[2] != gq[2] || l[3] != gq[3];
                if (ne) {
                    atomicAdd(&g_fb_dbg[(slot * 32 + orow) * 4 + 0], 1u);
                    if (t >= 4) {
                        const int64_t tk4 = (gfirst + (t - 4) * gstep) * DD_TOK + orow;
                        const dd_u32x4 g4 = *reinterpret_cast<const dd_u32x4 *>(a.H + tk4 * a.ldh + opart * 8);
                        if (l[0] == g4[0] && l[1] == g4[1] && l[2] == g4[2] && l[3] == g4[3]) atomicAdd(&g_fb_dbg[(slot * 32 + orow) * 4 + 1], 1u);
                    }
                }
                if (opart == 0) atomicAdd(&g_fb_dbg[(slot * 32 + orow) * 4 + 3], 1u);
            }
            if (tk < a.M) {
                const dd_u32x4 l = *reinterpret_cast<const dd_u32x4 *>(smem + slot * STAGE + DD_SUB + rowoff);
                const dd_u32x4 gq = *reinterpret_cast<const dd_u32x4 *>(a.X + tk * a.ldx + opart * 8);
                if (l[0] != gq[0] || l[1] != gq[1] || l[2] != gq[2] || l[3] != gq[3]) atomicAdd(&g_fb_dbg[(slot * 32 + orow) * 4 + 2], 1u);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#endif
        if (body) fetch(t + 3, (slot + 3) & 3);         // that stage held tile t - 1: phase 3 above was its last reader
        if (t > 0) store_rows(t - 1, dz_prev);
        if (!body) return;
        // ---- phase 2: dW2 += h^T dy, dh = (dy W2^T) o [h > 0] ----
        const char *sh = smem + slot * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const char *bx = sh + kk * 16 * 256, *bg = sDY + kk * 16 * 256;
            bf16x8 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = dd_frag_tr(bx + toffA[i][0], bx + toffA[i][1]);
            const bf16x8 fb = dd_frag_tr(bg + toffB[0], bg + toffB[1]);
            if (wm == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum2 += (float)fb[e];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb, acc2[i], 0, 0, 0);
        }
        f32x4 ax[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        bf16x8 fg[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fg[mi][q] = *reinterpret_cast<const bf16x8 *>(sDY + xoff[mi][q]);
        dd_bf16x4 hv[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) hv[mi] = *reinterpret_cast<const dd_bf16x4 *>(sh + poff[mi]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ax[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[q], fg[mi][q], ax[mi], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            dd_bf16x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (float)hv[mi][j] > 0.f ? (bf16_t)ax[mi][j] : (bf16_t)0.f;
            *reinterpret_cast<dd_bf16x4 *>(sDH + poff[mi]) = w;
        }
        __syncthreads();
    };
    for (int64_t t = 0; t <= t1; t += 2) {
        tile(std::false_type{}, t);
        if (t + 1 <= t1) tile(std::true_type{}, t + 1);
    }

    // ---- this workgroup's partial sums ----
    float *pw = a.part + (int64_t)blockIdx.x * FB_ROWS * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // acc[i]: row (t & 3) + 8 (t >> 2) + 4 hf = feature inside tile 2 wm + i, column r = gradient column inside tile wn
        const int n = wn * 32 + r;
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const f32x4 v2 = {acc2[i][4 * tq], acc2[i][4 * tq + 1], acc2[i][4 * tq + 2], acc2[i][4 * tq + 3]};
            const f32x4 v1 = {acc1[i][4 * tq], acc1[i][4 * tq + 1], acc1[i][4 * tq + 2], acc1[i][4 * tq + 3]};
            *reinterpret_cast<f32x4 *>(pw + (int64_t)n * 128 + (2 * wm + i) * 32 + 8 * tq + 4 * hf) = v2;
            *reinterpret_cast<f32x4 *>(pw + (int64_t)(128 + n) * 128 + (2 * wm + i) * 32 + 8 * tq + 4 * hf) = v1;
        }
    }
    if (wm == 0) {
        // lanes r and r + 32 hold the two token halves of gradient column 32 wn + r
        const float o2 = __shfl_xor(bsum2, 32), o1 = __shfl_xor(bsum1, 32);
        if (hf == 0) {
            pw[256 * 128 + wn * 32 + r] = bsum2 + o2;
            pw[257 * 128 + wn * 32 + r] = bsum1 + o1;
        }
    }
    // dgamma / dbeta: the 32 row groups' sums meet in row order (every request has landed: the last iteration drained the counter)
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);       // [32][256]
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[orow * 256 + opart * 8 + k] = pg[k]; red[orow * 256 + 128 + opart * 8 + k] = pb[k]; }
    __syncthreads();
    if (tid < 256) {
        float s = 0.f;
#pragma unroll 8
        for (int rg = 0; rg < 32; ++rg) s += red[rg * 256 + tid];
        pw[258 * 128 + tid] = s;
    }
}

struct FfnBwdOut {
    float *dW2, *dW1, *db2, *db1, *dgamma, *dbeta;
    int ld2, ld1, F;      // dW2 [F][ld2] (Keras kernel of the second Dense), dW1 [128][ld1], F valid hidden columns
};
// The workgroups' partial sums in a fixed order, as dxdw_reduce_kernel (gemm_dxdw.hip): a block owns 32 consecutive entries, its
// eight 32-lane groups each add every eighth workgroup's partial.
__global__ void __launch_bounds__(256) ffn_bwd_reduce_kernel(const float *__restrict__ part, int nwg, FfnBwdOut out) {
    __shared__ float sh[8][32];
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + c;                 // over FB_ROWS x 128
    float s = 0.f;
#pragma unroll 8
    for (int w = q; w < nwg; w += 8) s += part[(int64_t)w * FB_ROWS * 128 + idx];
    sh[q][c] = s;
    __syncthreads();
    if (q != 0) return;
    s = sh[0][c];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += sh[k][c];
    const int row = idx >> 7, k = idx & 127;
    if (row < 128) {                                     // dW2^T: row = output column, k = hidden column
        if (k < out.F) out.dW2[(int64_t)k * out.ld2 + row] += s;
    } else if (row < 256) {                              // dW1^T: row - 128 = hidden column, k = input feature
        if (row - 128 < out.F) out.dW1[(int64_t)k * out.ld1 + (row - 128)] += s;
    } else if (row == 256) {
        if (out.db2) out.db2[k] += s;
    } else if (row == 257) {
        if (out.db1 && k < out.F) out.db1[k] += s;
    } else if (row == 258) {
        out.dgamma[k] += s;
    } else {
        out.dbeta[k] += s;
    }
}

static int ffn_bwd_grid(int64_t M) {
    const int64_t ntiles = (M + DD_TOK - 1) / DD_TOK;
    return (int)(ntiles < 256 ? ntiles : 256);          // one persistent workgroup per CU; every workgroup has at least one tile
}

extern "C" int64_t b4c_ffn_bwd_workspace_bytes(int64_t M) {
    if (M <= 0) return 0;
    return (int64_t)ffn_bwd_grid(M) * FB_ROWS * 128 * 4;
}

extern "C" int b4c_ffn_bwd(const void *dout, const void *z, const float *stats, const float *gamma, float dropout_rate, uint64_t seed,
                           const void *H, int ldh, const void *X, int ldx, const void *W2c, int ldw2, const void *W1c, int ldw1,
                           int F, int Fp, void *dX, int ldo, float *dW1, int ld_dw1, float *db1, float *dW2, int ld_dw2, float *db2,
                           float *dgamma, float *dbeta, int64_t M, void *workspace, int64_t workspace_bytes, void *stream) {
    B4C_REQUIRE(dout && z && stats && gamma && H && X && W2c && W1c && dX && dW1 && dW2 && dgamma && dbeta && workspace,
                "ffn_bwd: null pointer");
    B4C_REQUIRE(M > 0 && F > 0 && F <= Fp && Fp <= 128 && Fp % 8 == 0, "ffn_bwd: hidden width %d (padded %d): 1..128, padded to a multiple of 8", F, Fp);
    B4C_REQUIRE(ldh >= Fp && ldx >= 128 && ldw2 >= 128 && ldw1 >= Fp && ldo >= 128 && ld_dw1 >= F && ld_dw2 >= 128, "ffn_bwd: shape");
    B4C_REQUIRE(ldh % 8 == 0 && ldx % 8 == 0 && ldw2 % 8 == 0 && ldw1 % 8 == 0 && ldo % 8 == 0 &&
                ((((uintptr_t)dout | (uintptr_t)z | (uintptr_t)H | (uintptr_t)X | (uintptr_t)W2c | (uintptr_t)W1c | (uintptr_t)dX |
                   (uintptr_t)gamma | (uintptr_t)workspace) & 15) == 0) && (((uintptr_t)stats & 7) == 0),
                "ffn_bwd: operands must be 16-byte aligned with pitches % 8 == 0");
    B4C_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f, "ffn_bwd: dropout rate");
    B4C_REQUIRE(M >= 2 && M < ((int64_t)1 << 24), "ffn_bwd: %lld rows (the row chunks are addressed with 32-bit byte offsets: < 16,777,216)", (long long)M);
    B4C_REQUIRE(workspace_bytes >= b4c_ffn_bwd_workspace_bytes(M), "ffn_bwd: workspace too small");
    FfnBwdArgs a = {};
    a.dOut = (const bf16_t *)dout; a.Z = (const bf16_t *)z; a.stats = stats; a.gamma = gamma;
    a.H = (const bf16_t *)H; a.X = (const bf16_t *)X; a.W2c = (const bf16_t *)W2c; a.W1c = (const bf16_t *)W1c; a.dX = (bf16_t *)dX;
    a.part = (float *)workspace;
    a.ldh = ldh; a.ldx = ldx; a.ldw2 = ldw2; a.ldw1 = ldw1; a.ldo = ldo; a.Fp = Fp;
    a.rate = dropout_rate; a.seed = seed; a.M = M;
#ifdef DD_EXPERIMENT
    { static const char *e = getenv("B4C_FFN_DEBUG"); a.debug = e ? atoi(e) : 3; }
#endif
    FfnBwdOut out = {dW2, dW1, db2, db1, dgamma, dbeta, ld_dw2, ld_dw1, F};
    const int grid = ffn_bwd_grid(M);
    const size_t lds = DD_RING * (size_t)2 * DD_SUB + 2 * DD_SUB + DD_TOK * FB_OSTR + 3 * 8192;
    static thread_local bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void *)ffn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    hipStream_t st = (hipStream_t)stream;
    ffn_bwd_kernel<<<grid, 512, lds, st>>>(a);
    ffn_bwd_reduce_kernel<<<FB_ROWS * 128 / 32, 256, 0, st>>>(a.part, grid, out);
    return b4c_check_launch("ffn_bwd");
}
