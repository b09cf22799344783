// Backward of one Dense layer of the encoder in ONE pass over its gradient tensor (round 4).
//
// The reference's Dense call sites (transformer.py:112-116, 158, 163-167) come back, in the backward pass, as two GEMMs that
// both read the layer's output gradient G [T][N]:   dX = G W^T (+ residual)   and   dW = X^T G,  db = colsum(G).
// As two kernels (b4c_gemm_nt + b4c_gemm_tn) G crosses HBM twice; for the fused Q | K | V projection that second read is
// 3 of the layer's ~36 [T][128] passes.  Here a persistent workgroup walks its share of the 32-token tiles once:
//
//   LDS-DMA     X tile [32][128] and G tile [32][128 NG] (NG = 3: q | k | v) into a four-stage ring of XOR-swizzled images
//   dW          X^T G for the tile: 128 x 128 NG accumulators stay in registers over ALL the workgroup's tiles
//               (8 waves x (2 x 3) MFMA 32x32x16 tiles = 96 registers per lane), fragments through ds_read_b64_tr_b16
//   dX          G W^T: wave w owns 16 output columns, its W fragments (128 NG x 16: 48 registers) resident for the whole
//               kernel, MFMA 16x16x32 with the G rows straight from the LDS image
//   epilogue    dX tile -> bf16 -> LDS transpose -> 16-B row chunks (+ residual: its chunk arrives by LDS-DMA a tile ahead; added in
//               fp32) -> global, a tile late (the staged tile is double-buffered: ONE barrier per tile)
//   end         the workgroup's dW / db partial -> scratch; dxdw_reduce_kernel adds the partials in workgroup order
//               (deterministic) into the Keras-layout gradient tensors, column segments (q | k | v) apart.
//
// HBM-bound like the kernels it replaces (AI ~ 110 FLOP / B): per token 128 (1 + NG) x 2 B in, 256 B (+ 256 B residual) for dX.
// d_model = 128, bf16 only (C2); every other shape keeps the two-kernel route.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#include "dxdw_common.h"

#ifdef DD_STAMPS
__device__ unsigned long long g_dd_stamps[256 * 8 * 8];
extern "C" int b4c_debug_dd_stamps(void *dst, size_t nbytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dd_stamps), nbytes < sizeof(g_dd_stamps) ? nbytes : sizeof(g_dd_stamps));
}
#define DSTAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[k] += t_ - t0_; t0_ = t_; } while (0)
#else
#define DSTAMP(k) do { } while (0)
#endif

struct DxDwArgs {
    const bf16_t *X;      // [M][ldx]   the layer's input (128 columns)
    const bf16_t *G;      // [M][ldg]   gradient of its output (128 NG columns)
    const bf16_t *Wc;     // [128][ldw] rows = input features, 128 NG columns (the dX operand: K-contiguous)
    const bf16_t *Res;    // [M][ldr] or NULL: added to dX (the residual branch's gradient)
    bf16_t *dX;           // [M][ldo]
#ifdef DD_EXPERIMENT
    int debug;            // scratch builds only: 1 = no MFMA work (memory alone), 2 = no DMA past the first tiles (compute alone), 3 = no dX stores
#endif
    float *part;          // [workgroups][128 NG + NG][128]: dW^T partials (row n = gradient column, 128 input features), then the 128 NG db sums
    int ldx, ldg, ldw, ldr, ldo;
    int64_t M;
    int64_t tiles_per_wg;
};

template <int NG>
__global__ void __launch_bounds__(512, 1) gemm_dxdw_kernel(DxDwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (1 + NG) * DD_SUB;
    constexpr int NK = 4 * NG;                          // 32-wide k-steps of dX (over the gradient's columns)
    char *sOut = smem + DD_RING * STAGE;                // [2][32][DD_OSTR]: the staged dX tile, double-buffered
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;            // dW: input-feature half (64 rows), gradient-column quarter (32 NG columns)
    // The workgroup's tiles are blockIdx.x, blockIdx.x + gridDim.x, ...: at any moment the workgroups read one contiguous stretch
    // of the tensors (every HBM channel busy).  Contiguous chunks per workgroup put 256 streams 1.4 MB apart -- a multiple of
    // 32 KB: a handful of channels at a time, 3.2 TB/s with the arithmetic switched off.
    // Local tile numbers below: tile i of this workgroup is global tile blockIdx.x + i * gridDim.x.
    const int64_t ntile_all = (a.M + DD_TOK - 1) / DD_TOK;
    const int64_t t0 = 0;
    const int64_t t1 = (ntile_all - blockIdx.x + gridDim.x - 1) / gridDim.x;      // this workgroup's tile count
    const int64_t gstep = gridDim.x, gfirst = blockIdx.x;

    // dX: this wave's 16 output columns of W^T, resident: B[k = 32 ks + 8 g + j][col = 16 wave + li]
    bf16x8 wcf[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
        wcf[ks] = *reinterpret_cast<const bf16x8 *>(a.Wc + (int64_t)(16 * wave + li) * a.ldw + ks * 32 + 8 * g);

    constexpr int NT = NG;                              // gradient 32-column tiles per wave: 4 NG / 4
    f32x16 acc[2][NT];                                  // [input-feature tiles of this wave][gradient-column tiles of this wave]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    float bsum[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bsum[j] = 0.f;

    // transposed-fragment offsets of this wave's tiles (computed, not looked up: an array indexed by the wave's coordinates
    // would live in scratch memory)
    int toffA[2][2], toffB[NT][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { toffA[i][0] = dd_tr_off(hf, li, g, 2 * wm + i, 0); toffA[i][1] = dd_tr_off(hf, li, g, 2 * wm + i, 1); }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nt = NT * wn + j;                     // 32-column tile of the gradient: sub-tile nt >> 2, dt = nt & 3
        toffB[j][0] = (nt >> 2) * DD_SUB + dd_tr_off(hf, li, g, nt & 3, 0);
        toffB[j][1] = (nt >> 2) * DD_SUB + dd_tr_off(hf, li, g, nt & 3, 1);
    }
    // dX A operand (16x16x32): row 16 mi + li, chunk 4 (ks & 3) + g of sub-tile ks >> 2
    constexpr int NMI = DD_TOK / 16;
    int xoff[NMI][4];
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) xoff[mi][q] = dd_chunk_off(16 * mi + li, 4 * q + g);

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    auto fetch = [&](int64_t t, int slot) {
        const unsigned st = lds0 + (unsigned)(slot * STAGE);
        const int64_t tok0 = (gfirst + t * gstep) * DD_TOK;
        const int64_t M = t < t1 ? a.M : 0;
        dd_dma(a.X, a.ldx, 0, tok0, M, st, wave, lane);
#pragma unroll
        for (int s = 0; s < NG; ++s) dd_dma(a.G, a.ldg, 128 * s, tok0, M, st + (1 + s) * DD_SUB, wave, lane);
    };
    if (t0 >= t1) return;                               // (a workgroup without tiles: nothing to add, its partial is never read)
    fetch(t0, 0);
    fetch(t0 + 1, 1);
    fetch(t0 + 2, 2);
    // (1 + NG) DMA instructions per thread and tile; the first tile must have landed: all but the two younger ones' are waited for
    if (NG == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (NG == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    // ---- pipeline of one tile: [DMA of tile t + 3] [everything older landed?] [dX rows of tile t - 1: staged tile + residual ->
    // global] [residual chunk of this tile requested] [dW, dX of tile t -> staged] [barrier].  ONE barrier per tile: the staged dX tile
    // is double-buffered and leaves a tile late.
    // The vector-memory counter retires in issue order (loads, stores and LDS-DMA alike: scratch/vmcnt_order.hip) and the compiler
    // knows nothing of the inline-assembly requests, so every wait is counted by hand: s_waitcnt vmcnt(N) with N = the operations
    // issued AFTER the one waited for.
    const int orow = tid >> 4, opart = tid & 15;
    constexpr int ND = 1 + NG;                          // DMA requests per thread and tile
#define DD_WAIT_VM(n) do { switch (n) { case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break; case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break; \
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break; case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break; \
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break; case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break; \
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break; } } while (0)
    // The residual chunk of a tile lands in LDS too (LDS-DMA, 16 B per thread at thread * 16; the thread reads back its own piece, so
    // the issuing wave's counted wait is all the ordering it needs).  It landed in REGISTERS first (an inline-assembly global load
    // a tile ahead): the compiler knows nothing of a load in flight and copies such a register whenever it likes -- a chunk returned
    // by value was copied to wherever the unrolled tiles' values merge, a workgroup's last tile then left with stale registers; in
    // csrc/ffn_bwd.hip hipcc 7.2 hoisted a copy above the counted s_waitcnt, and naming the registers on the wait made it copy all
    // of them in front of it.  Nothing can copy LDS.
    char *sRes = sOut + 2 * DD_TOK * DD_OSTR;           // [512 threads][16 B]
    dd_u32x4 ds_res = {0u, 0u, 0u, 0x00020000u};
    if (a.Res) {
        const uint64_t br = (uint64_t)a.Res;
        ds_res[0] = __builtin_amdgcn_readfirstlane((unsigned)br); ds_res[1] = __builtin_amdgcn_readfirstlane((unsigned)(br >> 32) & 0xFFFFu);
        ds_res[2] = __builtin_amdgcn_readfirstlane((unsigned)((a.M - 1) * a.ldr * 2 + 256));
    }
    const unsigned res_lds = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(sRes - smem) + (unsigned)(wave * 1024));
    auto res_load = [&](int64_t t) {
        if (a.Res) {                                    // (workgroup-uniform; rows past M read the last row: never stored)
            const int64_t tk = (gfirst + t * gstep) * DD_TOK + orow;
            const unsigned row = (unsigned)(tk < a.M ? tk : a.M - 1);
            const unsigned vo = row * (unsigned)(a.ldr * 2) + (unsigned)opart * 16u;      // (< 4 GB: checked by the host)
            // lgkmcnt(0): this thread's read of the previous chunk is done before anything can land on it
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(res_lds), "v"(vo), "s"(ds_res) : "m0", "memory");
        }
    };
    // rows of tile tp (staged in so, residual chunk res) -> global
    auto store_rows = [&](int64_t tp, const char *so) {
        const int64_t tk = (gfirst + tp * gstep) * DD_TOK + orow;
#ifdef DD_EXPERIMENT
        if (a.debug == 3) return;
#endif
        if (tk < a.M) {
            const dd_u32x4 w4 = *reinterpret_cast<const dd_u32x4 *>(so + orow * DD_OSTR + opart * 16);
            const bf16x8 cv = __builtin_bit_cast(bf16x8, w4);
            const bf16x8 rv = a.Res ? *reinterpret_cast<const bf16x8 *>(sRes + tid * 16) : cv;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (float)cv[k] + (a.Res ? (float)rv[k] : 0.f);
            Vec8<bf16_t>::template store_sel<B4C_NT(B4C_NT_GEMM)>(a.dX + tk * a.ldo + opart * 8, v);
        }
    };
#ifdef DD_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime();
#endif

    auto tile = [&](auto SLOT, int64_t t) {
        constexpr int slot = decltype(SLOT)::value;
        const char *sx = smem + slot * STAGE;
        const char *sg = sx + DD_SUB;
        // Vector-memory operations per thread and tile, in issue order: [ND DMA of tile t + 3] [store of tile t - 1] [residual chunk
        // of tile t].  One counted wait, behind this tile's DMA: everything older than them is done -- tile t - 1's residual chunk
        // and the x | g images of tiles t + 1 and t + 2 (the barrier at the end of the tile makes that true for every wave).
#ifdef DD_EXPERIMENT
        if (a.debug == 2) { for (int i_ = 0; i_ < ND; ++i_) dd_dma(a.X, a.ldx, 0, 0, 0, lds0, wave, lane); } else
#endif
        fetch(t + 3, (slot + 3) % DD_RING);             // that stage held tile t - 1: every wave is past it (the barrier below)
        if (t > t0) {
            if (ND == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (ND == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            store_rows(t - 1, sOut + ((slot + 1) & 1) * (DD_TOK * DD_OSTR));      // the dX rows of tile t - 1 leave now
        }
        res_load(t);
        DSTAMP(0);
        // ---- dW += X^T G: two 16-token steps ----
#ifdef DD_EXPERIMENT
        if (a.debug != 1)
#endif
#pragma unroll
        for (int kk = 0; kk < DD_TOK / 16; ++kk) {
            const char *bx = sx + kk * 16 * 256;
            bf16x8 fa[2], fb[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = dd_frag_tr(bx + toffA[i][0], bx + toffA[i][1]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const char *bg = sg + kk * 16 * 256;
                fb[j] = dd_frag_tr(bg + toffB[j][0], bg + toffB[j][1]);
                if (wm == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[j] += (float)fb[j][e];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        DSTAMP(1);
        // ---- dX^T = W G^T: this wave's 16 output columns x 32 tokens (roles swapped: a lane ends up with 4 consecutive output
        // columns of one token, an 8-byte piece of the staged row).  The eight G fragments of a 128-column block are requested
        // together, in front of their eight MFMAs: one read per MFMA exposes an LDS round trip 24 times per tile ----
        f32x4 ax[NMI];
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) ax[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef DD_EXPERIMENT
        if (a.debug != 1)
#endif
#pragma unroll
        for (int sb = 0; sb < NG; ++sb) {
            bf16x8 fg[NMI][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi) fg[mi][q] = *reinterpret_cast<const bf16x8 *>(sg + sb * DD_SUB + xoff[mi][q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < NMI; ++mi)
                    ax[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcf[4 * sb + q], fg[mi][q], ax[mi], 0, 0, 0);
        }
        DSTAMP(2);
        // D: lane holds output columns 16 wave + 4 g + j (j = 0..3) of token 16 mi + li
        char *so = sOut + (slot & 1) * (DD_TOK * DD_OSTR);
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            dd_bf16x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (bf16_t)ax[mi][j];
            *reinterpret_cast<dd_bf16x4 *>(so + (16 * mi + li) * DD_OSTR + (16 * wave + 4 * g) * 2) = w;
        }
        // tile t + 1 must have landed before the next tile reads it: the wait at the top of this tile saw to that, except in the
        // workgroup's first tile -- issued after tile t0 + 1's DMA: the DMA of tiles t0 + 2 and t0 + 3 and this tile's residual chunk
        DSTAMP(3);
        if (t == t0) {
            const int n = 2 * ND + (a.Res ? 1 : 0);
            DD_WAIT_VM(n);
        }
        DSTAMP(4);
        __syncthreads();                                // next stage landed for every wave; this stage and the older staged tile free again
        DSTAMP(5);
    };
    for (int64_t t = t0; t < t1; t += DD_RING) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < t1) tile(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < t1) tile(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < t1) tile(std::integral_constant<int, 3>{}, t + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the last residual chunk; requests past the last tile (zero rows, still LDS writes)
    store_rows(t1 - 1, sOut + (((t1 - 1 - t0) & 1) != 0 ? DD_TOK * DD_OSTR : 0));
#undef DD_WAIT_VM
#ifdef DD_STAMPS
    if (lane == 0) for (int k = 0; k < 8; ++k) g_dd_stamps[(blockIdx.x * 8 + wave) * 8 + k] = st_[k];
#endif

    // ---- this workgroup's partial sums: part[wg][n][k] (n = gradient column, k = input feature), then the db block ----
    float *pw = a.part + (int64_t)blockIdx.x * (128 * NG + NG) * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // acc[i][j]: row (t & 3) + 8 (t >> 2) + 4 hf = input feature inside tile 2 wm + i, column r = gradient column inside tile
            const int n = (NT * wn + j) * 32 + r;
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const f32x4 v = {acc[i][j][4 * tq], acc[i][j][4 * tq + 1], acc[i][j][4 * tq + 2], acc[i][j][4 * tq + 3]};
                *reinterpret_cast<f32x4 *>(pw + (int64_t)n * 128 + (2 * wm + i) * 32 + 8 * tq + 4 * hf) = v;
            }
        }
    if (wm == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // lanes r and r + 32 hold the two token halves of gradient column (NT wn + j) 32 + r
            const float other = __shfl_xor(bsum[j], 32);
            if (hf == 0) pw[(int64_t)128 * NG * 128 + (NT * wn + j) * 32 + r] = bsum[j] + other;
        }
    }
}

// dW_seg[k][n] += sum over workgroups (in workgroup order) of part[wg][seg 128 + n][k];  db_seg[n] += sum of the db rows
struct DxDwOut {
    float *dW[3];
    float *db[3];
    int ldw;
};
// The workgroups' partial sums meet here in a fixed order (bit-repeatable for a given grid).  A block owns 32 consecutive entries;
// its eight 32-lane groups each add every eighth workgroup's partial (128-B rows, 32 independent loads per thread), the groups'
// sums meet in group order.  (One thread per entry walking all 256 partials took 63 - 68 us -- as long as half the main kernel.)
template <int NG>
__global__ void __launch_bounds__(256) dxdw_reduce_kernel(const float *__restrict__ part, int nwg, DxDwOut out) {
    constexpr int ROWS = 128 * NG + NG;
    __shared__ float sh[8][32];
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + c;                 // over ROWS x 128 (a multiple of 32)
    float s = 0.f;
#pragma unroll 8
    for (int w = q; w < nwg; w += 8) s += part[(int64_t)w * ROWS * 128 + idx];
    sh[q][c] = s;
    __syncthreads();
    if (q != 0) return;
    s = sh[0][c];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += sh[k][c];
    const int row = idx >> 7, k = idx & 127;
    if (row < 128 * NG) {
        out.dW[row >> 7][(int64_t)k * out.ldw + (row & 127)] += s;
    } else {
        const int n = (row - 128 * NG) * 128 + k;
        if (out.db[n >> 7]) out.db[n >> 7][n & 127] += s;
    }
}

static int dxdw_grid(int64_t M, int64_t *tiles_per_wg) {
    const int64_t ntiles = (M + DD_TOK - 1) / DD_TOK;
    *tiles_per_wg = (ntiles + 255) / 256;
    return (int)(ntiles < 256 ? ntiles : 256);          // one persistent workgroup per CU; every workgroup has at least one tile
}

extern "C" int64_t b4c_gemm_dxdw_workspace_bytes(int64_t M, int n_seg) {
    if (M <= 0 || n_seg < 1 || n_seg > 3) return 0;
    int64_t per;
    const int grid = dxdw_grid(M, &per);
    return (int64_t)grid * (128 * n_seg + n_seg) * 128 * 4;      // dW^T rows + n_seg rows for the 128 n_seg db sums, per workgroup
}

template <int NG>
static int dxdw_launch(DxDwArgs a, DxDwOut out, hipStream_t st) {
    int64_t per;
    const int grid = dxdw_grid(a.M, &per);
    a.tiles_per_wg = per;
    const size_t lds = DD_RING * (size_t)(1 + NG) * DD_SUB + 2 * DD_TOK * DD_OSTR + 8192;
    static thread_local bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void *)gemm_dxdw_kernel<NG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    gemm_dxdw_kernel<NG><<<grid, 512, lds, st>>>(a);
    dxdw_reduce_kernel<NG><<<(128 * NG + NG) * 128 / 32, 256, 0, st>>>(a.part, grid, out);
    return b4c_check_launch("gemm_dxdw");
}

extern "C" int b4c_gemm_dxdw(const void *X, int ldx, const void *G, int ldg, const void *Wc, int ldw, const void *residual, int ldr,
                             void *dX, int ldo, int n_seg, float *const *h_dW, float *const *h_db, int ld_dw, int64_t M,
                             void *workspace, int64_t workspace_bytes, void *stream) {
    B4C_REQUIRE(X && G && Wc && dX && h_dW && workspace, "gemm_dxdw: null pointer");
    B4C_REQUIRE(n_seg >= 1 && n_seg <= 3, "gemm_dxdw: %d column segments (1 to 3)", n_seg);
    B4C_REQUIRE(M > 0 && ldx >= 128 && ldg >= 128 * n_seg && ldw >= 128 * n_seg && ldo >= 128 && (!residual || ldr >= 128), "gemm_dxdw: shape");
    B4C_REQUIRE(ldx % 8 == 0 && ldg % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0 && ldr % 8 == 0 &&
                ((((uintptr_t)X | (uintptr_t)G | (uintptr_t)Wc | (uintptr_t)dX | (uintptr_t)residual | (uintptr_t)workspace) & 15) == 0),
                "gemm_dxdw: operands must be 16-byte aligned with pitches % 8 == 0");
    B4C_REQUIRE(workspace_bytes >= b4c_gemm_dxdw_workspace_bytes(M, n_seg), "gemm_dxdw: workspace too small");
    B4C_REQUIRE(!residual || M * (int64_t)ldr * 2 < ((int64_t)1 << 32), "gemm_dxdw: residual of %lld rows x %d (its chunks are addressed with 32-bit byte offsets)", (long long)M, ldr);
    DxDwArgs a = {};
    a.X = (const bf16_t *)X; a.G = (const bf16_t *)G; a.Wc = (const bf16_t *)Wc; a.Res = (const bf16_t *)residual; a.dX = (bf16_t *)dX;
    a.part = (float *)workspace;
    a.ldx = ldx; a.ldg = ldg; a.ldw = ldw; a.ldr = ldr; a.ldo = ldo; a.M = M;
#ifdef DD_EXPERIMENT
    { static const char *e = getenv("B4C_DXDW_DEBUG"); a.debug = e ? atoi(e) : 0; }
#endif
    DxDwOut out = {};
    for (int s = 0; s < n_seg; ++s) {
        B4C_REQUIRE(h_dW[s], "gemm_dxdw: null dW segment %d", s);
        out.dW[s] = h_dW[s];
        out.db[s] = h_db ? h_db[s] : nullptr;
    }
    out.ldw = ld_dw;
    return n_seg == 3 ? dxdw_launch<3>(a, out, (hipStream_t)stream) : n_seg == 2 ? dxdw_launch<2>(a, out, (hipStream_t)stream) :
                        dxdw_launch<1>(a, out, (hipStream_t)stream);
}
