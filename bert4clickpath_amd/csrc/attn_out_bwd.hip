// Backward of the tail of the attention block of one encoder layer -- LayerNorm, dropout, residual add, output projection -- in ONE
// pass (round 4; the same skeleton as ffn_bwd.hip, one Dense layer instead of two).
//
// The reference's block (transformer.py:158-162 and 204-207: Dense(d_model) on the concatenated heads, dropout, residual add,
// LayerNormalization) came back, in the backward pass, as two kernels:
//     add_ln_bwd      dOut, z, stats -> dz (residual branch), dy (through the dropout mask); dgamma, dbeta      4 [T][128] passes
//     gemm_dxdw<1>    d_o = dy Wo^T, dWo += o^T dy, dbo += colsum(dy)                                           3
// Here dy lives in LDS only: dOut, z, o in; dz (the QKV projection's backward adds it to dx as the residual branch) and d_o out --
// 1,288 B per token against 1,800.
//
// A persistent 512-thread workgroup per CU walks 32-token tiles (tile i of a workgroup = global tile blockIdx.x + i gridDim.x):
//   LDS-DMA     o tile [32][128] into a four-stage ring of XOR-swizzled images (three tiles ahead); dOut / z chunks and the rows'
//               statistics a tile ahead, 16 B per thread at thread * 16 (every request lands in LDS: ffn_bwd.hip says why)
//   interval 1  LayerNorm + dropout backward of tile t in registers (16 threads per row: add_ln_bwd's layout and order) -> dz ->
//               global, dy -> LDS image; the staged d_o rows of tile t - 1 -> bf16 -> global
//   interval 2  dWo += o^T dy (MFMA 32x32x16, transposed LDS reads), d_o = dy Wo^T (MFMA 16x16x32, the wave's 16 columns of Wo
//               resident) -> staged fp32 tile
//   end         per-workgroup partial sums -> scratch; ao_bwd_reduce_kernel adds them in a fixed order (no float atomics).
// d_model = 128, bf16: every other shape keeps the two kernels.
#include <stdlib.h>

#include <type_traits>

#include "dxdw_common.h"

#define AO_OSTR 528                  // bytes per staged d_o row: 128 fp32 + 16
#define AO_ROWS 131                  // partial rows per workgroup: dWo^T (128), dbo, dgamma, dbeta

struct AoBwdArgs {
    const bf16_t *dOut;   // [M][128]  gradient of the block's output
    const bf16_t *Z;      // [M][128]  x + dropout(o Wo + bo): the LayerNorm's input, saved by the forward pass
    const float *stats;   // [M][2]    mean, 1 / std of z's rows
    const float *gamma;   // [128]
    const bf16_t *O;      // [M][ldo_in] the projection's input (the heads' outputs, concatenated)
    const bf16_t *Wc;     // [128][ldw]  row = input feature of the projection, 128 entries: the operand of d_o = dy Wo^T
    bf16_t *dZ;           // [M][128]
    bf16_t *dO;           // [M][ld_do]
    float *part;          // [workgroups][131][128]: dWo^T (row = output column, 128 inputs), dbo, dgamma, dbeta
    int ldo_in, ldw, ld_do;
    float rate;
    uint64_t seed;
    int64_t M;
};

__global__ void __launch_bounds__(512, 1) ao_bwd_kernel(AoBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = DD_SUB;                       // o
    char *sDY = smem + DD_RING * STAGE;                 // [32][128] bf16 image of dy
    char *sOut = sDY + DD_SUB;                          // [32][AO_OSTR] fp32 d_o
    char *sLN = sOut + DD_TOK * AO_OSTR;                // [3][512 threads][16 B]: this thread's dOut chunk, z chunk, statistics pair
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;            // dW tiles: feature half (64 rows), gradient-column quarter (32 columns)
    const int orow = tid >> 4, opart = tid & 15;        // row phases: token row of the tile, 8-column piece
    const int64_t ntile_all = (a.M + DD_TOK - 1) / DD_TOK;
    const int64_t t1 = (ntile_all - blockIdx.x + gridDim.x - 1) / gridDim.x;      // this workgroup's tile count (>= 1)
    const int64_t gstep = gridDim.x, gfirst = blockIdx.x;

    // resident operand of d_o = dy Wo^T: this wave's 16 output columns, B[k = 32 ks + 8 g + j][col = 16 wave + li]
    bf16x8 wf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const bf16x8 *>(a.Wc + (int64_t)(16 * wave + li) * a.ldw + ks * 32 + 8 * g);
    f32x16 acc[2];                                      // dWo^T tiles of this wave: [input-feature tiles 2 wm + i] x output-column tile wn
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[i][t] = 0.f;
    float bsum = 0.f;
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
    const int rowoff = dd_chunk_off(orow, opart);       // the row phase's 16-B chunk inside an image

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    auto fetch = [&](int64_t t, int slot) {
        const int64_t tok0 = (gfirst + t * gstep) * DD_TOK;
        const int64_t M = t < t1 ? a.M : 0;             // past the last tile: zero rows (still LDS writes, still counted)
        dd_dma(a.O, a.ldo_in, 0, tok0, M, lds0 + (unsigned)(slot * STAGE), wave, lane);
    };
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
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(wf[ks]));
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(gm[k]));
    fetch(0, 0);
    fetch(1, 1);
    fetch(2, 2);
    ln_load(0);

    const float inv_keep = a.rate > 0.f ? 1.0f / (1.0f - a.rate) : 1.0f;
    const uint32_t thr = b4c_keep_threshold(a.rate);

    // rows of tile tp: staged d_o -> bf16 -> global
    auto store_rows = [&](int64_t tp) {
        const int64_t tk = (gfirst + tp * gstep) * DD_TOK + orow;
        if (tk < a.M) {
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(sOut + orow * AO_OSTR + opart * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(sOut + orow * AO_OSTR + opart * 32 + 16);
            const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            Vec8<bf16_t>::template store_sel<B4C_NT(B4C_NT_GEMM)>(a.dO + tk * a.ld_do + opart * 8, v);
        }
    };

    // Iteration t (0 .. t1; the last one only sends tile t1 - 1's d_o rows off), two barrier intervals:
    //   interval 1: LayerNorm backward of tile t -> dz -> global, dy image; requests tile t + 1's rows; d_o rows of tile t - 1 leave
    //   interval 2: requests tile t + 3's o; dWo += o^T dy, d_o = dy Wo^T -> staged
    // Vector-memory operations per thread in issue order: ... [3 row requests of t + 1] [dz store of t] [d_o store of t - 1] [1 DMA of
    // t + 3] ...  The only counted wait: tile t's row requests at the top of iteration t -- issued since: two stores and one DMA
    // (one store and one DMA before iteration 1, nothing before iteration 0).  Everything older has landed with them.
    auto tile = [&](int slot, int64_t t) {
        const bool body = t < t1;
        if (body) {
            if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (t == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            // ---- LayerNorm + dropout backward of this thread's 8 columns of row orow (rowops.hip add_ln_bwd_kernel) ----
            const int64_t tk = (gfirst + t * gstep) * DD_TOK + orow;
            const bool live = tk < a.M;
            const bf16x8 dov = *reinterpret_cast<const bf16x8 *>(sLN + tid * 16), zv = *reinterpret_cast<const bf16x8 *>(sLN + 8192 + tid * 16);
            const f32x4 stp = *reinterpret_cast<const f32x4 *>(sLN + 16384 + tid * 16);
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
            bf16x8 dyv;
#pragma unroll
            for (int k = 0; k < 8; ++k) { o[k] = rstd * (gv[k] - s1 - xh[k] * s2); dyv[k] = (bf16_t)o[k]; }
            if (live) Vec8<bf16_t>::template store_sel<B4C_NT(B4C_NT_LNBWD_DZ)>(a.dZ + tk * 128 + opart * 8, o);
            if (a.rate > 0.f) {
                const uint32_t km = b4c_keep8(a.seed, (uint64_t)(tk * 128 + opart * 8), thr);
#pragma unroll
                for (int k = 0; k < 8; ++k) dyv[k] = (bf16_t)(((km >> k) & 1u) ? o[k] * inv_keep : 0.f);
            }
            *reinterpret_cast<bf16x8 *>(sDY + rowoff) = dyv;
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (t > 0) store_rows(t - 1);
        if (!body) return;
        __syncthreads();
        fetch(t + 3, (slot + 3) & 3);                   // that stage held tile t - 1: its last reader was interval 2 of iteration t - 1
        // ---- dWo += o^T dy, d_o = dy Wo^T ----
        const char *so = smem + slot * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const char *bx = so + kk * 16 * 256, *bg = sDY + kk * 16 * 256;
            bf16x8 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = dd_frag_tr(bx + toffA[i][0], bx + toffA[i][1]);
            const bf16x8 fb = dd_frag_tr(bg + toffB[0], bg + toffB[1]);
            if (wm == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum += (float)fb[e];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb, acc[i], 0, 0, 0);
        }
        f32x4 ax[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        bf16x8 fg[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fg[mi][q] = *reinterpret_cast<const bf16x8 *>(sDY + xoff[mi][q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ax[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q], fg[mi][q], ax[mi], 0, 0, 0);
        // D: lane holds output columns 16 wave + 4 g + j (j = 0..3) of token 16 mi + li
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) *reinterpret_cast<f32x4 *>(sOut + (16 * mi + li) * AO_OSTR + (16 * wave + 4 * g) * 4) = ax[mi];
        __syncthreads();
    };
    for (int64_t t = 0; t <= t1; ++t) tile((int)(t & 3), t);

    // ---- this workgroup's partial sums ----
    float *pw = a.part + (int64_t)blockIdx.x * AO_ROWS * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // acc[i]: row (t & 3) + 8 (t >> 2) + 4 hf = input feature inside tile 2 wm + i, column r = output column inside tile wn
        const int n = wn * 32 + r;
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const f32x4 v = {acc[i][4 * tq], acc[i][4 * tq + 1], acc[i][4 * tq + 2], acc[i][4 * tq + 3]};
            *reinterpret_cast<f32x4 *>(pw + (int64_t)n * 128 + (2 * wm + i) * 32 + 8 * tq + 4 * hf) = v;
        }
    }
    if (wm == 0) {
        const float other = __shfl_xor(bsum, 32);       // lanes r and r + 32 hold the two token halves of output column 32 wn + r
        if (hf == 0) pw[128 * 128 + wn * 32 + r] = bsum + other;
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
        pw[129 * 128 + tid] = s;
    }
}

struct AoBwdOut {
    float *dW, *db, *dgamma, *dbeta;
    int ldw;              // dWo [128][ldw] (Keras kernel of the projection)
};
// the workgroups' partial sums in a fixed order, as dxdw_reduce_kernel (gemm_dxdw.hip)
__global__ void __launch_bounds__(256) ao_bwd_reduce_kernel(const float *__restrict__ part, int nwg, AoBwdOut out) {
    __shared__ float sh[8][32];
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + c;                 // over AO_ROWS x 128
    float s = 0.f;
#pragma unroll 8
    for (int w = q; w < nwg; w += 8) s += part[(int64_t)w * AO_ROWS * 128 + idx];
    sh[q][c] = s;
    __syncthreads();
    if (q != 0) return;
    s = sh[0][c];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += sh[k][c];
    const int row = idx >> 7, k = idx & 127;
    if (row < 128) out.dW[(int64_t)k * out.ldw + row] += s;      // dWo^T: row = output column, k = input feature
    else if (row == 128) { if (out.db) out.db[k] += s; }
    else if (row == 129) out.dgamma[k] += s;
    else out.dbeta[k] += s;
}

static int ao_bwd_grid(int64_t M) {
    const int64_t ntiles = (M + DD_TOK - 1) / DD_TOK;
    return (int)(ntiles < 256 ? ntiles : 256);          // one persistent workgroup per CU; every workgroup has at least one tile
}

extern "C" int64_t b4c_attn_out_bwd_workspace_bytes(int64_t M) {
    if (M <= 0) return 0;
    return (int64_t)ao_bwd_grid(M) * AO_ROWS * 128 * 4;
}

extern "C" int b4c_attn_out_bwd(const void *dout, const void *z, const float *stats, const float *gamma, float dropout_rate, uint64_t seed,
                                const void *O, int ldo_in, const void *Wc, int ldw, void *dZ, void *dO, int ld_do,
                                float *dW, int ld_dw, float *db, float *dgamma, float *dbeta, int64_t M,
                                void *workspace, int64_t workspace_bytes, void *stream) {
    B4C_REQUIRE(dout && z && stats && gamma && O && Wc && dZ && dO && dW && dgamma && dbeta && workspace, "attn_out_bwd: null pointer");
    B4C_REQUIRE(M >= 2 && M < ((int64_t)1 << 24), "attn_out_bwd: %lld rows (2 .. 16,777,215: the row chunks are addressed with 32-bit byte offsets)", (long long)M);
    B4C_REQUIRE(ldo_in >= 128 && ldw >= 128 && ld_do >= 128 && ld_dw >= 128, "attn_out_bwd: shape");
    B4C_REQUIRE(ldo_in % 8 == 0 && ldw % 8 == 0 && ld_do % 8 == 0 &&
                ((((uintptr_t)dout | (uintptr_t)z | (uintptr_t)O | (uintptr_t)Wc | (uintptr_t)dZ | (uintptr_t)dO | (uintptr_t)gamma |
                   (uintptr_t)workspace) & 15) == 0) && (((uintptr_t)stats & 7) == 0),
                "attn_out_bwd: operands must be 16-byte aligned with pitches % 8 == 0");
    B4C_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f, "attn_out_bwd: dropout rate");
    B4C_REQUIRE(workspace_bytes >= b4c_attn_out_bwd_workspace_bytes(M), "attn_out_bwd: workspace too small");
    AoBwdArgs a = {};
    a.dOut = (const bf16_t *)dout; a.Z = (const bf16_t *)z; a.stats = stats; a.gamma = gamma;
    a.O = (const bf16_t *)O; a.Wc = (const bf16_t *)Wc; a.dZ = (bf16_t *)dZ; a.dO = (bf16_t *)dO;
    a.part = (float *)workspace;
    a.ldo_in = ldo_in; a.ldw = ldw; a.ld_do = ld_do;
    a.rate = dropout_rate; a.seed = seed; a.M = M;
    AoBwdOut out = {dW, db, dgamma, dbeta, ld_dw};
    const int grid = ao_bwd_grid(M);
    const size_t lds = DD_RING * (size_t)DD_SUB + DD_SUB + DD_TOK * AO_OSTR + 3 * 8192;
    static thread_local bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void *)ao_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    hipStream_t st = (hipStream_t)stream;
    ao_bwd_kernel<<<grid, 512, lds, st>>>(a);
    ao_bwd_reduce_kernel<<<(AO_ROWS * 128 + 31) / 32, 256, 0, st>>>(a.part, grid, out);
    return b4c_check_launch("attn_out_bwd");
}
