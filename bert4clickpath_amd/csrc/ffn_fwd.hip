// Forward of the position-wise feed-forward block of one encoder layer in ONE pass (round 4):
//     out = LayerNorm(x + dropout(relu(x W1 + b1) W2 + b2))                                (transformer.py:154-170)
// As two kernels (b4c_gemm_nt with ReLU, b4c_gemm_nt_add_ln) the hidden activation h [T][Fp] is written and read back and x is read
// twice: 658 MB per launch at the C2 token count.  Here h leaves once (the backward pass needs it) and is consumed from LDS, x is
// read once: x in; h, z, out, stats out -- 446 MB.
//
// A persistent 512-thread workgroup walks 32-token tiles (tile i of a workgroup = global tile blockIdx.x + i gridDim.x), the skeleton
// of ffn_bwd.hip / attn_out_bwd.hip:
//   LDS-DMA     x tile [32][128] into a four-stage ring of XOR-swizzled images, three tiles ahead
//   interval 1  h = relu(x W1^T + b1) of tile t (MFMA 16x16x32: the wave's 16 hidden columns of W1 resident) -> bf16 LDS image  |
//               rows of tile t - 1: y + b2 from the staged tile, dropout, z = x + drop(y), LayerNorm with 16-lane sums
//               (add_ln_fwd's layout and order) -> z, out, stats -> global
//   interval 2  y = h W2^T of tile t (the wave's 16 output columns of W2 resident) -> fp32 staged tile  |  h rows of tile t -> global
// No register ever waits for a load in flight (every request is an LDS-DMA); the only counted wait is for tile t's x image.
// d_model = 128, dff <= 128, bf16: every other shape keeps the two kernels.
#include <stdlib.h>

#include "dxdw_common.h"

#define FF_OSTR 528                  // bytes per staged y row: 128 fp32 + 16

struct FfnFwdArgs {
    const bf16_t *X;      // [M][ldx]
    const bf16_t *W1t;    // [Fp][ldw1]  row = hidden column, 128 entries (the forward operand of the first Dense)
    const float *b1;      // [Fp]
    const bf16_t *W2t;    // [128][ldw2] row = output column, Fp entries
    const float *b2, *gamma, *beta;   // [128]
    bf16_t *H;            // [M][ldh]   relu(x W1 + b1), Fp columns
    bf16_t *Z;            // [M][128] or NULL (inference)
    bf16_t *Out;          // [M][128]
    float *stats;         // [M][2] or NULL
    int ldx, ldw1, ldw2, ldh, Fp;
    float eps, rate;
    uint64_t seed;
    int64_t M;
};

__global__ void __launch_bounds__(512, 4) ffn_fwd_kernel(FfnFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = DD_SUB;                       // x
    char *sH = smem + DD_RING * STAGE;                  // [32][128] bf16 image of h
    char *sOut = sH + DD_SUB;                           // [32][FF_OSTR] fp32 y = h W2^T
    float *sPar = reinterpret_cast<float *>(sOut + DD_TOK * FF_OSTR);   // [3][128]: b2, gamma, beta (read per tile: registers are short at two workgroups per CU)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int orow = tid >> 4, opart = tid & 15;        // row phase: token row of the tile, 8-column piece
    const int64_t ntile_all = (a.M + DD_TOK - 1) / DD_TOK;
    const int64_t t1 = (ntile_all - blockIdx.x + gridDim.x - 1) / gridDim.x;      // this workgroup's tile count (>= 1)
    const int64_t gstep = gridDim.x, gfirst = blockIdx.x;

    // resident operands: this wave's 16 output columns of each layer, B[k = 32 ks + 8 g + j][col = 16 wave + li]
    bf16x8 w1f[4], w2f[4];
    float bias1[4];
    {
        const bf16x8 zero = __builtin_bit_cast(bf16x8, (dd_u32x4){0u, 0u, 0u, 0u});
        const int col = 16 * wave + li;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = ks * 32 + 8 * g;
            w1f[ks] = col < a.Fp ? *reinterpret_cast<const bf16x8 *>(a.W1t + (int64_t)col * a.ldw1 + k) : zero;
            w2f[ks] = k < a.Fp ? *reinterpret_cast<const bf16x8 *>(a.W2t + (int64_t)col * a.ldw2 + k) : zero;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = 16 * wave + 4 * g + j; bias1[j] = c < a.Fp ? a.b1[c] : 0.f; }
    }
    if (tid < 128) { sPar[tid] = a.b2[tid]; sPar[128 + tid] = a.gamma[tid]; sPar[256 + tid] = a.beta[tid]; }
    // the compiler's own loads are consumed HERE: its wait for them would otherwise sit at their first use inside the tile loop, where
    // it counts none of the requests below and drains them every iteration
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { asm volatile("" : "+v"(w1f[ks])); asm volatile("" : "+v"(w2f[ks])); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (and the parameter loads above)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bias1[j]));

    int xoff[2][4], poff[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xoff[mi][q] = dd_chunk_off(16 * mi + li, 4 * q + g);
        // the lane's 4 consecutive columns 16 wave + 4 g .. + 3 of token 16 mi + li inside an image (8-B piece of a 16-B chunk)
        poff[mi] = dd_chunk_off(16 * mi + li, (16 * wave + 4 * g) >> 3) + ((4 * g) & 7) * 2;
    }
    const int rowoff = dd_chunk_off(orow, opart);

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    auto fetch = [&](int64_t t) {
        const int64_t tok0 = (gfirst + t * gstep) * DD_TOK;
        const int64_t M = t < t1 ? a.M : 0;             // past the last tile: zero rows (still an LDS write, still counted)
        dd_dma(a.X, a.ldx, 0, tok0, M, lds0 + (unsigned)((int)(t & 3) * STAGE), wave, lane);
    };
    fetch(0);
    fetch(1);
    fetch(2);
    const float inv_keep = a.rate > 0.f ? 1.0f / (1.0f - a.rate) : 1.0f;
    const uint32_t thr = b4c_keep_threshold(a.rate);

    // rows of tile tp: staged y, the x image of its ring stage -> z, out, stats
    auto row_phase = [&](int64_t tp) {
        const int64_t tk = (gfirst + tp * gstep) * DD_TOK + orow;
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(sOut + orow * FF_OSTR + opart * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(sOut + orow * FF_OSTR + opart * 32 + 16);
        const bf16x8 xv = *reinterpret_cast<const bf16x8 *>(smem + (int)(tp & 3) * STAGE + rowoff);
        const uint32_t km = a.rate > 0.f ? b4c_keep8(a.seed, (uint64_t)(tk * 128 + opart * 8), thr) : 0xFFu;
        float v[8], sum = 0.f, bias2[8];
        Vec8<float>::load(sPar + 8 * opart, bias2);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float yy = (k < 4 ? lo[k] : hi[k - 4]) + bias2[k];
            if (a.rate > 0.f) yy = ((km >> k) & 1u) ? yy * inv_keep : 0.f;
            v[k] = (float)xv[k] + yy;
            sum += v[k];
        }
        const float mean = group_sum<16>(sum) * (1.0f / 128.0f);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float dlt = v[k] - mean; sq += dlt * dlt; }
        const float var = group_sum<16>(sq) * (1.0f / 128.0f);
        const float rstd = 1.0f / sqrtf(var + a.eps);
        if (tk < a.M) {
            float o[8], gm[8], bt[8];
            Vec8<float>::load(sPar + 128 + 8 * opart, gm);
            Vec8<float>::load(sPar + 256 + 8 * opart, bt);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (v[k] - mean) * rstd * gm[k] + bt[k];
            if (a.Z) Vec8<bf16_t>::template store_sel<B4C_NT(B4C_NT_GEMMLN_Z)>(a.Z + tk * 128 + opart * 8, v);
            Vec8<bf16_t>::template store_sel<B4C_NT(B4C_NT_GEMMLN_OUT)>(a.Out + tk * 128 + opart * 8, o);
            if (a.stats && opart == 0) { a.stats[tk * 2] = mean; a.stats[tk * 2 + 1] = rstd; }
        }
    };

    // Iteration t (0 .. t1; the last one only finishes tile t1 - 1), two barrier intervals.  Vector-memory operations per thread in
    // issue order: [z, out, stats stores of tile t - 1] | [DMA of x(t + 3)] [h store of tile t].  The only wait: x(t) landed -- its DMA
    // went out in iteration t - 3; issued since, at least: the h store of that iteration and [out store, DMA, h store] of the two
    // iterations between (inference: no z, no stats) = 7 operations, 11 when training.  `vmcnt(6)` is safe in every mode (a smaller
    // count only waits for more); the first three iterations wait for all but one.
    for (int64_t t = 0; t <= t1; ++t) {
        const bool body = t < t1;
        const int slot = (int)(t & 3);
        if (body) { if (t < 3) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        __syncthreads();                                // x(t) landed for every wave; the staged y of tile t - 1 complete, its h image read
        if (body) {
            // ---- h = relu(x W1^T + b1): this wave's 16 hidden columns x 32 tokens ----
            const char *sx = smem + slot * STAGE;
            f32x4 ax[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            bf16x8 fg[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) fg[mi][q] = *reinterpret_cast<const bf16x8 *>(sx + xoff[mi][q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) ax[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[q], fg[mi][q], ax[mi], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                dd_bf16x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (bf16_t)fmaxf(ax[mi][j] + bias1[j], 0.f);
                *reinterpret_cast<dd_bf16x4 *>(sH + poff[mi]) = w;
            }
        }
        if (t > 0) row_phase(t - 1);
        if (!body) break;
        __syncthreads();                                // the h image of tile t complete; the staged y and the x image of tile t - 1 read
        fetch(t + 3);                                   // into the stage that held tile t - 1
        {
            // ---- y = h W2^T: this wave's 16 output columns x 32 tokens ----
            f32x4 ax[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            bf16x8 fg[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) fg[mi][q] = *reinterpret_cast<const bf16x8 *>(sH + xoff[mi][q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) ax[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[q], fg[mi][q], ax[mi], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) *reinterpret_cast<f32x4 *>(sOut + (16 * mi + li) * FF_OSTR + (16 * wave + 4 * g) * 4) = ax[mi];
            // h rows of tile t -> global (16-B chunks of the image)
            const int64_t tk = (gfirst + t * gstep) * DD_TOK + orow;
            if (tk < a.M && opart * 8 < a.Fp) {
                const dd_u32x4 hv = *reinterpret_cast<const dd_u32x4 *>(sH + rowoff);
                *reinterpret_cast<dd_u32x4 *>(a.H + tk * a.ldh + opart * 8) = hv;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // requests past the last tile (zero rows) are still LDS writes: none may outlive the workgroup
}

extern "C" int b4c_ffn_fwd(const void *X, int ldx, const void *W1t, int ldw1, const float *b1, const void *W2t, int ldw2, const float *b2,
                           const float *gamma, const float *beta, int F, int Fp, void *H, int ldh, void *Z, void *Out, float *stats,
                           int64_t M, float eps, float dropout_rate, uint64_t seed, void *stream) {
    B4C_REQUIRE(X && W1t && b1 && W2t && b2 && gamma && beta && H && Out, "ffn_fwd: null pointer");
    B4C_REQUIRE(M > 0 && F > 0 && F <= Fp && Fp <= 128 && Fp % 8 == 0, "ffn_fwd: hidden width %d (padded %d): 1..128, padded to a multiple of 8", F, Fp);
    B4C_REQUIRE(ldx >= 128 && ldw1 >= 128 && ldw2 >= Fp && ldh >= Fp, "ffn_fwd: shape");
    B4C_REQUIRE(ldx % 8 == 0 && ldw1 % 8 == 0 && ldw2 % 8 == 0 && ldh % 8 == 0 &&
                ((((uintptr_t)X | (uintptr_t)W1t | (uintptr_t)W2t | (uintptr_t)H | (uintptr_t)Z | (uintptr_t)Out | (uintptr_t)b2 |
                   (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) && (((uintptr_t)stats & 7) == 0) && (((uintptr_t)b1 & 3) == 0),
                "ffn_fwd: operands must be 16-byte aligned with pitches % 8 == 0");
    B4C_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f, "ffn_fwd: dropout rate");
    FfnFwdArgs a = {};
    a.X = (const bf16_t *)X; a.W1t = (const bf16_t *)W1t; a.b1 = b1; a.W2t = (const bf16_t *)W2t; a.b2 = b2; a.gamma = gamma; a.beta = beta;
    a.H = (bf16_t *)H; a.Z = (bf16_t *)Z; a.Out = (bf16_t *)Out; a.stats = stats;
    a.ldx = ldx; a.ldw1 = ldw1; a.ldw2 = ldw2; a.ldh = ldh; a.Fp = Fp;
    a.eps = eps; a.rate = dropout_rate; a.seed = seed; a.M = M;
    const int64_t ntiles = (M + DD_TOK - 1) / DD_TOK;
    const int grid = (int)(ntiles < 512 ? ntiles : 512);        // two workgroups per CU (57 KB of LDS, <= 128 registers each)
    const size_t lds = DD_RING * (size_t)DD_SUB + DD_SUB + DD_TOK * FF_OSTR + 3 * 128 * 4;
    static thread_local bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void *)ffn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    ffn_fwd_kernel<<<grid, 512, lds, (hipStream_t)stream>>>(a);
    return b4c_check_launch("ffn_fwd");
}
