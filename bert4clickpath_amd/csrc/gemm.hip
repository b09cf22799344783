// MFMA GEMMs for the dense layers (QKV / out-proj / FFN / head MLP / vocab projection).
//
//  gemm_nt : C[M][N]  = epilogue(A[M][K] . Bt[N][K]^T)       forward and backward-dX
//  gemm_tn : dW[K][N] += A[M][K]^T . G[M][N], db += colsum(G)  backward-dW (reduce over tokens)
//
// Both run the same LDS-tiled core: a 128 x 128 output tile per 256-thread workgroup (4 waves as
// 2 x 2, each wave 64 x 64 = 2 x 2 MFMA 32x32 tiles), operands staged through LDS as
// [128 rows][128 B of K] with a 16-B row pad (144-B stride: conflict-free ds_read_b128), two LDS
// stages with register prefetch (global loads for tile t+1 are in flight while tile t is multiplied).
//   bf16: v_mfma_f32_32x32x16_bf16, 64 K-elements per stage;  fp32: v_mfma_f32_32x32x2_f32 (exact
//   fp32 FMA chain, parity path), 32 K-elements per stage.
// gemm_tn stages its operands TRANSPOSED (token-major global rows -> feature-major LDS rows) so the
// reduction axis (tokens) is contiguous for the fragment reads.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// > 64 KiB of dynamic LDS needs an explicit opt-in per kernel (once per process).
template <typename K> static void allow_lds(K kernel, size_t bytes) {
    static thread_local const void *done[16];
    static thread_local int ndone = 0;
    for (int i = 0; i < ndone; ++i)
        if (done[i] == (const void *)kernel) return;
    (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (ndone < 16) done[ndone++] = (const void *)kernel;
}

// Branch-free guarded loads: a raw buffer descriptor over the tile's rows makes out-of-range rows read as 0,
// and an out-of-range K chunk is steered out of range by its offset.  (A per-chunk `if` inside the unrolled
// staging loops becomes control flow, and the compiler then waits vmcnt(0) at every join -- which also waits
// for stores and for younger prefetches.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void *base, int64_t rows_left, int tile_rows, int64_t row_bytes) {
    const int64_t rows = rows_left < tile_rows ? (rows_left < 0 ? 0 : rows_left) : tile_rows;
    const int64_t bytes = rows * row_bytes;           // < 1 GiB for every tile this file uses
    // Every caller passes workgroup-uniform operands, but the clamp above comes out of the vector ALU (v_med3 / v_cndmask) and
    // a descriptor word in a VGPR makes the compiler wrap EVERY buffer load in a readfirstlane "waterfall" loop (gemm_nt_ln:
    // four in the prologue and four inside the k loop).  Saying it once here keeps the descriptor in SGPRs.
    const unsigned nb = __builtin_amdgcn_readfirstlane((unsigned)(bytes > 0x3FFFFFF0ll ? 0x3FFFFFF0ll : bytes));
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, nb, 0x00020000);
}
// or-ing this bit into a byte offset pushes it past every tile descriptor (plain integer arithmetic: a select
// between a real offset and a large constant gets if-converted into two differently-encoded loads)
__device__ __forceinline__ int oob_if(bool invalid) { return invalid ? 0x40000000 : 0; }

#define TILE 128
#define LDS_STRIDE 144                     // bytes per tile row (128 + 16 pad)
#define TILE_BYTES (TILE * LDS_STRIDE)     // 18432
#define STAGE_BYTES (2 * TILE_BYTES)       // A tile + B tile

template <typename T> struct MM;
template <> struct MM<bf16_t> {
    static constexpr int VE = 8;       // elements per 16 B
    static constexpr int BKE = 64;     // K elements per stage
    static constexpr int KSTEPS = 4;   // MFMAs per stage along K (16 each)
    typedef bf16x8 frag_t;
    static __device__ __forceinline__ frag_t ldfrag(const char *row, int kk, int h) {
        return *reinterpret_cast<const bf16x8 *>(row + kk * 32 + h * 16);
    }
    static __device__ __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct MM<float> {
    static constexpr int VE = 4;
    static constexpr int BKE = 32;
    static constexpr int KSTEPS = 16;  // 2 each
    typedef float frag_t;
    static __device__ __forceinline__ frag_t ldfrag(const char *row, int kk, int h) {
        return *reinterpret_cast<const float *>(row + (kk * 2 + h) * 4);
    }
    static __device__ __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

// one stage of MFMAs: acc[i][j] += A_tile(rows wm*64+i*32..) . B_tile(rows wn*64+j*32..)^T
template <typename T>
__device__ __forceinline__ void mma_stage(const char *sA, const char *sB, int wm, int wn, int r, int h, f32x16 (&acc)[2][2]) {
    const char *a0 = sA + (wm * 64 + r) * LDS_STRIDE;
    const char *b0 = sB + (wn * 64 + r) * LDS_STRIDE;
#pragma unroll
    for (int kk = 0; kk < MM<T>::KSTEPS; ++kk) {
        const typename MM<T>::frag_t fa0 = MM<T>::ldfrag(a0, kk, h);
        const typename MM<T>::frag_t fa1 = MM<T>::ldfrag(a0 + 32 * LDS_STRIDE, kk, h);
        const typename MM<T>::frag_t fb0 = MM<T>::ldfrag(b0, kk, h);
        const typename MM<T>::frag_t fb1 = MM<T>::ldfrag(b0 + 32 * LDS_STRIDE, kk, h);
        acc[0][0] = MM<T>::mma(fa0, fb0, acc[0][0]);
        acc[0][1] = MM<T>::mma(fa0, fb1, acc[0][1]);
        acc[1][0] = MM<T>::mma(fa1, fb0, acc[1][0]);
        acc[1][1] = MM<T>::mma(fa1, fb1, acc[1][1]);
    }
}

// ------------------------------------------------------------------------------------------
// NT
// ------------------------------------------------------------------------------------------
// K-contiguous staging: 128 rows x 128 B = 1024 16-B chunks, 4 per thread; 8 lanes cover one row.
template <typename T>
__device__ __forceinline__ void nt_load(const T *__restrict__ P, int ld, int row0, int nrows, int k0, int K, int tid, u32x4 (&reg)[4]) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(P + (int64_t)row0 * ld, (int64_t)nrows - row0, TILE, (int64_t)ld * sizeof(T));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 3, gk = k0 + (c & 7) * MM<T>::VE;
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((row * ld + gk) * (int)sizeof(T)) | oob_if(gk >= K), 0, 0);
    }
}
__device__ __forceinline__ void nt_store(char *s, int tid, const u32x4 (&reg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256;
        *reinterpret_cast<u32x4 *>(s + (c >> 3) * LDS_STRIDE + (c & 7) * 16) = reg[i];
    }
}

// Epilogue on 8 consecutive columns of one row (16-B accesses for bias / gate / residual / C).
template <typename T, typename OutT>
__device__ __forceinline__ void epilogue8(float (&v)[8], int64_t row, int col, OutT *__restrict__ C, int ldc,
                                          const float *__restrict__ bias, int act, const T *__restrict__ gate, int ldg,
                                          const T *__restrict__ residual, int ldr) {
    if (bias) {
        float bv[8];
        Vec8<float>::load(bias + col, bv);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bv[k];
    }
    if (act == B4C_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    if (gate) {
        float g[8];
        Vec8<T>::load(gate + row * ldg + col, g);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = g[k] > 0.f ? v[k] : 0.f;
    }
    if (residual) {
        float rr[8];
        Vec8<T>::load(residual + row * ldr + col, rr);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += rr[k];
    }
    Vec8<OutT>::store(C + row * ldc + col, v);
}

// One LDS stage (36.9 KB -> 4 workgroups per CU) with register prefetch of the next K tile.  The kernel is
// latency-bound at K = 128 (two K tiles), so dependent round trips to memory are issued early:
//   * the bias chunk of each lane is loaded at entry,
//   * gate / residual chunks are requested before the last tile's MFMAs and consumed after them.
// MFMA roles are swapped (acc^T = Bt_tile . A_tile^T: rows = n in registers, col = m on the lane) so a lane
// owns 4 consecutive n of one output row: 2-byte outputs go to the LDS transpose as 8-byte writes and leave
// as 16-byte row chunks.
#define OUT_STRIDE 136   // bytes per staged output row (128 + 8: 8-byte writes spread over the banks)

template <typename T, typename OutT, bool EPI>
__global__ void __launch_bounds__(256, 3) gemm_nt_kernel(const T *__restrict__ A, int lda, const T *__restrict__ Bt, int ldb,
                                                      OutT *__restrict__ C, int ldc, int M, int N, int K,
                                                      const float *__restrict__ bias, int act, const T *__restrict__ gate,
                                                      int ldg, const T *__restrict__ residual, int ldr, int vec_ok, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    // blockIdx.x walks the N tiles first: neighbours in dispatch order share the A tile (L2) and write
    // adjacent 256-B segments of the same output rows (DRAM pages), which matters for the vocabulary GEMM
    const int ntn = (N + TILE - 1) / TILE;
    int mt_i = blockIdx.x / ntn, nt_i = blockIdx.x % ntn;
    if (xcd_map) {
        // Workgroup ids go round-robin over the 8 XCDs, each with its own L2: with the N tiles of one M tile on
        // consecutive ids, its A tile is fetched from HBM once per N tile (measured: the QKV projection, 3 N tiles,
        // moved 1.5x its algorithmic bytes).  Here ids i, i+8, i+16, ... (one XCD) walk the N tiles of one M tile.
        const int grp = blockIdx.x / (8 * ntn), rem = blockIdx.x % (8 * ntn);
        mt_i = grp * 8 + (rem & 7);
        nt_i = rem >> 3;
    }
    const int m0 = mt_i * TILE, n0 = nt_i * TILE;
    if (m0 >= M) return;          // (the XCD map rounds the M tiles up to a multiple of 8)
    const bool vec = (sizeof(OutT) == 2) && vec_ok;
    f32x16 acc[2][2];   // acc[j][i]: rows = n (tile j of this wave's 64 columns), col = m (tile i of its 64 rows)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;

    const int nk = (K + MM<T>::BKE - 1) / MM<T>::BKE;
    u32x4 xa[4], xb[4];
    nt_load<T>(A, lda, m0, M, 0, K, tid, xa);
    nt_load<T>(Bt, ldb, n0, N, 0, K, tid, xb);
    // epilogue chunk q of this lane: row (lane + 64 q) >> 3 of the wave's 64 rows, columns 8 * (lane & 7) ...
    const int part = lane & 7;
    const int gcol = n0 + wn * 64 + part * 8;
    float bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bv[k] = 0.f;
    if (vec && bias && gcol < N) Vec8<float>::load(bias + gcol, bv);
    nt_store(smem, tid, xa);
    nt_store(smem + TILE_BYTES, tid, xb);
    __syncthreads();
    // one prefetched epilogue operand (gate if given, else residual): requested before the last tile's MFMAs,
    // into the staging registers that are idle by then
    const T *epi = EPI ? (gate ? gate : residual) : nullptr;
    const int lde = gate ? ldg : ldr;
    u32x4 pe[EPI ? 8 : 1];
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            nt_load<T>(A, lda, m0, M, (kt + 1) * MM<T>::BKE, K, tid, xa);
            nt_load<T>(Bt, ldb, n0, N, (kt + 1) * MM<T>::BKE, K, tid, xb);
        } else if (EPI && vec && epi) {
            // the first half of the epilogue operand rides under the last tile's MFMAs, the second half is requested
            // right after them (16 registers less: three workgroups per CU instead of two)
#pragma unroll
            for (int q = 0; q < (EPI ? 4 : 1); ++q) {
                const int64_t grow = m0 + wm * 64 + ((lane + q * 64) >> 3);
                pe[q] = (grow < M && gcol < N) ? *reinterpret_cast<const u32x4 *>(epi + grow * lde + gcol) : (u32x4){0u, 0u, 0u, 0u};
            }
        }
        mma_stage<T>(smem + TILE_BYTES, smem, wn, wm, r, h, acc);
        __syncthreads();
        if (more) {
            nt_store(smem, tid, xa);
            nt_store(smem + TILE_BYTES, tid, xb);
            __syncthreads();
        }
    }
    // acc[j][i] register t: n = j*32 + (t&3) + 8*(t>>2) + 4*h (4 consecutive n per t>>2), m = i*32 + r.
    if (vec) {
        if (EPI && epi) {
#pragma unroll
            for (int q = 4; q < (EPI ? 8 : 4); ++q) {
                const int64_t grow = m0 + wm * 64 + ((lane + q * 64) >> 3);
                pe[EPI ? q : 0] = (grow < M && gcol < N) ? *reinterpret_cast<const u32x4 *>(epi + grow * lde + gcol) : (u32x4){0u, 0u, 0u, 0u};
            }
        }
        char *ws = smem + wave * (64 * OUT_STRIDE);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
                    bf16x4_t w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = (bf16_t)acc[j][i][4 * tq + k];
                    *reinterpret_cast<bf16x4_t *>(ws + (i * 32 + r) * OUT_STRIDE + (j * 32 + 8 * tq + 4 * h) * 2) = w;
                }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = (lane + q * 64) >> 3;
            const int64_t grow = m0 + wm * 64 + row;
            if (grow < M && gcol < N) {
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(ws + row * OUT_STRIDE + part * 16);
                const u32x2 hi = *reinterpret_cast<const u32x2 *>(ws + row * OUT_STRIDE + part * 16 + 8);
                const u32x4 w4 = {lo[0], lo[1], hi[0], hi[1]};
                const bf16x8 cv = __builtin_bit_cast(bf16x8, w4);
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (float)cv[k] + bv[k];
                if (act == B4C_ACT_RELU) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                }
                if (EPI && gate) {
                    const bf16x8 gv = __builtin_bit_cast(bf16x8, pe[EPI ? q : 0]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = ((float)gv[k] > 0.f) ? v[k] : 0.f;
                    if (residual) {   // both operands at once: not on the model's path, loaded late
                        float rr[8];
                        Vec8<T>::load(residual + grow * ldr + gcol, rr);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += rr[k];
                    }
                } else if (EPI && residual) {
                    const bf16x8 rv = __builtin_bit_cast(bf16x8, pe[EPI ? q : 0]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += (float)rv[k];
                }
                Vec8<OutT>::template store_sel<B4C_NT(B4C_NT_GEMM)>(C + grow * ldc + gcol, v);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + wm * 64 + i * 32 + r;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int col = n0 + wn * 64 + j * 32 + (t & 3) + 8 * (t >> 2) + 4 * h;
                if (row < M && col < N) {
                    float v = acc[j][i][t] + (bias ? bias[col] : 0.f);
                    if (act == B4C_ACT_RELU) v = fmaxf(v, 0.f);
                    if (gate) v = ((float)gate[(int64_t)row * ldg + col] > 0.f) ? v : 0.f;
                    if (residual) v += (float)residual[(int64_t)row * ldr + col];
                    C[(int64_t)row * ldc + col] = (OutT)v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// NT + residual + dropout + LayerNorm (bf16, N <= 128 = one tile wide): the out-projection / FFN2 GEMM of an
// encoder layer and the "x + dropout(y) -> LayerNorm" that follows it (transformer.py:204-206, 211-213) in one
// pass: y never goes to HBM (2 of the 6 [T][d] passes of the pair).  Mainloop = gemm_nt_kernel.  The epilogue
// stages the whole 128 x 128 bf16 tile in LDS, then 16 consecutive lanes own one row (8 columns each), exactly
// the lane layout and the summation order of add_ln_fwd_kernel<bf16, 16>: z, out and stats are bit-identical
// to gemm_nt followed by add_ln_fwd.
// ------------------------------------------------------------------------------------------
#define LN_OUT_STRIDE 264   // bytes per staged row: 256 + 8 (8-byte writes of 16 rows spread over the 32 write banks)

__global__ void __launch_bounds__(256, 3) gemm_nt_ln_kernel(const bf16_t *__restrict__ A, int lda, const bf16_t *__restrict__ Bt, int ldb,
                                                         const float *__restrict__ bias, const bf16_t *__restrict__ x, int ldx,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         bf16_t *__restrict__ z, bf16_t *__restrict__ out, float *__restrict__ stats,
                                                         int M, int N, int K, float eps, float rate, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * TILE;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    const int nk = (K + MM<bf16_t>::BKE - 1) / MM<bf16_t>::BKE;
    u32x4 xa[4], xb[4];
    nt_load<bf16_t>(A, lda, m0, M, 0, K, tid, xa);
    nt_load<bf16_t>(Bt, ldb, 0, N, 0, K, tid, xb);
    // epilogue chunk q of this thread: row (tid >> 4) + 16 q of the tile, columns 8 (tid & 15) ...
    const int part = tid & 15, col = part * 8;
    float bv[8], gv[8], be[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { bv[k] = 0.f; gv[k] = 0.f; be[k] = 0.f; }
    if (col < N) {
        if (bias) Vec8<float>::load(bias + col, bv);
        Vec8<float>::load(gamma + col, gv);
        Vec8<float>::load(beta + col, be);
    }
    nt_store(smem, tid, xa);
    nt_store(smem + TILE_BYTES, tid, xb);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t xrs = tile_rsrc(x + (int64_t)m0 * ldx, (int64_t)M - m0, TILE, (int64_t)ldx * 2);
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            nt_load<bf16_t>(A, lda, m0, M, (kt + 1) * MM<bf16_t>::BKE, K, tid, xa);
            nt_load<bf16_t>(Bt, ldb, 0, N, (kt + 1) * MM<bf16_t>::BKE, K, tid, xb);
        } else {
            // residual chunks 0..3 ride under the last tile's MFMAs (xa is idle by now)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xa[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ((((tid >> 4) + 16 * i) * ldx + col) * 2) | oob_if(col >= N), 0, 0);
        }
        mma_stage<bf16_t>(smem + TILE_BYTES, smem, wn, wm, r, h, acc);
        __syncthreads();
        if (more) {
            nt_store(smem, tid, xa);
            nt_store(smem + TILE_BYTES, tid, xb);
            __syncthreads();
        }
    }
    // acc[j][i] register t: n = wn*64 + j*32 + (t&3) + 8*(t>>2) + 4*h, m = wm*64 + i*32 + r  ->  LDS [m][n] bf16
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
                bf16x4_t w;
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = (bf16_t)acc[j][i][4 * tq + k];
                *reinterpret_cast<bf16x4_t *>(smem + (wm * 64 + i * 32 + r) * LN_OUT_STRIDE + (wn * 64 + j * 32 + 8 * tq + 4 * h) * 2) = w;
            }
    __syncthreads();
    const float inv_keep = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
    const float inv_d = 1.0f / (float)N;
    // two halves of four chunks: chunks 4..7 of the residual are requested while 0..3 are processed (the loop over
    // the halves is not unrolled: the epilogue is VALU work -- dropout hashes -- and needs occupancy, i.e. few registers)
#pragma unroll 1
    for (int hq = 0; hq < 2; ++hq) {
      if (hq == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xb[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ((((tid >> 4) + 16 * (4 + i)) * ldx + col) * 2) | oob_if(col >= N), 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = hq * 4 + i;
        const int row = (tid >> 4) + 16 * q;
        const int64_t grow = m0 + row;
        const bool on = grow < M && col < N;     // rows are uniform over their 16 lanes
        const u32x4 xcur = xa[i];
        float v[8];
        float sum = 0.f;
        if (on) {
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(smem + row * LN_OUT_STRIDE + part * 16);
            const u32x2 hi = *reinterpret_cast<const u32x2 *>(smem + row * LN_OUT_STRIDE + part * 16 + 8);
            const u32x4 w4 = {lo[0], lo[1], hi[0], hi[1]};
            const bf16x8 cv = __builtin_bit_cast(bf16x8, w4);
            const bf16x8 xv = __builtin_bit_cast(bf16x8, xcur);
            const uint32_t km = rate > 0.f ? b4c_keep8(seed, (uint64_t)(grow * N + col), b4c_keep_threshold(rate)) : 0xFFu;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float yy = (float)(bf16_t)((float)cv[k] + bv[k]);      // the bf16 y that gemm_nt would have stored
                if (rate > 0.f) yy = ((km >> k) & 1u) ? yy * inv_keep : 0.f;
                v[k] = (float)xv[k] + yy;
                sum += v[k];
            }
            if (z) Vec8<bf16_t>::store_sel<B4C_NT(B4C_NT_GEMMLN_Z)>(z + grow * N + col, v);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
        const float mean = group_sum<16>(sum) * inv_d;
        float sq = 0.f;
        if (on) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float dlt = v[k] - mean;
                sq += dlt * dlt;
            }
        }
        const float var = group_sum<16>(sq) * inv_d;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (on) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (v[k] - mean) * rstd * gv[k] + be[k];
            Vec8<bf16_t>::store_sel<B4C_NT(B4C_NT_GEMMLN_OUT)>(out + grow * N + col, o);
            if (part == 0 && stats) {
                stats[grow * 2] = mean;
                stats[grow * 2 + 1] = rstd;
            }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) xa[i] = xb[i];
    }
}

// The same for 128 < N <= 256 (config 4: d_model = 256): a workgroup owns 64 whole rows x 256 columns (4 waves side by
// side along N, each 64 x 64), the B tile is 256 weight rows per K stage.  32 consecutive lanes own one row in the
// epilogue (8 columns each): the lane layout and summation order of add_ln_fwd_kernel<bf16, 32>, so z / out / stats
// are again bit-identical to gemm_nt followed by add_ln_fwd.
#define LN2_TM 64
#define LN2_TN 256
#define LN2_STAGE_BYTES ((LN2_TM + LN2_TN) * LDS_STRIDE)     // 46,080: three workgroups per CU
#define LN2_OUT_STRIDE 520                                   // bytes per staged output row: 512 + 8

__device__ __forceinline__ void ln2_load_a(const bf16_t *__restrict__ P, int ld, int row0, int nrows, int k0, int K, int tid, u32x4 (&reg)[2]) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(P + (int64_t)row0 * ld, (int64_t)nrows - row0, LN2_TM, (int64_t)ld * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 3, gk = k0 + (c & 7) * 8;
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((row * ld + gk) * 2) | oob_if(gk >= K), 0, 0);
    }
}
__device__ __forceinline__ void ln2_load_b(const bf16_t *__restrict__ P, int ld, int nrows, int k0, int K, int tid, u32x4 (&reg)[8]) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(P, (int64_t)nrows, LN2_TN, (int64_t)ld * 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 3, gk = k0 + (c & 7) * 8;
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((row * ld + gk) * 2) | oob_if(gk >= K), 0, 0);
    }
}
template <int NR> __device__ __forceinline__ void ln2_store(char *s, int tid, const u32x4 (&reg)[NR]) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = tid + i * 256;
        *reinterpret_cast<u32x4 *>(s + (c >> 3) * LDS_STRIDE + (c & 7) * 16) = reg[i];
    }
}

__global__ void __launch_bounds__(256, 2) gemm_nt_ln256_kernel(const bf16_t *__restrict__ A, int lda, const bf16_t *__restrict__ Bt, int ldb,
                                                            const float *__restrict__ bias, const bf16_t *__restrict__ x, int ldx,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            bf16_t *__restrict__ z, bf16_t *__restrict__ out, float *__restrict__ stats,
                                                            int M, int N, int K, float eps, float rate, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem, *sB = smem + LN2_TM * LDS_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * LN2_TM;
    f32x16 acc[2][2];     // acc[j][i]: rows = n (tile j of the wave's 64 columns), col = m (tile i of the 64 rows)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    const int nk = (K + 63) / 64;
    u32x4 xa[2], xb[8];
    ln2_load_a(A, lda, m0, M, 0, K, tid, xa);
    ln2_load_b(Bt, ldb, N, 0, K, tid, xb);
    // epilogue chunk q of this thread: row (tid >> 5) + 8 q of the tile, columns 8 (tid & 31) ...
    const int part = tid & 31, col = part * 8;
    float bv[8], gv[8], be[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { bv[k] = 0.f; gv[k] = 0.f; be[k] = 0.f; }
    if (col < N) {
        if (bias) Vec8<float>::load(bias + col, bv);
        Vec8<float>::load(gamma + col, gv);
        Vec8<float>::load(beta + col, be);
    }
    ln2_store<2>(sA, tid, xa);
    ln2_store<8>(sB, tid, xb);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t xrs = tile_rsrc(x + (int64_t)m0 * ldx, (int64_t)M - m0, LN2_TM, (int64_t)ldx * 2);
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            ln2_load_a(A, lda, m0, M, (kt + 1) * 64, K, tid, xa);
            ln2_load_b(Bt, ldb, N, (kt + 1) * 64, K, tid, xb);
        } else {
            // the eight residual chunks of this thread ride under the last stage's MFMAs (xb is idle by now)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                xb[q] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ((((tid >> 5) + 8 * q) * ldx + col) * 2) | oob_if(col >= N), 0, 0);
        }
        mma_stage<bf16_t>(sB, sA, wave, 0, r, h, acc);
        __syncthreads();
        if (more) {
            ln2_store<2>(sA, tid, xa);
            ln2_store<8>(sB, tid, xb);
            __syncthreads();
        }
    }
    // acc[j][i] register t: n = wave*64 + j*32 + (t&3) + 8*(t>>2) + 4*h, m = i*32 + r  ->  LDS [m][n] bf16
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
                bf16x4_t w;
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = (bf16_t)acc[j][i][4 * tq + k];
                *reinterpret_cast<bf16x4_t *>(smem + (i * 32 + r) * LN2_OUT_STRIDE + (wave * 64 + j * 32 + 8 * tq + 4 * h) * 2) = w;
            }
    __syncthreads();
    const float inv_keep = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
    const float inv_d = 1.0f / (float)N;
#pragma unroll 2
    for (int q = 0; q < 8; ++q) {
        const int row = (tid >> 5) + 8 * q;
        const int64_t grow = m0 + row;
        const bool on = grow < M && col < N;     // rows are uniform over their 32 lanes
        float v[8];
        float sum = 0.f;
        if (on) {
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(smem + row * LN2_OUT_STRIDE + part * 16);
            const u32x2 hi = *reinterpret_cast<const u32x2 *>(smem + row * LN2_OUT_STRIDE + part * 16 + 8);
            const u32x4 w4 = {lo[0], lo[1], hi[0], hi[1]};
            const bf16x8 cv = __builtin_bit_cast(bf16x8, w4);
            const bf16x8 xv = __builtin_bit_cast(bf16x8, xb[q]);
            const uint32_t km = rate > 0.f ? b4c_keep8(seed, (uint64_t)(grow * N + col), b4c_keep_threshold(rate)) : 0xFFu;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float yy = (float)(bf16_t)((float)cv[k] + bv[k]);      // the bf16 y that gemm_nt would have stored
                if (rate > 0.f) yy = ((km >> k) & 1u) ? yy * inv_keep : 0.f;
                v[k] = (float)xv[k] + yy;
                sum += v[k];
            }
            if (z) Vec8<bf16_t>::store_sel<B4C_NT(B4C_NT_GEMMLN_Z)>(z + grow * N + col, v);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        }
        const float mean = group_sum<32>(sum) * inv_d;
        float sq = 0.f;
        if (on) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float dlt = v[k] - mean;
                sq += dlt * dlt;
            }
        }
        const float var = group_sum<32>(sq) * inv_d;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (on) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (v[k] - mean) * rstd * gv[k] + be[k];
            Vec8<bf16_t>::store_sel<B4C_NT(B4C_NT_GEMMLN_OUT)>(out + grow * N + col, o);
            if (part == 0 && stats) {
                stats[grow * 2] = mean;
                stats[grow * 2 + 1] = rstd;
            }
        }
    }
}

extern "C" int b4c_gemm_nt_add_ln(const void *A, int lda, const void *Bt, int ldb, const float *bias, const void *x, int ldx,
                                  const float *gamma, const float *beta, void *z, void *out, float *stats, int M, int N, int K,
                                  float eps, float dropout_rate, uint64_t seed, int dtype, void *stream) {
    B4C_REQUIRE(A && Bt && x && gamma && beta && out, "gemm_nt_add_ln: null pointer");     // z / stats may be NULL (inference)
    B4C_REQUIRE(dtype == B4C_BF16, "gemm_nt_add_ln: bf16 only (dtype %d)", dtype);
    B4C_REQUIRE(M > 0 && N > 0 && N <= LN2_TN && N % 8 == 0 && K > 0 && K % 8 == 0, "gemm_nt_add_ln: M=%d N=%d K=%d (N <= 256, N, K %% 8 == 0)", M, N, K);
    B4C_REQUIRE(lda >= K && ldb >= K && ldx >= N && lda % 8 == 0 && ldb % 8 == 0 && ldx % 8 == 0, "gemm_nt_add_ln: pitches");
    B4C_REQUIRE(((((uintptr_t)A | (uintptr_t)Bt | (uintptr_t)x | (uintptr_t)z | (uintptr_t)out) & 15) == 0) &&
                ((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)bias) & 15) == 0), "gemm_nt_add_ln: operands must be 16-byte aligned");
    B4C_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f, "gemm_nt_add_ln: dropout_rate %f", (double)dropout_rate);
    if (N > TILE) {
        const int grid2 = (M + LN2_TM - 1) / LN2_TM;
        gemm_nt_ln256_kernel<<<grid2, 256, LN2_STAGE_BYTES, (hipStream_t)stream>>>((const bf16_t *)A, lda, (const bf16_t *)Bt, ldb, bias,
                                                                                   (const bf16_t *)x, ldx, gamma, beta, (bf16_t *)z,
                                                                                   (bf16_t *)out, stats, M, N, K, eps, dropout_rate, seed);
        return b4c_check_launch("gemm_nt_add_ln256");
    }
    const int grid = (M + TILE - 1) / TILE;
    gemm_nt_ln_kernel<<<grid, 256, STAGE_BYTES, (hipStream_t)stream>>>((const bf16_t *)A, lda, (const bf16_t *)Bt, ldb, bias,
                                                                       (const bf16_t *)x, ldx, gamma, beta, (bf16_t *)z,
                                                                       (bf16_t *)out, stats, M, N, K, eps, dropout_rate, seed);
    return b4c_check_launch("gemm_nt_add_ln");
}


// Wide-N, small-K form (the vocabulary projection: [R][128] x [V][128]^T -> [R][V]).  The generic kernel
// above spends most of its time outside stores and MFMAs there (with both removed it still runs at half its
// full duration: 125 k workgroups each pay tile loads, staging and barriers for 32 KB of output).  Here a
// workgroup keeps its 128-row A tile in LDS and walks a chunk of N tiles: per tile one W tile (L2), 32 MFMAs
// per consumer wave over the whole K, and the output leaves through an LDS transpose as 16-B row chunks.
#define WIDE_STR 272                         // bytes per LDS row: K = 128 bf16 + 16 pad
#define WIDE_TILE_BYTES (TILE * WIDE_STR)    // 34816 == 4 waves * 64 rows * OUT_STRIDE

__device__ __forceinline__ void wide_load(const bf16_t *__restrict__ P, int ld, int row0, int nrows, int K, int tid, u32x4 (&reg)[8]) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(P + (int64_t)row0 * ld, (int64_t)nrows - row0, TILE, (int64_t)ld * 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 4, cc = c & 15;
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((row * ld + cc * 8) * 2) | oob_if(cc * 8 >= K), 0, 0);
    }
}
template <int OFF>
__device__ __forceinline__ void wide_load_r(const bf16_t *__restrict__ P, int ld, int row0, int nrows, int K, int tid, u32x4 (&R)[16]) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(P + (int64_t)row0 * ld, (int64_t)nrows - row0, TILE, (int64_t)ld * 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 4, cc = c & 15;
        R[OFF + i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((row * ld + cc * 8) * 2) | oob_if(cc * 8 >= K), 0, 0);
    }
}
template <int OFF>
__device__ __forceinline__ void wide_store_r(char *s, int tid, const u32x4 (&R)[16]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + i * 256;
        *reinterpret_cast<u32x4 *>(s + (c >> 4) * WIDE_STR + (c & 15) * 16) = R[OFF + i];
    }
}
__device__ __forceinline__ void wide_store(char *s, int tid, const u32x4 (&reg)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + i * 256;
        *reinterpret_cast<u32x4 *>(s + (c >> 4) * WIDE_STR + (c & 15) * 16) = reg[i];
    }
}

// Two wave roles in one 512-thread workgroup per CU.  Ablation of a version whose compute waves also stored
// showed its phases ADDING instead of overlapping (loop skeleton 1.06 us/tile + W loads 0.68 + stores 1.43): a
// wave that issues 8 x 1 KB stores stalls in store ISSUE at the CU's ~22 GB/s drain rate.  So:
//   waves 0-3  compute: A fragments live in registers for the whole chunk; 32 MFMAs per tile over K <= 128,
//              + bias, bf16, into a double-buffered LDS transpose; they never touch global memory in the loop
//   waves 4-7  move: W tile (+ bias slice) L2 -> registers (two tiles in flight) -> LDS, double buffered, and the
//              PREVIOUS output tile LDS -> 16-B row chunks (4 rows x 256 B per wave-instruction) -> HBM.  Their
//              loads are always older than the stores they issue next, so the counted vmcnt wait for a W tile
//              never waits for the latest stores.
// One LDS-only barrier per tile.  (Measured at R = 40,960, V = 50,000: 1.32 ms = 3.1 TB/s; splitting the move
// role further into 2 load + 2 store waves was slower, 1.53 ms.)
#define WOUT_STR 264                          // bytes per staged output row: 128 bf16 + 8 pad (8-B writes 2-way at most)
#define WOUT_BYTES (TILE * WOUT_STR)          // 33792

__global__ void __launch_bounds__(512) gemm_nt_wide_kernel(const bf16_t *__restrict__ A, int lda, const bf16_t *__restrict__ Bt, int ldb,
                                                           bf16_t *__restrict__ C, int ldc, int M, int N, int K,
                                                           const float *__restrict__ bias, int mt, int tiles_per_chunk, int chunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sW = smem;                                     // 2 x WIDE_TILE_BYTES
    char *sOut = smem + 2 * WIDE_TILE_BYTES;             // 2 x WOUT_BYTES
    float *sBias = reinterpret_cast<float *>(smem + 2 * WIDE_TILE_BYTES + 2 * WOUT_BYTES);   // 2 x 128
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = wave >> 2;                          // 0 compute, 1 move
    const int rtid = tid & 255;
    const int ntn = (N + TILE - 1) / TILE;
    // chunks > 0: the `chunks` workgroups of one M tile have consecutive ids (co-resident, the A rows shared in L2) and
    // walk the N tiles INTERLEAVED (workgroup c takes tiles c, c + chunks, ...): at any moment they extend the same 128
    // output rows by adjacent 256-B pieces -- longer contiguous runs per row for the write-back than when each workgroup
    // sweeps a distant column range.  chunks == 0: the first form (one contiguous range of N tiles per workgroup).
    const int tstep = chunks > 0 ? chunks : 1;
    const int m0 = (chunks > 0 ? (int)(blockIdx.x / chunks) : (int)(blockIdx.x % mt)) * TILE;
    const int nt0 = chunks > 0 ? (int)(blockIdx.x % chunks) : (int)(blockIdx.x / mt) * tiles_per_chunk;
    const int nt1 = chunks > 0 ? ntn : min(ntn, nt0 + tiles_per_chunk);
    if (nt0 >= nt1) return;
    const int ntiles = (nt1 - nt0 + tstep - 1) / tstep;
    const int wm = (wave & 3) >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;

    // One 16 x 16-B register block serves whichever role the wave has (the roles are exclusive; separate arrays
    // would be allocated side by side): compute waves keep their A fragments in it
    // (AF(i,kk) = A[m0 + wm*64 + i*32 + r][kk*16 + 8h ..]), move waves two in-flight W tiles (R[0..7] / R[8..15]).
    u32x4 R[16];
#define AF(i, kk) __builtin_bit_cast(bf16x8, R[(i) * 8 + (kk)])
    float xb = 0.f, yb = 0.f;
    if (role == 0) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(A + (int64_t)m0 * lda, (int64_t)M - m0, TILE, (int64_t)lda * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int row = wm * 64 + i * 32 + r, col = kk * 16 + h * 8;
                R[i * 8 + kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((row * lda + col) * 2) | oob_if(col >= K), 0, 0);
            }
    } else {
        wide_load_r<0>(Bt, ldb, nt0 * TILE, N, K, rtid, R);
        wide_store_r<0>(sW, rtid, R);
        if (rtid < 128) sBias[rtid] = (bias && nt0 * TILE + rtid < N) ? bias[nt0 * TILE + rtid] : 0.f;
        if (ntiles > 1) {
            wide_load_r<0>(Bt, ldb, (nt0 + tstep) * TILE, N, K, rtid, R);
            if (rtid < 128 && bias && (nt0 + tstep) * TILE + rtid < N) xb = bias[(nt0 + tstep) * TILE + rtid];
        }
    }
    __syncthreads();

    for (int t = 0; t <= ntiles; ++t) {                  // one extra iteration: the stores trail by one tile
        const int buf = t & 1;
        const int n0 = (nt0 + t * tstep) * TILE;
        const int n2 = n0 + 2 * tstep * TILE;            // the tile two steps ahead
        if (role == 1) {
            if (buf == 0) {          // x (R[0..7]) -> LDS buffer 1, refill y (R[8..15])
                if (t + 2 < ntiles) {
                    wide_load_r<8>(Bt, ldb, n2, N, K, rtid, R);
                    yb = (rtid < 128 && bias && n2 + rtid < N) ? bias[n2 + rtid] : 0.f;
                }
                if (t + 1 < ntiles) {
                    wide_store_r<0>(sW + WIDE_TILE_BYTES, rtid, R);
                    if (rtid < 128) sBias[128 + rtid] = xb;
                }
            } else {                 // y -> LDS buffer 0, refill x
                if (t + 2 < ntiles) {
                    wide_load_r<0>(Bt, ldb, n2, N, K, rtid, R);
                    xb = (rtid < 128 && bias && n2 + rtid < N) ? bias[n2 + rtid] : 0.f;
                }
                if (t + 1 < ntiles) {
                    wide_store_r<8>(sW, rtid, R);
                    if (rtid < 128) sBias[rtid] = yb;
                }
            }
            if (t >= 1) {            // tile t-1: rows of 256 B, 16 lanes per row
                const char *so = sOut + (buf ^ 1) * WOUT_BYTES;
                const int ns = n0 - tstep * TILE;
                const bool interior = (m0 + TILE <= M) && (ns + TILE <= N);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = rtid + q * 256;
                    const int row = c >> 4, pc = c & 15;
                    const int64_t grow = m0 + row;
                    const int gcol = ns + pc * 8;
                    const u32x2 lo = *reinterpret_cast<const u32x2 *>(so + row * WOUT_STR + pc * 16);
                    const u32x2 hi = *reinterpret_cast<const u32x2 *>(so + row * WOUT_STR + pc * 16 + 8);
                    if (interior || (grow < M && gcol < N)) {     // interior tiles: uniform, no per-store control flow
                        const u32x4 w4 = {lo[0], lo[1], hi[0], hi[1]};
                        *reinterpret_cast<u32x4 *>(C + grow * ldc + gcol) = w4;
                    }
                }
            }
        } else {
            if (t < ntiles) {
                const char *w0 = sW + buf * WIDE_TILE_BYTES + (wn * 64 + r) * WIDE_STR;
                f32x16 acc[2][2];    // acc[j][i]: rows = n (tile j of the wave's 64 columns), col = m (tile i of its 64 rows)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 16; ++q) acc[j][i][q] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const bf16x8 fw0 = *reinterpret_cast<const bf16x8 *>(w0 + kk * 32 + h * 16);
                    const bf16x8 fw1 = *reinterpret_cast<const bf16x8 *>(w0 + 32 * WIDE_STR + kk * 32 + h * 16);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0, AF(0, kk), acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0, AF(1, kk), acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1, AF(0, kk), acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1, AF(1, kk), acc[1][1], 0, 0, 0);
                }
                // reg 4 tq + k of acc[j][i]: n = wn*64 + j*32 + 8 tq + 4 h + k, m = wm*64 + i*32 + r
                char *so = sOut + buf * WOUT_BYTES;
                const float *sb = sBias + buf * 128;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int tq = 0; tq < 4; ++tq) {
                        const int nl = wn * 64 + j * 32 + 8 * tq + 4 * h;
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(sb + nl);
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
                            bf16x4_t w;
#pragma unroll
                            for (int k = 0; k < 4; ++k) w[k] = (bf16_t)(acc[j][i][4 * tq + k] + b4[k]);
                            *reinterpret_cast<bf16x4_t *>(so + (wm * 64 + i * 32 + r) * WOUT_STR + nl * 2) = w;
                        }
                    }
            }
        }
        B4C_LDS_BARRIER();
    }
#undef AF
}

// Second form of the materialised vocabulary projection (the DEFAULT since the logits live on a 256-B aligned row pitch,
// ops.row_pitch: with V = 50,000 on its natural 100,000-B pitch both forms sat at 3.1 - 3.3 TB/s because every 256-B row
// piece straddled 128-B lines; aligned, this form writes 4.8 - 4.9 TB/s = 0.60 of the HBM spec, the two-role form 3.3 - 3.5).
// It is sensitive to code generation: a wave-uniform `if (plain) store else nontemporal store` in the store loop cost 17 %.
// 256-thread workgroups, TWO per CU (69 KB of LDS each), every wave
// computes AND moves; the phases of the two co-resident workgroups overlap (one stores its tile while the other
// multiplies) instead of two wave roles inside one workgroup.  A fragments stay in registers for the workgroup's
// whole chunk of N tiles, the next W tile is in flight in registers during the current tile's MFMAs and stores.
// SM: the epilogue writes softmax probabilities 2^((x + b) log2 e - lse2[row]) instead of logits (lse2 from b4c_vocab_lse):
// Dense(V, softmax) of head.py:36 in one pass over the (R x V) tensor instead of three.
template <bool SM>
__global__ void __launch_bounds__(256, 2) gemm_nt_wide2_kernel(const bf16_t *__restrict__ A, int lda, const bf16_t *__restrict__ Bt, int ldb,
                                                            bf16_t *__restrict__ C, int ldc, int M, int N, int K,
                                                            const float *__restrict__ bias, int mt, int tiles_per_chunk,
                                                            const float *__restrict__ lse2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sW = smem;
    char *sOut = smem + WIDE_TILE_BYTES;
    float *sBias = reinterpret_cast<float *>(smem + WIDE_TILE_BYTES + WOUT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (blockIdx.x % mt) * TILE;
    const int ntn = (N + TILE - 1) / TILE;
    const int nt0 = (blockIdx.x / mt) * tiles_per_chunk;
    const int nt1 = min(ntn, nt0 + tiles_per_chunk);
    if (nt0 >= nt1) return;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    u32x4 R[16];
#define AF2(i, kk) __builtin_bit_cast(bf16x8, R[(i) * 8 + (kk)])
    {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(A + (int64_t)m0 * lda, (int64_t)M - m0, TILE, (int64_t)lda * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int row = wm * 64 + i * 32 + r, col = kk * 16 + h * 8;
                R[i * 8 + kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((row * lda + col) * 2) | oob_if(col >= K), 0, 0);
            }
    }
    float ls[2] = {0.f, 0.f};
    if (SM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + wm * 64 + i * 32 + r;
            ls[i] = row < M ? lse2[row] : 0.f;
        }
    }
    u32x4 pw[8];
    float pb = 0.f;
    auto fetch = [&](int nt) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(Bt + (int64_t)nt * TILE * ldb, (int64_t)N - (int64_t)nt * TILE, TILE, (int64_t)ldb * 2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 4, cc = c & 15;
            pw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((row * ldb + cc * 8) * 2) | oob_if(cc * 8 >= K), 0, 0);
        }
        const int col = nt * TILE + (tid & 127);
        pb = (bias && col < N) ? bias[col] : 0.f;
    };
    fetch(nt0);
    for (int nt = nt0; nt < nt1; ++nt) {
        const int n0 = nt * TILE;
        wide_store(sW, tid, pw);
        if (tid < 128) sBias[tid] = pb;
        __syncthreads();
        if (nt + 1 < nt1) fetch(nt + 1);
        const char *w0 = sW + (wn * 64 + r) * WIDE_STR;
        f32x16 acc[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[j][i][q] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const bf16x8 fw0 = *reinterpret_cast<const bf16x8 *>(w0 + kk * 32 + h * 16);
            const bf16x8 fw1 = *reinterpret_cast<const bf16x8 *>(w0 + 32 * WIDE_STR + kk * 32 + h * 16);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0, AF2(0, kk), acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0, AF2(1, kk), acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1, AF2(0, kk), acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1, AF2(1, kk), acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const int nl = wn * 64 + j * 32 + 8 * tq + 4 * h;
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(sBias + nl);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
                    bf16x4_t w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float x = acc[j][i][4 * tq + k] + b4[k];
                        w[k] = (bf16_t)(SM ? __builtin_amdgcn_exp2f(__builtin_fmaf(x, 1.4426950408889634f, -ls[i])) : x);
                    }
                    *reinterpret_cast<bf16x4_t *>(sOut + (wm * 64 + i * 32 + r) * WOUT_STR + nl * 2) = w;
                }
            }
        __syncthreads();
        const bool interior = (m0 + TILE <= M) && (n0 + TILE <= N);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = tid + q * 256;
            const int row = c >> 4, pc = c & 15;
            const int64_t grow = m0 + row;
            const int gcol = n0 + pc * 8;
            const u32x2 lo = *reinterpret_cast<const u32x2 *>(sOut + row * WOUT_STR + pc * 16);
            const u32x2 hi = *reinterpret_cast<const u32x2 *>(sOut + row * WOUT_STR + pc * 16 + 8);
            if (interior || (grow < M && gcol < N)) {
                const u32x4 w4 = {lo[0], lo[1], hi[0], hi[1]};
                __builtin_nontemporal_store(w4, reinterpret_cast<u32x4 *>(C + grow * ldc + gcol));
            }
        }
    }
#undef AF2
}

static int g_wide_form = -1;
static int wide_form() {
    if (g_wide_form < 0) {
        const char *e = getenv("B4C_WIDE_FORM");
        g_wide_form = e ? atoi(e) : 2;
    }
    return g_wide_form;
}
extern "C" void b4c_set_wide_form(int f) { g_wide_form = f; }     // scratch A/B switch (not part of include/b4c.h)

static bool vec_ok_wide(const void *C, int ldc, int N, const float *bias) {
    return (N % 8 == 0) && (ldc % 8 == 0) && (((uintptr_t)C & 15) == 0) && (!bias || ((uintptr_t)bias & 15) == 0);
}

// probabilities [M][N] (bf16) = softmax over N of (A Bt^T + bias), given lse2[row] = log2 sum_j 2^(x_j log2 e) (b4c_vocab_lse)
extern "C" int b4c_gemm_nt_softmax(const void *A, int lda, const void *Bt, int ldb, void *C, int ldc, int M, int N, int K,
                                   const float *bias, const float *lse2, void *stream) {
    B4C_REQUIRE(A && Bt && C && lse2 && M > 0 && N > 0 && K > 0, "gemm_nt_softmax: null pointer / empty (M=%d N=%d K=%d)", M, N, K);
    B4C_REQUIRE(K % 8 == 0 && K <= 128 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K,
                "gemm_nt_softmax: K=%d (<= 128) lda=%d ldb=%d must be multiples of 8 with ld >= K", K, lda, ldb);
    B4C_REQUIRE(ldc >= N && vec_ok_wide(C, ldc, N, bias), "gemm_nt_softmax: N=%d ldc=%d must be multiples of 8, C / bias 16-byte aligned", N, ldc);
    B4C_REQUIRE((((uintptr_t)A | (uintptr_t)Bt) & 15) == 0, "gemm_nt_softmax: operands must be 16-byte aligned");
    const int mt = (int)ceil_div64(M, TILE), ntn = (int)ceil_div64(N, TILE);
    int chunks = (int)ceil_div64(512 * 5, mt);       // two workgroups per CU resident, ~5 rounds over the launch
    if (chunks > ntn) chunks = ntn;
    if (chunks < 1) chunks = 1;
    const int tpc = (int)ceil_div64(ntn, chunks);
    chunks = (int)ceil_div64(ntn, tpc);
    const size_t shm = WIDE_TILE_BYTES + WOUT_BYTES + 128 * sizeof(float);
    allow_lds(gemm_nt_wide2_kernel<true>, shm);
    gemm_nt_wide2_kernel<true><<<mt * chunks, 256, shm, (hipStream_t)stream>>>((const bf16_t *)A, lda, (const bf16_t *)Bt, ldb, (bf16_t *)C, ldc,
                                                                           M, N, K, bias, mt, tpc, lse2);
    return b4c_check_launch("gemm_nt_softmax");
}

extern "C" int b4c_gemm_nt(const void *A, int lda, const void *Bt, int ldb, void *C, int ldc, int M, int N, int K,
                           const float *bias, int act, const void *gate, int ldg, const void *residual, int ldr,
                           int dtype, int out_dtype, void *stream) {
    B4C_REQUIRE(A && Bt && C && M > 0 && N > 0 && K > 0, "gemm_nt: null pointer / empty (M=%d N=%d K=%d)", M, N, K);
    B4C_REQUIRE(dtype == B4C_F32 || dtype == B4C_BF16, "gemm_nt: dtype %d", dtype);
    const int ve = dtype == B4C_BF16 ? 8 : 4;
    B4C_REQUIRE(K % ve == 0 && lda % ve == 0 && ldb % ve == 0 && lda >= K && ldb >= K,
                "gemm_nt: K=%d lda=%d ldb=%d must be multiples of %d with ld >= K", K, lda, ldb, ve);
    B4C_REQUIRE(ldc >= N, "gemm_nt: ldc %d < N %d", ldc, N);
    B4C_REQUIRE((((uintptr_t)A | (uintptr_t)Bt) & 15) == 0, "gemm_nt: operands must be 16-byte aligned");
    B4C_REQUIRE(out_dtype == dtype || out_dtype == B4C_F32, "gemm_nt: out_dtype %d", out_dtype);
    B4C_REQUIRE(act == B4C_ACT_NONE || act == B4C_ACT_RELU, "gemm_nt: act %d", act);
    hipStream_t st_w = (hipStream_t)stream;
    if (dtype == B4C_BF16 && out_dtype == B4C_BF16 && K <= 128 && N >= 2048 && act == B4C_ACT_NONE && !gate && !residual &&
        vec_ok_wide(C, ldc, N, bias)) {
        const int mt = (int)ceil_div64(M, TILE), ntn = (int)ceil_div64(N, TILE);
        int chunks = (int)ceil_div64(1280, mt);          // 1 workgroup per CU resident, ~5 per CU over the launch
        if (chunks > ntn) chunks = ntn;
        if (chunks < 1) chunks = 1;
        const int tpc = (int)ceil_div64(ntn, chunks);
        chunks = (int)ceil_div64(ntn, tpc);
        if (wide_form() == 2) {
            int chunks2 = (int)ceil_div64(512 * 5, mt);      // two workgroups per CU resident, ~5 rounds over the launch
            if (chunks2 > ntn) chunks2 = ntn;
            if (chunks2 < 1) chunks2 = 1;
            const int tpc2 = (int)ceil_div64(ntn, chunks2);
            chunks2 = (int)ceil_div64(ntn, tpc2);
            const size_t shm2 = WIDE_TILE_BYTES + WOUT_BYTES + 128 * sizeof(float);
            allow_lds(gemm_nt_wide2_kernel<false>, shm2);
            gemm_nt_wide2_kernel<false><<<mt * chunks2, 256, shm2, st_w>>>((const bf16_t *)A, lda, (const bf16_t *)Bt, ldb, (bf16_t *)C, ldc, M, N, K, bias, mt, tpc2, nullptr);
            return b4c_check_launch("gemm_nt_wide2");
        }
        const size_t shm_w = 2 * WIDE_TILE_BYTES + 2 * WOUT_BYTES + 2 * 128 * sizeof(float);
        allow_lds(gemm_nt_wide_kernel, shm_w);
        const int interleave = wide_form() == 3 ? chunks : 0;
        gemm_nt_wide_kernel<<<mt * chunks, 512, shm_w, st_w>>>((const bf16_t *)A, lda, (const bf16_t *)Bt, ldb, (bf16_t *)C, ldc, M, N, K, bias, mt, tpc, interleave);
        return b4c_check_launch("gemm_nt_wide");
    }
    const int64_t mt_n = ceil_div64(M, TILE), nt_n = ceil_div64(N, TILE);
    const int xcd_map = (nt_n > 1 && nt_n <= 8) ? 1 : 0;      // few N tiles: keep one M tile's workgroups on one XCD (L2)
    const int64_t nblocks = (xcd_map ? ceil_div64(mt_n, 8) * 8 : mt_n) * nt_n;
    B4C_REQUIRE(nblocks < (1ll << 31), "gemm_nt: too many tiles");
    dim3 grid((unsigned)nblocks);
    hipStream_t st = (hipStream_t)stream;
    const size_t shm = STAGE_BYTES;
    // the 16-byte epilogue needs whole, aligned 8-column chunks in C / gate / residual / bias
    const int vec_ok = (N % 8 == 0) && (ldc % 8 == 0) && (((uintptr_t)C & 15) == 0) &&
                       (!gate || (ldg % 8 == 0 && ((uintptr_t)gate & 15) == 0)) &&
                       (!residual || (ldr % 8 == 0 && ((uintptr_t)residual & 15) == 0)) &&
                       (!bias || ((uintptr_t)bias & 15) == 0);
#define NT_ARGS(TT, OT) (const TT *)A, lda, (const TT *)Bt, ldb, (OT *)C, ldc, M, N, K, bias, act, (const TT *)gate, ldg, (const TT *)residual, ldr, vec_ok, xcd_map
    const bool epi = gate || residual;
    if (dtype == B4C_F32) {
        if (epi) gemm_nt_kernel<float, float, true><<<grid, 256, shm, st>>>(NT_ARGS(float, float));
        else gemm_nt_kernel<float, float, false><<<grid, 256, shm, st>>>(NT_ARGS(float, float));
    } else if (out_dtype == B4C_F32) {
        if (epi) gemm_nt_kernel<bf16_t, float, true><<<grid, 256, shm, st>>>(NT_ARGS(bf16_t, float));
        else gemm_nt_kernel<bf16_t, float, false><<<grid, 256, shm, st>>>(NT_ARGS(bf16_t, float));
    } else {
        if (epi) gemm_nt_kernel<bf16_t, bf16_t, true><<<grid, 256, shm, st>>>(NT_ARGS(bf16_t, bf16_t));
        else gemm_nt_kernel<bf16_t, bf16_t, false><<<grid, 256, shm, st>>>(NT_ARGS(bf16_t, bf16_t));
    }
#undef NT_ARGS
    return b4c_check_launch("gemm_nt");
}

// ------------------------------------------------------------------------------------------
// TN (dW): transposing stage.  Per step a workgroup consumes TOK tokens x 128 features of each
// operand.  Global rows are token-major; a wave reads whole 256-B / 512-B feature rows (coalesced),
// each lane keeps its feature(s) for 8 (bf16) / 4 (fp32) consecutive tokens and writes them as one
// 16-B token-contiguous LDS chunk.
// ------------------------------------------------------------------------------------------
// dW / db destinations of gemm_tn: N may be cut into up to 4 equal column segments, each with its own
// fp32 matrix [K][segw] (the fused Q|K|V projection writes straight into the three gradient tensors).
struct TNOut {
    float *dW[4];
    float *db[4];
    int segw;      // columns per segment (== N when there is one)
    int ld;        // row pitch of every dW segment
    int mode;      // how a workgroup's partial tile leaves: TN_ATOMIC | TN_DIRECT | TN_WS
    float *ws;     // TN_WS: partial tiles [split][tile][4096 float4 groups]
    float *ws_db;  // TN_WS: partial column sums [split][tile column][128]
};
enum { TN_ATOMIC = 0, TN_DIRECT = 1, TN_WS = 2 };
__device__ __forceinline__ float *tn_w_ptr(const TNOut &o, int row, int col) {
    const int seg = col / o.segw;
    return o.dW[seg] + (int64_t)row * o.ld + (col - seg * o.segw);
}
__device__ __forceinline__ float *tn_b_ptr(const TNOut &o, int col) {
    const int seg = col / o.segw;
    return o.db[seg] + (col - seg * o.segw);
}
// The accumulator tile of one workgroup (4 waves x 2 x 2 MFMA 32x32 tiles, 128 x 128 outputs) leaves the kernel
//   TN_WS:     as it sits in the registers -- float4 group ((wave*2+i)*2+j)*4+tq of lane -> 1 KB per wave
//              store, no atomics; tn_reduce_kernel sums the splits in a fixed order (deterministic dW);
//   TN_DIRECT: one split only -> plain read-modify-write;
//   TN_ATOMIC: float atomics (no workspace given).  256 splits hitting the same 512 cache lines serialise
//              at the memory-side atomic unit: 28 us for 16 MB at C2, which is why TN_WS exists.
struct TNBlock { int bx, by, bz, gx, gy; };     // tile coordinates / tile-grid size of one problem (a launch may hold several)
__device__ __forceinline__ void tn_emit_tile(const TNOut &out, const f32x16 (&acc)[2][2], int k0, int n0, int K, int N,
                                             int wave, int lane, const TNBlock &bk) {
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    if (out.mode == TN_WS) {
        const int64_t tile = ((int64_t)bk.bz * bk.gx + bk.bx) * bk.gy + bk.by;
        f32x4 *w = reinterpret_cast<f32x4 *>(out.ws) + tile * 4096;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    const f32x4 v = {acc[i][j][4 * tq], acc[i][j][4 * tq + 1], acc[i][j][4 * tq + 2], acc[i][j][4 * tq + 3]};
                    w[((((wave * 2 + i) * 2 + j) * 4 + tq) << 6) + lane] = v;
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int row = k0 + wm * 64 + i * 32 + (t & 3) + 8 * (t >> 2) + 4 * h;
                if (row < K && col < N) {
                    float *p = tn_w_ptr(out, row, col);
                    if (out.mode == TN_DIRECT) *p += acc[i][j][t];
                    else atomicAdd(p, acc[i][j][t]);
                }
            }
        }
}
// column sums of the workgroup's 128 columns (red[128] in LDS, complete)
__device__ __forceinline__ void tn_emit_bias(const TNOut &out, const float *red, int n0, int N, int tid, const TNBlock &bk) {
    if (tid >= 128) return;
    if (out.mode == TN_WS) {
        out.ws_db[((int64_t)bk.bz * bk.gy + bk.by) * 128 + tid] = red[tid];
    } else if (n0 + tid < N) {
        float *p = tn_b_ptr(out, n0 + tid);
        if (out.mode == TN_DIRECT) *p += red[tid];
        else atomicAdd(p, red[tid]);
    }
}

// Second pass of TN_WS (latency-bound: every thread has at most ceil(nsplit/64) independent 16-B loads in
// flight).  Blocks [0, tiles*256): 1024 threads = 16 float4 groups of one output tile x 64 split phases;
// shuffles, then LDS across the 16 waves, then dW += sum.  Blocks [tiles*256, +tn): column sums.
// blocks per 128 x 128 output tile.  A block sums GPB groups of four outputs over 1024 / GPB partitions of the splits: 16
// groups x 64 partitions for the encoder layers' ~40 splits; with few splits (the head trunk: 8) most of those partitions
// would idle, so 64 groups x 16 partitions (a quarter of the blocks).
__host__ __device__ __forceinline__ int tn_reduce_bpt(int nsplit) { return nsplit <= 16 ? 64 : 256; }

__device__ __forceinline__ void tn_reduce_body(const TNOut &out, int K, int N, int tk, int tn, int nsplit, int blk) {
    __shared__ f32x4 part[16][64];
    const int tid = threadIdx.x;
    const int tiles = tk * tn;
    const int bpt = tn_reduce_bpt(nsplit);
    if (blk >= tiles * bpt) {
        if (!out.db[0]) return;
        const int by = blk - tiles * bpt, c = tid & 127, zp = tid >> 7;
        float s = 0.f;
        for (int z = zp; z < nsplit; z += 8) s += out.ws_db[((int64_t)z * tn + by) * 128 + c];
        float *red = reinterpret_cast<float *>(part);
        red[zp * 128 + c] = s;
        __syncthreads();
        if (!zp && by * 128 + c < N) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q * 128 + c];
            *tn_b_ptr(out, by * 128 + c) += t;
        }
        return;
    }
    const bool narrow = bpt == 64;          // uniform over the launch
    const int tile = blk / bpt;
    const int grp = narrow ? (((blk & 63) << 6) + (tid & 63)) : (((blk & 255) << 4) + (tid & 15));
    const int zp = narrow ? (tid >> 6) : (tid >> 4);
    const f32x4 *w = reinterpret_cast<const f32x4 *>(out.ws) + (int64_t)tile * 4096 + grp;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const int64_t zstride = (int64_t)tiles * 4096;
    if (narrow) {
        for (int z = zp; z < nsplit; z += 16) s += w[z * zstride];
        part[tid >> 6][tid & 63] = s;
    } else {
#pragma unroll 4
        for (int z = zp; z < nsplit; z += 64) s += w[z * zstride];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] += __shfl_xor(s[k], 16);
            s[k] += __shfl_xor(s[k], 32);
        }
        if ((tid & 63) < 16) part[tid >> 6][tid & 15] = s;
    }
    __syncthreads();
    if (tid < (narrow ? 64 : 16)) {
        f32x4 v = part[0][tid];
#pragma unroll
        for (int q = 1; q < 16; ++q) v += part[q][tid];
        const int lane = grp & 63, tq = (grp >> 6) & 3, j = (grp >> 8) & 1, i = (grp >> 9) & 1, wave = grp >> 10;
        const int k0 = (tile / tn) * TILE, n0 = (tile % tn) * TILE;
        const int col = n0 + (wave & 1) * 64 + j * 32 + (lane & 31);
        const int row = k0 + (wave >> 1) * 64 + i * 32 + 8 * tq + 4 * (lane >> 5);
        if (col < N)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (row + k < K) *tn_w_ptr(out, row + k, col) += v[k];
    }
}

__global__ void __launch_bounds__(1024) tn_reduce_kernel(TNOut out, int K, int N, int tk, int tn, int nsplit) {
    tn_reduce_body(out, K, N, tk, tn, nsplit, (int)blockIdx.x);
}

template <typename T> struct TNStage;

template <> struct TNStage<bf16_t> {
    static constexpr int TOK = 64;  // tokens per step; wave w owns tokens [16w, 16w+16)
    unsigned v[2][8];               // [group of 8 tokens][token] -> features (2*lane, 2*lane+1) packed
    __device__ __forceinline__ void load(const bf16_t *__restrict__ P, int ld, int f0, int64_t tok0, int64_t tok_end, int lane, int wave) {
        const int f = f0 + 2 * lane;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t t = tok0 + wave * 16 + g * 8 + j;
                v[g][j] = (t < tok_end && f < ld) ? *reinterpret_cast<const unsigned *>(P + t * ld + f) : 0u;
            }
    }
    __device__ __forceinline__ void store(char *s, int lane, int wave) const {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            u32x4 lo, hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned a = v[g][2 * q], b = v[g][2 * q + 1];
                lo[q] = (a & 0xFFFFu) | (b << 16);
                hi[q] = (a >> 16) | (b & 0xFFFF0000u);
            }
            char *base = s + (2 * lane) * LDS_STRIDE + (wave * 16 + g * 8) * 2;
            *reinterpret_cast<u32x4 *>(base) = lo;
            *reinterpret_cast<u32x4 *>(base + LDS_STRIDE) = hi;
        }
    }
    // column sums of this thread's two features over its 16 tokens
    __device__ __forceinline__ void colsum(float &s0, float &s1) const {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s0 += __uint_as_float(v[g][j] << 16);
                s1 += __uint_as_float(v[g][j] & 0xFFFF0000u);
            }
    }
    static __device__ __forceinline__ int feat0(int lane) { return 2 * lane; }
    static __device__ __forceinline__ int feat1(int lane) { return 2 * lane + 1; }
};

template <> struct TNStage<float> {
    static constexpr int TOK = 32;  // wave w owns tokens [8w, 8w+8)
    float v[2][4][2];               // [group of 4 tokens][token][feature lane / lane+64]
    __device__ __forceinline__ void load(const float *__restrict__ P, int ld, int f0, int64_t tok0, int64_t tok_end, int lane, int wave) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t t = tok0 + wave * 8 + g * 4 + j;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f = f0 + lane + 64 * q;
                    v[g][j][q] = (t < tok_end && f < ld) ? P[t * ld + f] : 0.f;
                }
            }
    }
    __device__ __forceinline__ void store(char *s, int lane, int wave) const {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 w = {v[g][0][q], v[g][1][q], v[g][2][q], v[g][3][q]};
                *reinterpret_cast<f32x4 *>(s + (lane + 64 * q) * LDS_STRIDE + (wave * 8 + g * 4) * 4) = w;
            }
    }
    __device__ __forceinline__ void colsum(float &s0, float &s1) const {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0 += v[g][j][0]; s1 += v[g][j][1]; }
    }
    static __device__ __forceinline__ int feat0(int lane) { return lane; }
    static __device__ __forceinline__ int feat1(int lane) { return lane + 64; }
};

template <typename T>
__global__ void __launch_bounds__(256) gemm_tn_kernel(const T *__restrict__ A, int lda, const T *__restrict__ G, int ldg,
                                                      TNOut out, int64_t M,
                                                      int K, int N, int64_t chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int k0 = blockIdx.x * TILE, n0 = blockIdx.y * TILE;
    const int64_t m_begin = blockIdx.z * chunk;
    const int64_t m_end = (m_begin + chunk < M) ? m_begin + chunk : M;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    float bs0 = 0.f, bs1 = 0.f;
    const bool want_db = (out.db[0] != nullptr) && blockIdx.x == 0;

    constexpr int TOK = TNStage<T>::TOK;
    const int nsteps = (int)((m_end - m_begin + TOK - 1) / TOK);
    TNStage<T> sa, sg;
    if (nsteps > 0) {
        sa.load(A, lda, k0, m_begin, m_end, lane, wave);
        sg.load(G, ldg, n0, m_begin, m_end, lane, wave);
        sa.store(smem, lane, wave);
        sg.store(smem + TILE_BYTES, lane, wave);
        if (want_db) sg.colsum(bs0, bs1);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const bool more = st + 1 < nsteps;
        if (more) {
            sa.load(A, lda, k0, m_begin + (int64_t)(st + 1) * TOK, m_end, lane, wave);
            sg.load(G, ldg, n0, m_begin + (int64_t)(st + 1) * TOK, m_end, lane, wave);
        }
        mma_stage<T>(smem, smem + TILE_BYTES, wm, wn, r, h, acc);
        __syncthreads();
        if (more) {
            sa.store(smem, lane, wave);
            sg.store(smem + TILE_BYTES, lane, wave);
            if (want_db) sg.colsum(bs0, bs1);
            __syncthreads();
        }
    }
    const TNBlock bk = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y};
    tn_emit_tile(out, acc, k0, n0, K, N, wave, lane, bk);
    if (want_db) {
        // the 4 waves hold partial sums of the same features over different tokens
        float *red = reinterpret_cast<float *>(smem);  // all MFMA reads are behind the last barrier
        float *tmp = red + 128;
        tmp[wave * 128 + TNStage<T>::feat0(lane)] = bs0;
        tmp[wave * 128 + TNStage<T>::feat1(lane)] = bs1;
        __syncthreads();
        if (tid < 128) red[tid] = tmp[tid] + tmp[128 + tid] + tmp[256 + tid] + tmp[384 + tid];
        __syncthreads();
        tn_emit_bias(out, red, n0, N, tid, bk);
    }
}

// bf16 dW: token-major tiles are staged AS THEY ARE (16-B loads, 16-B LDS writes: 64 tokens x 128 features
// per operand, 320-B rows so the transposed reads are conflict-free) and the MFMA fragments -- 8 consecutive
// tokens of one feature -- come out of ds_read_b64_tr_b16.  No transposing stage, 4x fewer load instructions
// than the dword path above (which remains the fp32 path).
#define TN_STR 320                      // bytes per token row in LDS (256 + 64)
#define TN_TILE_BYTES (64 * TN_STR)     // one operand, one stage

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ bf16x8 tn_frag(const char *p) {   // tokens +0..3 and +4..7 of the lane's feature
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(p));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3))) *)(p + 4 * TN_STR));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t w = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ void tn_load16(const bf16_t *__restrict__ P, int ld, int f0, int64_t tok0, int64_t tok_end, int tid, u32x4 (&reg)[4]) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(P + tok0 * ld, tok_end - tok0, 64, (int64_t)ld * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256;
        const int f = f0 + (c & 15) * 8;
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (((c >> 4) * ld + f) * 2) | oob_if(f >= ld), 0, 0);
    }
}
__device__ __forceinline__ void tn_store16(char *s, int tid, const u32x4 (&reg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256;
        *reinterpret_cast<u32x4 *>(s + (c >> 4) * TN_STR + (c & 15) * 16) = reg[i];
    }
}

template <int D>
__device__ __forceinline__ void tn_bf16_body(const bf16_t *__restrict__ A, int lda, const bf16_t *__restrict__ G, int ldg,
                                             const TNOut &out, int64_t M, int K, int N, int64_t chunk, const TNBlock &bk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5, li = lane & 15, g = lane >> 4;
    const int k0 = bk.bx * TILE, n0 = bk.by * TILE;
    const int64_t m_begin = bk.bz * chunk;
    const int64_t m_end = (m_begin + chunk < M) ? m_begin + chunk : M;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    float bs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bs[k] = 0.f;
    const bool want_db = (out.db[0] != nullptr) && bk.bx == 0;
    const int nsteps = (int)((m_end - m_begin + 63) / 64);
    char *sA = smem, *sG = smem + TN_TILE_BYTES;
    // D register sets: while the MFMAs run on the tile in LDS, tiles st+1 .. st+D are in flight (the barriers
    // order LDS only, so loads stay outstanding across them and the waits are counted).
    u32x4 ra[D][4], rg[D][4];
    auto colsum = [&](const u32x4 (&q)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, q[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) bs[k] += (float)v[k];
        }
    };
    // the lane's transposed-read base: token row (li >> 2) (+ 8 h), feature column 16 (g & 1) + 4 (li & 3) of a 32-wide tile
    const int lane_off = ((li >> 2) + 8 * h) * TN_STR + (16 * (g & 1) + 4 * (li & 3)) * 2;
    // tile t (t >= nsteps loads nothing: the descriptor has no rows, every lane reads zeros)
    auto load = [&](int t, u32x4 (&qa)[4], u32x4 (&qg)[4]) {
        const int64_t tok0 = m_begin + (int64_t)t * 64;
        const int64_t e = tok0 < m_end ? m_end : tok0;
        tn_load16(A, lda, k0, tok0, e, tid, qa);
        tn_load16(G, ldg, n0, tok0, e, tid, qg);
    };
#pragma unroll
    for (int p = 0; p < D; ++p) load(p, ra[p], rg[p]);
    tn_store16(sA, tid, ra[0]);
    tn_store16(sG, tid, rg[0]);
    if (want_db) colsum(rg[0]);
    B4C_LDS_BARRIER();
    for (int st = 0; st < nsteps; st += D) {
#pragma unroll
        for (int p = 0; p < D; ++p) {
            const int t = st + p;                // the tile in LDS; set p is free again
            if (t < nsteps) {
                load(t + D, ra[p], rg[p]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {   // 16 tokens per MFMA
                    const char *pa = sA + kk * 16 * TN_STR + lane_off + (wm * 64) * 2;
                    const char *pg = sG + kk * 16 * TN_STR + lane_off + (wn * 64) * 2;
                    const bf16x8 fa0 = tn_frag(pa), fa1 = tn_frag(pa + 64);
                    const bf16x8 fb0 = tn_frag(pg), fb1 = tn_frag(pg + 64);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
                }
                B4C_LDS_BARRIER();
                if (t + 1 < nsteps) {
                    constexpr int dummy = 0; (void)dummy;
                    tn_store16(sA, tid, ra[(p + 1) % D]);
                    tn_store16(sG, tid, rg[(p + 1) % D]);
                    if (want_db) colsum(rg[(p + 1) % D]);
                    B4C_LDS_BARRIER();
                }
            }
        }
    }
    tn_emit_tile(out, acc, k0, n0, K, N, wave, lane, bk);
    if (want_db) {
        float *red = reinterpret_cast<float *>(smem);   // all fragment reads are behind the last barrier
        float *tmp = red + 128;                          // [16 token rows of the staging pattern][128 columns]
#pragma unroll
        for (int k = 0; k < 8; ++k) tmp[(tid >> 4) * 128 + (tid & 15) * 8 + k] = bs[k];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += tmp[q * 128 + tid];
            red[tid] = t;
        }
        __syncthreads();
        tn_emit_bias(out, red, n0, N, tid, bk);
    }
}

template <int D>
__global__ void __launch_bounds__(256) gemm_tn_bf16_kernel(const bf16_t *__restrict__ A, int lda, const bf16_t *__restrict__ G, int ldg,
                                                           TNOut out, int64_t M, int K, int N, int64_t chunk) {
    const TNBlock bk = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y};
    tn_bf16_body<D>(A, lda, G, ldg, out, M, K, N, chunk, bk);
}

// Several dW problems over the SAME token axis in one launch (the four weight gradients of an encoder layer):
// blockIdx.x walks the output tiles of all problems, blockIdx.z the token splits.  The ~256 workgroups are shared
// by all problems, so each output tile has ~256 / tiles partial sums instead of 256 -- 6x less partial-tile
// traffic at C2 -- and the layer needs one main + one reduce launch instead of four of each.
#define TN_GROUP_MAX 8
struct TNProb {
    const bf16_t *A, *G;
    int lda, ldg, K, N, tk, tn, tile0;      // tile0 = index of the problem's first tile in the launch
    TNOut out;
};
struct TNGroup {
    TNProb p[TN_GROUP_MAX];
    int np, tiles;
};
template <int D>
__global__ void __launch_bounds__(256) gemm_tn_bf16_group_kernel(TNGroup g, int64_t M, int64_t chunk) {
    // Workgroups go to the 8 XCDs round-robin by linear id, and every output tile of a token chunk re-reads that chunk's
    // A / G panels (d = 256: 40 panel reads for 18 distinct panels per layer).  When the number of chunks is a multiple
    // of 8, all tiles of chunk c are therefore put on XCD c % 8, next to each other in its dispatch order: they stream the
    // chunk together and share the panels in that XCD's L2.
    int tile = blockIdx.x, cz = blockIdx.z;
    if ((gridDim.z & 7) == 0) {
        const int lin = blockIdx.x + g.tiles * blockIdx.z, slot = lin >> 3;
        tile = slot % g.tiles;
        cz = (slot / g.tiles) * 8 + (lin & 7);
    }
    int pi = 0;
#pragma unroll
    for (int i = 1; i < TN_GROUP_MAX; ++i)
        if (i < g.np && tile >= g.p[i].tile0) pi = i;
    const TNProb &pr = g.p[pi];
    const int lt = tile - pr.tile0;
    const TNBlock bk = {lt / pr.tn, lt % pr.tn, cz, pr.tk, pr.tn};
    tn_bf16_body<D>(pr.A, pr.lda, pr.G, pr.ldg, pr.out, M, pr.K, pr.N, chunk, bk);
}

// one reduce launch for the whole group: blocks are laid out problem after problem (tiles * 256 + tn blocks each)
__global__ void __launch_bounds__(1024) tn_reduce_group_kernel(TNGroup g, int nsplit) {
    int blk = blockIdx.x;
    for (int i = 0; i < g.np; ++i) {
        const TNProb &pr = g.p[i];
        const int nb = pr.tk * pr.tn * tn_reduce_bpt(nsplit) + pr.tn;
        if (blk < nb) {
            tn_reduce_body(pr.out, pr.K, pr.N, pr.tk, pr.tn, nsplit, blk);
            return;
        }
        blk -= nb;
    }
}

// Split rule.  Few output tiles (encoder layers: 1-3): ~256 workgroups in all, each streaming its token chunk
// with two tiles in flight -- one workgroup per CU runs at copy speed (5.2 TB/s measured) and every extra split
// only adds partial tiles to reduce.  Many tiles (vocabulary projection: 391): one split, no reduction at all.
static void tn_plan(int M, int K, int N, int dtype, int *tk, int *tn, int64_t *nsplit, int64_t *chunk) {
    *tk = (K + TILE - 1) / TILE;
    *tn = (N + TILE - 1) / TILE;
    const int tok = dtype == B4C_BF16 ? 64 : 32;
    const int64_t tiles = (int64_t)*tk * *tn;
    int64_t ns = tiles >= 192 ? 1 : (256 + tiles / 2) / tiles;
    const int64_t max_split = ceil_div64(M, (int64_t)tok * 4);
    if (ns > max_split) ns = max_split;
    if (ns < 1) ns = 1;
    *chunk = ceil_div64(ceil_div64(M, ns), tok) * tok;
    *nsplit = ceil_div64(M, *chunk);
}

extern "C" int64_t b4c_gemm_tn_workspace_bytes(int M, int K, int N, int dtype) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    int tk, tn;
    int64_t nsplit, chunk;
    tn_plan(M, K, N, dtype, &tk, &tn, &nsplit, &chunk);
    if (nsplit == 1) return 0;
    return nsplit * ((int64_t)tk * tn * 16384 + (int64_t)tn * 128) * 4;
}

static int gemm_tn_launch(const void *A, int lda, const void *G, int ldg, TNOut out, int M, int K, int N, int dtype,
                          void *workspace, int64_t workspace_bytes, void *stream) {
    B4C_REQUIRE(A && G && out.dW[0] && M > 0 && K > 0 && N > 0, "gemm_tn: null pointer / empty");
    B4C_REQUIRE(dtype == B4C_F32 || dtype == B4C_BF16, "gemm_tn: dtype %d", dtype);
    B4C_REQUIRE(lda >= K && ldg >= N, "gemm_tn: pitches too small (lda=%d K=%d ldg=%d N=%d)", lda, K, ldg, N);
    B4C_REQUIRE(lda % 2 == 0 && ldg % 2 == 0, "gemm_tn: operand pitches must be even");
    B4C_REQUIRE((((uintptr_t)A | (uintptr_t)G) & 3) == 0, "gemm_tn: operands must be 4-byte aligned");
    int tk, tn;
    int64_t nsplit, chunk;
    tn_plan(M, K, N, dtype, &tk, &tn, &nsplit, &chunk);
    B4C_REQUIRE(tn <= 65535 && nsplit <= 65535, "gemm_tn: N too large");
    const int64_t need = nsplit == 1 ? 0 : nsplit * ((int64_t)tk * tn * 16384 + (int64_t)tn * 128) * 4;
    if (nsplit == 1) {
        out.mode = TN_DIRECT;
    } else if (workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0) {
        out.mode = TN_WS;
        out.ws = (float *)workspace;
        out.ws_db = out.ws + nsplit * tk * tn * 16384;
    } else {
        out.mode = TN_ATOMIC;
    }
    dim3 grid(tk, tn, (unsigned)nsplit);
    hipStream_t st = (hipStream_t)stream;
    const size_t shm = STAGE_BYTES;
    const bool vec16 = (lda % 8 == 0) && (ldg % 8 == 0) && ((((uintptr_t)A | (uintptr_t)G) & 15) == 0);
    if (dtype == B4C_F32)
        gemm_tn_kernel<float><<<grid, 256, shm, st>>>((const float *)A, lda, (const float *)G, ldg, out, M, K, N, chunk);
    else if (vec16)
        gemm_tn_bf16_kernel<2><<<grid, 256, 2 * TN_TILE_BYTES, st>>>((const bf16_t *)A, lda, (const bf16_t *)G, ldg, out, M, K, N, chunk);
    else
        gemm_tn_kernel<bf16_t><<<grid, 256, shm, st>>>((const bf16_t *)A, lda, (const bf16_t *)G, ldg, out, M, K, N, chunk);
    if (out.mode == TN_WS)
        tn_reduce_kernel<<<tk * tn * tn_reduce_bpt((int)nsplit) + tn, 1024, 0, st>>>(out, K, N, tk, tn, (int)nsplit);
    return b4c_check_launch("gemm_tn");
}

extern "C" int b4c_gemm_tn(const void *A, int lda, const void *G, int ldg, float *dW, int ldw, float *db, int M, int K,
                           int N, int dtype, void *workspace, int64_t workspace_bytes, void *stream) {
    B4C_REQUIRE(ldw >= N, "gemm_tn: ldw %d < N %d", ldw, N);
    TNOut out = {{dW, nullptr, nullptr, nullptr}, {db, nullptr, nullptr, nullptr}, N, ldw, TN_ATOMIC, nullptr, nullptr};
    return gemm_tn_launch(A, lda, G, ldg, out, M, K, N, dtype, workspace, workspace_bytes, stream);
}

extern "C" int b4c_gemm_tn_seg(const void *A, int lda, const void *G, int ldg, int n_seg, float *const *h_dW,
                               float *const *h_db, int seg_width, int M, int K, int dtype, void *workspace,
                               int64_t workspace_bytes, void *stream) {
    B4C_REQUIRE(n_seg >= 1 && n_seg <= 4 && seg_width > 0 && h_dW, "gemm_tn_seg: n_seg %d / seg_width %d", n_seg, seg_width);
    TNOut out = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}, seg_width, seg_width, TN_ATOMIC, nullptr, nullptr};
    for (int i = 0; i < n_seg; ++i) {
        B4C_REQUIRE(h_dW[i] && (!h_db || h_db[i] || !h_db[0]), "gemm_tn_seg: null segment pointer %d", i);
        out.dW[i] = h_dW[i];
        out.db[i] = h_db ? h_db[i] : nullptr;
    }
    return gemm_tn_launch(A, lda, G, ldg, out, M, K, n_seg * seg_width, dtype, workspace, workspace_bytes, stream);
}

// ---- grouped dW: several (A_i^T G_i) over the same M tokens in one launch ----
// Split rule of the grouped launch.  Few tiles per chunk (d = 128: 6): ~256 workgroups, one per CU.  Many (d = 256: 20): ~512
// -- measured at config 4: 320 workgroups 4.05 ms per step, 480: 3.25, 800: 3.33; at C2 240: 1.05, 384: 1.09, 528: 1.20.
// The number of chunks is made a multiple of 8 so that the kernel's XCD-aware tile order applies (C4: 5.25 -> 4.05 ms at the
// same workgroup count, C2: 1.17 -> 1.06).
static void tn_group_plan(int M, int tiles, int64_t *nsplit, int64_t *chunk) {
    const int target = tiles > 8 ? 512 : 256;
    int64_t ns = (target + tiles / 2) / tiles;
    const int64_t max_split = ceil_div64(M, 64 * 4);
    if (ns > max_split) ns = max_split;
    if (ns < 1) ns = 1;
    if (ns >= 8) {
        ns = (ns + 4) / 8 * 8;
        while (ns > max_split) ns -= 8;
        for (int64_t t = ns; t >= 8; t -= 8) {       // the chunk rounding below must not change the count
            const int64_t c = ceil_div64(ceil_div64(M, t), 64) * 64;
            if (ceil_div64(M, c) == t || t == 8) { ns = t; break; }
        }
    }
    *chunk = ceil_div64(ceil_div64(M, ns), 64) * 64;
    *nsplit = ceil_div64(M, *chunk);
}
static int tn_group_tiles(const b4c_tn_desc *d, int n) {
    int tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (int)(ceil_div64(d[i].K, TILE) * ceil_div64((int64_t)d[i].n_seg * d[i].seg_width, TILE));
    return tiles;
}

extern "C" int64_t b4c_gemm_tn_group_workspace_bytes(const b4c_tn_desc *h_desc, int n_prob, int M) {
    if (!h_desc || n_prob <= 0 || M <= 0) return 0;
    int64_t nsplit, chunk, bytes = 0;
    tn_group_plan(M, tn_group_tiles(h_desc, n_prob), &nsplit, &chunk);
    for (int i = 0; i < n_prob; ++i) {
        const int64_t tk = ceil_div64(h_desc[i].K, TILE), tn = ceil_div64((int64_t)h_desc[i].n_seg * h_desc[i].seg_width, TILE);
        bytes += nsplit * (tk * tn * 16384 + tn * 128) * 4;
    }
    return bytes;
}

extern "C" int b4c_gemm_tn_group(const b4c_tn_desc *h_desc, int n_prob, int M, int dtype, void *workspace, int64_t workspace_bytes,
                                 void *stream) {
    B4C_REQUIRE(h_desc && n_prob >= 1 && n_prob <= TN_GROUP_MAX && M > 0, "gemm_tn_group: n_prob %d (1..%d), M %d", n_prob, TN_GROUP_MAX, M);
    B4C_REQUIRE(dtype == B4C_BF16, "gemm_tn_group: bf16 only (dtype %d)", dtype);
    B4C_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= b4c_gemm_tn_group_workspace_bytes(h_desc, n_prob, M),
                "gemm_tn_group: workspace missing / too small");
    TNGroup g = {};
    g.np = n_prob;
    int64_t nsplit, chunk;
    tn_group_plan(M, tn_group_tiles(h_desc, n_prob), &nsplit, &chunk);
    float *ws = (float *)workspace;
    int reduce_blocks = 0;
    for (int i = 0; i < n_prob; ++i) {
        const b4c_tn_desc &d = h_desc[i];
        const int N = d.n_seg * d.seg_width;
        B4C_REQUIRE(d.A && d.G && d.K > 0 && N > 0 && d.n_seg >= 1 && d.n_seg <= 4 && d.lda >= d.K && d.ldg >= N, "gemm_tn_group: problem %d shape", i);
        B4C_REQUIRE(d.lda % 8 == 0 && d.ldg % 8 == 0 && ((((uintptr_t)d.A | (uintptr_t)d.G) & 15) == 0), "gemm_tn_group: problem %d needs 16-byte aligned operands, pitches %% 8 == 0", i);
        TNProb &p = g.p[i];
        p.A = (const bf16_t *)d.A; p.G = (const bf16_t *)d.G; p.lda = d.lda; p.ldg = d.ldg; p.K = d.K; p.N = N;
        p.tk = (int)ceil_div64(d.K, TILE); p.tn = (int)ceil_div64(N, TILE); p.tile0 = g.tiles;
        g.tiles += p.tk * p.tn;
        p.out = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}, d.seg_width, d.ldw, nsplit == 1 ? TN_DIRECT : TN_WS, nullptr, nullptr};
        for (int sgi = 0; sgi < d.n_seg; ++sgi) {
            B4C_REQUIRE(d.dW[sgi], "gemm_tn_group: problem %d segment %d has no dW", i, sgi);
            p.out.dW[sgi] = d.dW[sgi];
            p.out.db[sgi] = d.db[sgi];
        }
        B4C_REQUIRE(d.ldw >= d.seg_width, "gemm_tn_group: problem %d ldw", i);
        if (nsplit > 1) {
            p.out.ws = ws;
            p.out.ws_db = ws + nsplit * p.tk * p.tn * 16384;
            ws = p.out.ws_db + nsplit * p.tn * 128;
        }
        reduce_blocks += p.tk * p.tn * tn_reduce_bpt((int)nsplit) + p.tn;
    }
    hipStream_t st = (hipStream_t)stream;
    gemm_tn_bf16_group_kernel<2><<<dim3(g.tiles, 1, (unsigned)nsplit), 256, 2 * TN_TILE_BYTES, st>>>(g, M, chunk);
    if (nsplit > 1) tn_reduce_group_kernel<<<reduce_blocks, 1024, 0, st>>>(g, (int)nsplit);
    return b4c_check_launch("gemm_tn_group");
}
