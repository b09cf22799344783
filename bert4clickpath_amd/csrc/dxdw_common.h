// Pieces shared by the persistent backward kernels of the encoder's Dense layers (gemm_dxdw.hip, ffn_bwd.hip): the XOR-swizzled
// [32][128] bf16 LDS image of a token tile, its LDS-DMA request, the transposed MFMA fragment reads.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned dd_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned dd_u32x2;
typedef __attribute__((ext_vector_type(4))) short dd_s16x4;
typedef __attribute__((ext_vector_type(8))) short dd_s16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 dd_bf16x4;

#define DD_TOK 32                    // tokens per tile
#define DD_SUB (DD_TOK * 256)        // one [32][128] bf16 sub-tile: 8 KB
#define DD_RING 4                    // LDS stages: the tile in work + three on their way (one tile of cover leaves the memory
                                     // system idle while the workgroup computes and waits in turn: 2.7 TB/s measured)
#define DD_OSTR 272                  // bytes per staged dX row (256 + 16)

// 16-B chunk c of row j of a [rows][128] bf16 sub-tile sits at chunk c ^ swz(j): the direct 16-B fragment reads and the
// transposed 8-B reads both spread over the 64 banks (same image as csrc/vocab_ce.hip's VTile<128>)
__device__ __forceinline__ int dd_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int dd_chunk_off(int row, int chunk) { return row * 256 + ((chunk ^ dd_swz(row)) << 4); }
// transposed fragment piece of MFMA 32x32x16 (A or B operand: feature `32 dt + r`, tokens 8 hf + 0..7 of a 16-token step):
// the lane's address is token row 4 hf + (li >> 2) (+ 8 for the second piece), features 32 dt + 16 (g & 1) + 4 (li & 3)
__device__ __forceinline__ int dd_tr_off(int hf, int li, int g, int dt, int second) {
    const int row = 4 * hf + (li >> 2) + 8 * second;
    const int e = dt * 32 + 16 * (g & 1) + 4 * (li & 3);
    return dd_chunk_off(row, e >> 3) + (e & 7) * 2;
}
__device__ __forceinline__ bf16x8 dd_frag_tr(const char *p0, const char *p1) {
    const dd_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dd_s16x4 __attribute__((address_space(3))) *)(p0));
    const dd_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dd_s16x4 __attribute__((address_space(3))) *)(p1));
    const dd_s16x8 w = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, w);
}
// rows [tok0, tok0 + 32) x columns [c0, c0 + 128) of P (row pitch ld) -> LDS sub-tile at byte address lds_dst, this wave's share
// (one of the 8 wave instructions of 1 KiB: 4 rows); rows >= M arrive as zeros.
// Inline assembly, not __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler cannot tell the DMA's destination from the stages the
// loop's ds_reads address and drains the vector-memory counter right behind every request (s_waitcnt vmcnt(0): the four-stage ring
// ran as one stage).  The kernel orders a tile's arrival against its first read itself (counted s_waitcnt + barrier).
// `width_bytes` < 256: a tensor narrower than 128 columns (row pitch ld < 128): the image's columns past the width hold the next
// row's first entries, and zeros in the tile's last row -- the descriptor ends with the tile's last valid byte.
__device__ __forceinline__ void dd_dma(const bf16_t *__restrict__ P, int ld, int c0, int64_t tok0, int64_t M, unsigned lds_dst, int wave, int lane,
                                       int width_bytes = 256) {
    const int64_t left = M - tok0;
    const int64_t rows = left < 0 ? 0 : (left < DD_TOK ? left : DD_TOK);
    const int64_t bytes = rows > 0 ? (rows - 1) * (int64_t)ld * 2 + width_bytes : 0;
    const uint64_t base = (uint64_t)(P + tok0 * ld + c0);
    dd_u32x4 rs;
    rs[0] = __builtin_amdgcn_readfirstlane((unsigned)base);
    rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xFFFFu);
    rs[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    rs[3] = 0x00020000u;
    const int row = wave * 4 + (lane >> 4), slot = lane & 15;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst + (unsigned)(wave * 1024));
    const unsigned voff = (unsigned)((row * ld + ((slot ^ dd_swz(row)) << 3)) * 2);
    // (s_nop 0: a SALU write of M0 needs one wait state before an LDS-DMA reads it -- the compiler pads its own requests, it cannot
    // see into this one.  A write of M0 right BEHIND a 16-B-per-lane request does not reach it: scratch/m0_hazard.hip, 0 misplaced
    // pieces in 39 M with and without a backlog of loads in front.)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(dst), "v"(voff), "s"(rs) : "m0");
}

