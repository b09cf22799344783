// Shared device/host helpers for libb4c_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/b4c.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define B4C_WAVE 64

// Workgroup barrier that orders LDS only.  __syncthreads() also makes every wave wait for ALL of its
// outstanding global loads and stores (s_waitcnt vmcnt(0)): inside a loop that stores results or keeps a
// prefetch in flight, that drains the memory pipeline once per iteration.  Use this where only LDS data is
// exchanged between the waves.
#define B4C_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

void b4c_set_error(const char *fmt, ...);
int b4c_check_launch(const char *what);

#define B4C_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            b4c_set_error(__VA_ARGS__);   \
            return B4C_EINVAL;            \
        }                                 \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- element IO: 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for fp32) ----
// Which streaming outputs use nontemporal stores (bit per site): measured per site on the C2 step.
#ifndef B4C_NT_MASK
#define B4C_NT_MASK 255
#endif
#define B4C_NT(site) (((B4C_NT_MASK) >> (site)) & 1)
#define B4C_NT_EMBED 0
#define B4C_NT_GEMM 1
#define B4C_NT_GEMMLN_Z 2
#define B4C_NT_GEMMLN_OUT 3
#define B4C_NT_LNBWD_DZ 4
#define B4C_NT_LNBWD_DY 5
#define B4C_NT_ATTN_O 6
#define B4C_NT_ATTN_DQKV 7
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(p + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[8]) {
        f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4 *>(p) = a;
        *reinterpret_cast<f32x4 *>(p + 4) = b;
    }
    // streaming store (nt bit): for outputs that are far larger than L2 and are next read by another kernel
    static __device__ __forceinline__ void store_nt(float *p, const float (&v)[8]) {
        f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        __builtin_nontemporal_store(a, reinterpret_cast<f32x4 *>(p));
        __builtin_nontemporal_store(b, reinterpret_cast<f32x4 *>(p + 4));
    }
    template <bool NT> static __device__ __forceinline__ void store_sel(float *p, const float (&v)[8]) {
        if (NT) store_nt(p, v); else store(p, v);
    }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t *p, float (&v)[8]) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void store(bf16_t *p, const float (&v)[8]) {
        bf16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
        *reinterpret_cast<bf16x8 *>(p) = a;
    }
    static __device__ __forceinline__ void store_nt(bf16_t *p, const float (&v)[8]) {
        bf16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
        __builtin_nontemporal_store(a, reinterpret_cast<bf16x8 *>(p));
    }
    template <bool NT> static __device__ __forceinline__ void store_sel(bf16_t *p, const float (&v)[8]) {
        if (NT) store_nt(p, v); else store(p, v);
    }
};

// ---- counter-based dropout mask ----
// Threefry-2x32 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), 12 rounds,
// key = seed, counter = e >> 2: 64 random bits = four 16-bit uniforms, one per element.  Adds, rotates and xors
// only: 32-bit integer multiplies run at quarter rate on CDNA, and the 64-bit-multiply mixer used first made the
// mask the most expensive part of every dropout site (0.9 ms of the 19.2 ms C2 step).
__host__ __device__ __forceinline__ uint32_t b4c_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__host__ __device__ __forceinline__ uint64_t b4c_rand64(uint64_t seed, uint64_t ctr) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), k2 = 0x1BD11BDAu ^ k0 ^ k1;
    uint32_t x0 = (uint32_t)ctr + k0, x1 = (uint32_t)(ctr >> 32) + k1;
#define B4C_TF_ROUND(R) x0 += x1; x1 = b4c_rotl32(x1, R); x1 ^= x0;
    B4C_TF_ROUND(13) B4C_TF_ROUND(15) B4C_TF_ROUND(26) B4C_TF_ROUND(6)
    x0 += k1; x1 += k2 + 1u;
    B4C_TF_ROUND(17) B4C_TF_ROUND(29) B4C_TF_ROUND(16) B4C_TF_ROUND(24)
    x0 += k2; x1 += k0 + 2u;
    B4C_TF_ROUND(13) B4C_TF_ROUND(15) B4C_TF_ROUND(26) B4C_TF_ROUND(6)
    x0 += k0; x1 += k1 + 3u;
#undef B4C_TF_ROUND
    return (uint64_t)x0 | ((uint64_t)x1 << 32);
}
// element e is kept iff its 16-bit uniform >= ceil(rate * 65536)
__host__ __device__ __forceinline__ uint32_t b4c_keep_threshold(float rate) {
    const float t = rate * 65536.0f;
    uint32_t thr = (uint32_t)t;
    if ((float)thr < t) ++thr;
    return thr;
}
__host__ __device__ __forceinline__ bool b4c_keep_elem(uint64_t seed, uint64_t e, float rate) {
    const uint64_t h = b4c_rand64(seed, e >> 2);
    return (uint32_t)((h >> (16 * (e & 3))) & 0xFFFFu) >= b4c_keep_threshold(rate);
}
// keep bits of the 8 consecutive elements e0 .. e0+7 (e0 % 4 == 0): bit k = element e0 + k.  Two hashes.
__host__ __device__ __forceinline__ uint32_t b4c_keep8(uint64_t seed, uint64_t e0, uint32_t thr) {
    const uint64_t h0 = b4c_rand64(seed, e0 >> 2), h1 = b4c_rand64(seed, (e0 >> 2) + 1);
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m |= (((uint32_t)(h0 >> (16 * k)) & 0xFFFFu) >= thr ? 1u : 0u) << k;
        m |= (((uint32_t)(h1 >> (16 * k)) & 0xFFFFu) >= thr ? 1u : 0u) << (4 + k);
    }
    return m;
}

// ---- wave / block reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// reduce over groups of G consecutive lanes (G power of two <= 64)
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
