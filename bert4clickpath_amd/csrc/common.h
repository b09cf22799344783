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
};

// ---- counter-based dropout mask (splitmix64 finaliser; two 24-bit uniforms per hash) ----
__host__ __device__ __forceinline__ uint64_t b4c_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// keep element e?  u in [0,1) with 24 bits; keep iff u >= rate.
__host__ __device__ __forceinline__ bool b4c_keep_elem(uint64_t seed, uint64_t e, float rate) {
    const uint64_t h = b4c_mix64(seed + ((e >> 1) + 1) * 0x9E3779B97F4A7C15ULL);
    const uint32_t bits = (e & 1) ? (uint32_t)(h >> 40) : (uint32_t)((h >> 8) & 0xFFFFFFu);
    return (float)bits * (1.0f / 16777216.0f) >= rate;
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
