// Does a write of M0 that follows a 16-B-per-lane LDS-DMA (buffer_load_dwordx4 ... lds) reach that request?  (gfx950)
// Every wave of a 512-thread workgroup per CU, NITER times: a backlog of ordinary global loads, then DMA A (M0 = region a)
// directly followed by DMA B (M0 = region b) with PAD wait states between, everything waited for, both regions compared with
// what they should hold (the source words carry iteration, region and lane), regions cleared.  Counts misplaced 16-B pieces per
// quarter of the wave.   usage: m0_hazard [pad 0|1|2 = none / s_nop 7 / 8 x s_nop 7] [backlog loads 0..16] [dword 0|1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define NITER 400
template <int PAD, int WIDE>
__global__ void __launch_bounds__(512, 1) k(const unsigned *src, const unsigned *junk, unsigned *errs, unsigned *sink, int backlog) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    constexpr int PIECE = WIDE ? 16 : 4;                  // bytes per lane
    // two regions per wave, 32 KB apart: a at wave * 1 KB, b at 32 KB + wave * 1 KB
    const unsigned ra = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024), rb = __builtin_amdgcn_readfirstlane(lds0 + 32768 + wave * 1024);
    unsigned acc = 0;
    unsigned e[4] = {0, 0, 0, 0};
    for (int it = 0; it < NITER; ++it) {
        // clear both regions
        *reinterpret_cast<u32x4 *>(smem + wave * 1024 + lane * 16) = (u32x4){0xdeadu, 0xdeadu, 0xdeadu, 0xdeadu};
        *reinterpret_cast<u32x4 *>(smem + 32768 + wave * 1024 + lane * 16) = (u32x4){0xdeadu, 0xdeadu, 0xdeadu, 0xdeadu};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // source blocks: [wg][it & 7][which][wave][lane] pieces
        const unsigned *pa = src + ((((size_t)blockIdx.x * 8 + (it & 7)) * 2 + 0) * 8 + wave) * 64 * 4;
        const unsigned *pb = src + ((((size_t)blockIdx.x * 8 + (it & 7)) * 2 + 1) * 8 + wave) * 64 * 4;
        u32x4 da, db;
        da[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)pa); da[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)pa >> 32) & 0xFFFFu); da[2] = 1024; da[3] = 0x00020000u;
        db[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)pb); db[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)pb >> 32) & 0xFFFFu); db[2] = 1024; db[3] = 0x00020000u;
        const unsigned voff = lane * 16;
        // backlog: ordinary loads of cold lines in front of the requests
        for (int b = 0; b < backlog; ++b) {
            unsigned v;
            const unsigned *p = junk + (((size_t)(blockIdx.x * NITER + it) * 16 + b) * 512 + tid) * 16 % (size_t)(1u << 28);
            asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p));
            acc += v;                                     // (never waited for before the end: the sum is garbage, only the traffic matters)
        }
        if (WIDE) {
            if (PAD == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, 0 offen lds" : : "s"(ra), "v"(voff), "s"(da), "s"(rb), "s"(db) : "m0");
            if (PAD == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_nop 7\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, 0 offen lds" : : "s"(ra), "v"(voff), "s"(da), "s"(rb), "s"(db) : "m0");
            if (PAD == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, 0 offen lds" : : "s"(ra), "v"(voff), "s"(da), "s"(rb), "s"(db) : "m0");
        } else {
            const unsigned v4 = lane * 4;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %4, 0 offen lds" : : "s"(ra), "v"(v4), "s"(da), "s"(rb), "s"(db) : "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // compare (own wave's regions only: no barrier needed)
        if (WIDE) {
            const u32x4 ga = *reinterpret_cast<const u32x4 *>(smem + wave * 1024 + lane * 16), gb = *reinterpret_cast<const u32x4 *>(smem + 32768 + wave * 1024 + lane * 16);
            const u32x4 wa = *reinterpret_cast<const u32x4 *>(pa + lane * 4), wb = *reinterpret_cast<const u32x4 *>(pb + lane * 4);
            const bool bad = ga[0] != wa[0] || ga[3] != wa[3] || gb[0] != wb[0] || gb[3] != wb[3];
            if (bad) e[lane >> 4] += 1;
        } else {
            const unsigned ga = *reinterpret_cast<const unsigned *>(smem + wave * 1024 + lane * 4), gb = *reinterpret_cast<const unsigned *>(smem + 32768 + wave * 1024 + lane * 4);
            if (ga != pa[lane] || gb != pb[lane]) e[lane >> 4] += 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int q = 0; q < 4; ++q) if (e[q]) atomicAdd(&errs[q], e[q]);
    if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char **argv) {
    const int pad = argc > 1 ? atoi(argv[1]) : 0, backlog = argc > 2 ? atoi(argv[2]) : 8, wide = argc > 3 ? atoi(argv[3]) : 1;
    const size_t nsrc = (size_t)256 * 8 * 2 * 8 * 64 * 4;
    std::vector<unsigned> h(nsrc);
    for (size_t i = 0; i < nsrc; ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
    unsigned *src, *junk, *errs, *sink;
    hipMalloc(&src, nsrc * 4); hipMalloc(&junk, (size_t)4 << 28); hipMalloc(&errs, 16); hipMalloc(&sink, 4);
    hipMemcpy(src, h.data(), nsrc * 4, hipMemcpyHostToDevice); hipMemset(errs, 0, 16); hipMemset(junk, 0, (size_t)4 << 28);
    hipFuncSetAttribute((const void *)k<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 3; ++rep) {
        if (!wide) k<0, 0><<<256, 512, 65536>>>(src, junk, errs, sink, backlog);
        else if (pad == 0) k<0, 1><<<256, 512, 65536>>>(src, junk, errs, sink, backlog);
        else if (pad == 1) k<1, 1><<<256, 512, 65536>>>(src, junk, errs, sink, backlog);
        else k<2, 1><<<256, 512, 65536>>>(src, junk, errs, sink, backlog);
    }
    hipError_t err = hipDeviceSynchronize();
    unsigned e[4];
    hipMemcpy(e, errs, 16, hipMemcpyDeviceToHost);
    printf("m0_hazard pad=%d backlog=%d %s: %s; misplaced or missing pieces per wave quarter: %u %u %u %u of %d per quarter\n", pad, backlog, wide ? "dwordx4" : "dword",
           hipGetErrorString(err), e[0], e[1], e[2], e[3], 3 * 256 * 8 * NITER * 16);
    return 0;
}
