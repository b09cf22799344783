// Is vmcnt retired in issue order across operation kinds on gfx950?  Per wave, NITER times:
//   1. a SLOW load (a cold line far away, new page every iteration) into a register preset with a sentinel,
//   2. a FAST younger operation of another kind (a store to a hot line / an LDS-DMA of a hot line / a load of a hot line),
//   3. s_waitcnt vmcnt(1)  -- "all but the youngest one are done": the slow load must have landed,
//   4. is the register still the sentinel?  (counted per kind)
// usage: vmcnt_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define NITER 2000
template <int KIND>     // 0 store, 1 LDS-DMA dwordx4, 2 load
__global__ void __launch_bounds__(512, 1) k(const unsigned *cold, size_t cold_words, unsigned *hot, unsigned *errs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem + wave * 1024);
    unsigned *myhot = hot + ((size_t)blockIdx.x * 512 + tid) * 4;
    u32x4 dh;
    dh[0] = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(hot + (size_t)blockIdx.x * 2048 + wave * 256)); dh[1] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)(hot + (size_t)blockIdx.x * 2048 + wave * 256) >> 32) & 0xFFFFu); dh[2] = 1024; dh[3] = 0x00020000u;
    unsigned bad = 0, sink = 0;
    size_t pos = ((size_t)blockIdx.x * 512 + tid) * 1031 % cold_words;
    for (int it = 0; it < NITER; ++it) {
        pos = (pos + (size_t)524309 * 4) % cold_words;             // ~2 MB further every time: a cold line on another page
        const unsigned *p = cold + pos;
        unsigned v = 0xFFFFFFFFu, w = 0;
        asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(p));
        if (KIND == 0) asm volatile("global_store_dword %0, %1, off" : : "v"(myhot), "v"(it) : "memory");
        if (KIND == 1) { const unsigned voff = lane * 16; asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds0), "v"(voff), "s"(dh) : "m0"); }
        if (KIND == 2) asm volatile("global_load_dword %0, %1, off" : "+v"(w) : "v"(myhot));
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(v) : : "memory");
        if (v == 0xFFFFFFFFu) ++bad;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v), "+v"(w) : : "memory");
        sink += v + w;
    }
    if (bad) atomicAdd(&errs[KIND], bad);
    if (sink == 0x12345u) errs[3] = sink;
}
int main() {
    const size_t cold_words = (size_t)3 << 28;            // 3 Gi words = 12 GB (cold data holds zeros: never the sentinel)
    unsigned *cold, *hot, *errs;
    if (hipMalloc(&cold, cold_words * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&hot, (size_t)256 * 2048 * 4 * 4); hipMalloc(&errs, 16);
    hipMemset(cold, 0, cold_words * 4); hipMemset(hot, 0, (size_t)256 * 2048 * 4 * 4); hipMemset(errs, 0, 16);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192);
    k<0><<<256, 512, 8192>>>(cold, cold_words, hot, errs);
    k<1><<<256, 512, 8192>>>(cold, cold_words, hot, errs);
    k<2><<<256, 512, 8192>>>(cold, cold_words, hot, errs);
    hipError_t e = hipDeviceSynchronize();
    unsigned h[4]; hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost);
    printf("vmcnt_order (%s): slow load still pending after s_waitcnt vmcnt(1) behind a younger  store: %u   LDS-DMA: %u   load: %u   (of %d checks each)\n",
           hipGetErrorString(e), h[0], h[1], h[2], 256 * 512 * NITER);
    return 0;
}
