// buffer_load ... lds semantics check on gfx950: lane l of a wave writes its 16 B to LDS at (M0 base + 16 l);
// the per-lane voffset chooses the SOURCE.  Out-of-range sources read as zero.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const unsigned short *src, unsigned *out, int n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(src), 0, (unsigned)n * 2, 0x00020000);
    const int wave = tid >> 6, lane = tid & 63;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(smem + wave * 1024), 16, (lane ^ 1) * 16 + wave * 1024, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[tid] = reinterpret_cast<unsigned *>(smem)[tid * 4];
}
int main() {
    const int n = 256 * 8 - 64;   // last 8 chunks are out of range
    std::vector<unsigned short> h(256 * 8);
    for (int i = 0; i < 256 * 8; ++i) h[i] = (unsigned short)i;
    unsigned short *d; unsigned *o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    k<<<1, 256, 4096>>>(d, o, n);
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const int chunk = (t & ~63) + ((t & 63) ^ 1);
        const unsigned e0 = chunk * 8, expect = (chunk * 8 < n) ? (e0 | ((e0 + 1) << 16)) : 0u;
        if (r[t] != expect) { if (bad < 5) printf("tid %d got %08x expect %08x\n", t, r[t], expect); ++bad; }
    }
    printf("dma_test: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
