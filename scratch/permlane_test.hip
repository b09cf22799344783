// v_permlane32_swap with BOTH operands the same value: what do the two results hold?  (scratch; build: hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    const unsigned lane = threadIdx.x;
    unsigned tm = 100 + lane;
    const auto sw = __builtin_amdgcn_permlane32_swap(tm, tm, false, false);
    out[lane * 2] = sw[0];
    out[lane * 2 + 1] = sw[1];
    // the form with two DIFFERENT registers holding the same value
    unsigned a = 100 + lane, b = 100 + lane;
    asm volatile("v_mov_b32 %0, %0" : "+v"(b));
    const auto s2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + lane * 2] = s2[0];
    out[128 + lane * 2 + 1] = s2[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 5, 31, 32, 37, 63}) printf("lane %2d: same-operand (%u, %u)   two registers (%u, %u)\n", l, h[l * 2], h[l * 2 + 1], h[128 + l * 2], h[128 + l * 2 + 1]);
    return 0;
}
