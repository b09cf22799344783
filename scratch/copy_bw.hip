// Achievable HBM bandwidth of a read+write stream on MI355X under different access recipes (scratch experiment).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int MODE, int UNROLL>
__global__ void __launch_bounds__(256) copyk(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, u32x4 *__restrict__ c, size_t n) {
    // MODE 0: c = a (1R 1W); 1: c = a ^ b (2R 1W); +2: nontemporal stores; +4: nontemporal loads
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (MODE & 4) v[u] = __builtin_nontemporal_load(a + i + u * stride); else v[u] = a[i + u * stride];
            if (MODE & 1) { u32x4 w = (MODE & 4) ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride]; v[u] ^= w; }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (MODE & 2) __builtin_nontemporal_store(v[u], c + i + u * stride); else c[i + u * stride] = v[u];
        }
    }
}
template <int MODE, int UNROLL> void run(const char *name, u32x4 *a, u32x4 *b, u32x4 *c, size_t n, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) copyk<MODE, UNROLL><<<grid, 256>>>(a, b, c, n);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) copyk<MODE, UNROLL><<<grid, 256>>>(a, b, c, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n * 16 * ((MODE & 1) ? 3 : 2);
    printf("%-44s grid %6d  %7.1f us  %6.0f GB/s\n", name, grid, ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e9);
}
int main() {
    const size_t n = (size_t)(getenv("COPY_MB") ? atol(getenv("COPY_MB")) : 210) * 1000000 / 16;   // bytes per tensor
    u32x4 *a, *b, *c;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16);
    hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16);
    for (int grid : {1024, 2048, 4096, 16384}) {
        run<0, 4>("1R1W plain u4", a, b, c, n, grid);
        run<2, 4>("1R1W nt-store u4", a, b, c, n, grid);
        run<6, 4>("1R1W nt-load nt-store u4", a, b, c, n, grid);
        run<1, 4>("2R1W plain u4", a, b, c, n, grid);
        run<3, 4>("2R1W nt-store u4", a, b, c, n, grid);
        run<7, 4>("2R1W nt-load nt-store u4", a, b, c, n, grid);
    }
    run<0, 8>("1R1W plain u8", a, b, c, n, 2048);
    run<2, 8>("1R1W nt-store u8", a, b, c, n, 2048);
    run<0, 1>("1R1W plain u1", a, b, c, n, 51200);
    run<2, 1>("1R1W nt-store u1 (one elem per thread)", a, b, c, n, (int)(n / 256));
    return 0;
}
