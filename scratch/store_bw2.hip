// Write-only bandwidth of the (R x V) bf16 logits as a function of (a) the row pitch (V = 50,000 elements: 100,000 B rows
// start 32 B further into a 128-B line each; 50,048: every row 256-B aligned), (b) how many waves per CU issue the
// stores (the projection kernel has 4 store waves per CU; store_bw.hip had 32) and (c) the tile shape.
//   hipcc --offload-arch=gfx950 -O3 scratch/store_bw2.hip -o scratch/bin/store_bw2 && scratch/bin/store_bw2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int TR, int TC, int NTHR>
__global__ void __launch_bounds__(NTHR) fill_kernel(uint16_t *C, int64_t ld, int R, int V, int mt, int tiles_per_chunk) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x;
    const int m0 = (blockIdx.x % mt) * TR;
    const int ntn = (V + TC - 1) / TC;
    const int nt0 = (blockIdx.x / mt) * tiles_per_chunk;
    const int nt1 = nt0 + tiles_per_chunk < ntn ? nt0 + tiles_per_chunk : ntn;
    constexpr int CPR = TC / 8;                 // 16-B chunks per tile row
    constexpr int NCH = TR * CPR / NTHR;        // chunks per thread
    if (tid == 0 && V < 0) smem[0] = 1;
    for (int nt = nt0; nt < nt1; ++nt) {
        const int n0 = nt * TC;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = tid + q * NTHR;
            const int row = c / CPR, pc = c % CPR;
            const int64_t grow = m0 + row;
            const int gcol = n0 + pc * 8;
            if (grow < R && gcol < V) {
                const u32x4 w = {(unsigned)c, (unsigned)nt, 3u, 4u};
                *reinterpret_cast<u32x4 *>(C + grow * ld + gcol) = w;
            }
        }
    }
}

template <int TR, int TC, int NTHR> static void run(uint16_t *C, int R, int V, int64_t ld, int wg_per_cu, const char *name) {
    const int mt = (R + TR - 1) / TR, ntn = (V + TC - 1) / TC;
    int chunks = (256 * wg_per_cu * 5 + mt - 1) / mt;
    if (chunks > ntn) chunks = ntn;
    if (chunks < 1) chunks = 1;
    const int tpc = (ntn + chunks - 1) / chunks;
    chunks = (ntn + tpc - 1) / tpc;
    const size_t shm = 160 * 1024 / wg_per_cu - 1024;
    hipFuncSetAttribute((const void *)fill_kernel<TR, TC, NTHR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(a);
        fill_kernel<TR, TC, NTHR><<<mt * chunks, NTHR, shm>>>(C, ld, R, V, mt, tpc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it > 0 && ms < best) best = ms;
    }
    printf("%-26s ld %6ld  %2d wg/CU x %3d thr  %7.3f ms  %6.0f GB/s\n", name, (long)ld, wg_per_cu, NTHR, best, (double)R * V * 2 / best / 1e6);
}

int main() {
    const int R = 40960, V = 50000;
    uint16_t *C;
    hipMalloc(&C, (size_t)R * 50176 * 2);
    for (int64_t ld : {50000, 50048, 50176}) {
        for (int w : {1, 2, 4, 8}) {
            run<128, 128, 256>(C, R, V, ld, w, "128 x 128");
            run<64, 256, 256>(C, R, V, ld, w, "64 x 256");
            run<32, 512, 256>(C, R, V, ld, w, "32 x 512");
        }
        run<128, 128, 512>(C, R, V, ld, 1, "128 x 128");
        run<64, 256, 512>(C, R, V, ld, 1, "64 x 256");
        run<128, 128, 1024>(C, R, V, ld, 1, "128 x 128");
        run<64, 256, 1024>(C, R, V, ld, 1, "64 x 256");
    }
    hipFree(C);
    return 0;
}
