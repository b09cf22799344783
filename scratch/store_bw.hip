// Write-only bandwidth of the materialised (R x V) bf16 logits by store shape: what one wave-instruction (64 lanes x 16 B)
// covers -- SEG bytes contiguous in each of 1024 / SEG rows -- and how a workgroup walks its tiles.  Tells whether the
// vocabulary projection's 4 rows x 256 B epilogue stores sit at the pattern's ceiling.
//   hipcc --offload-arch=gfx950 -O3 scratch/store_bw.hip -o scratch/bin/store_bw && scratch/bin/store_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// workgroup = 256 threads (4 waves); tile = TR rows x TC columns (bf16) = 32 KB; workgroup b walks tiles along N
template <int TR, int TC, bool NT>
__global__ void __launch_bounds__(256) fill_kernel(uint16_t *C, int64_t ld, int R, int V, int mt, int tiles_per_chunk) {
    const int tid = threadIdx.x;
    const int m0 = (blockIdx.x % mt) * TR;
    const int ntn = (V + TC - 1) / TC;
    const int nt0 = (blockIdx.x / mt) * tiles_per_chunk;
    const int nt1 = nt0 + tiles_per_chunk < ntn ? nt0 + tiles_per_chunk : ntn;
    constexpr int CPR = TC / 8;                 // 16-B chunks per tile row
    constexpr int NCH = TR * CPR / 256;         // chunks per thread
    for (int nt = nt0; nt < nt1; ++nt) {
        const int n0 = nt * TC;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = tid + q * 256;
            const int row = c / CPR, pc = c % CPR;
            const int64_t grow = m0 + row;
            const int gcol = n0 + pc * 8;
            if (grow < R && gcol < V) {
                const u32x4 w = {(unsigned)c, (unsigned)nt, 3u, 4u};
                if (NT) __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(C + grow * ld + gcol));
                else *reinterpret_cast<u32x4 *>(C + grow * ld + gcol) = w;
            }
        }
    }
}

template <int TR, int TC, bool NT> static void run(uint16_t *C, int R, int V, const char *name) {
    const int mt = (R + TR - 1) / TR, ntn = (V + TC - 1) / TC;
    int chunks = (2560 + mt - 1) / mt;
    if (chunks > ntn) chunks = ntn;
    if (chunks < 1) chunks = 1;
    const int tpc = (ntn + chunks - 1) / chunks;
    chunks = (ntn + tpc - 1) / tpc;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(a);
        fill_kernel<TR, TC, NT><<<mt * chunks, 256>>>(C, V, R, V, mt, tpc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it > 0 && ms < best) best = ms;
    }
    printf("%-34s %7.3f ms  %6.0f GB/s\n", name, best, (double)R * V * 2 / best / 1e6);
}

int main() {
    const int R = 40960, V = 50000;
    uint16_t *C;
    hipMalloc(&C, (size_t)R * V * 2);
    run<128, 128, false>(C, R, V, "128 x 128 tile (4 rows x 256 B)");
    run<128, 128, true>(C, R, V, "128 x 128 tile, nontemporal");
    run<64, 256, false>(C, R, V, "64 x 256 tile (2 rows x 512 B)");
    run<64, 256, true>(C, R, V, "64 x 256 tile, nontemporal");
    run<32, 512, false>(C, R, V, "32 x 512 tile (1 row x 1 KB)");
    run<32, 512, true>(C, R, V, "32 x 512 tile, nontemporal");
    run<16, 1024, true>(C, R, V, "16 x 1024 tile, nontemporal");
    hipFree(C);
    return 0;
}
