/* The drop-in boundary without Python: a plain C program that links libb4c_hip.so (include/b4c.h) and the HIP
 * runtime, hands the library device pointers + sizes + a stream, and checks two entry points against loops written
 * here -- the [MASK]-position index generation (reference clickstream_transformer.py:260-297; integer work, compared
 * bit for bit) and residual + LayerNorm (reference transformer.py:204-206 with LayerNormalization(epsilon=1e-6), fp32).
 *
 *   hipcc -x c examples/c_abi/c_abi_smoke.c -Iinclude -Lbert4clickpath_amd -lb4c_hip -Wl,-rpath,$PWD/bert4clickpath_amd -o /tmp/c_abi_smoke
 *
 * (tests/test_gpu_c_abi.py builds and runs it on the GPU box.)  Exit code 0 = both checks passed. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "b4c.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_B4C(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, b4c_last_error()); return 3; } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rng_next(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static float rng_unit(void) { return (float)((rng_next() >> 40) / 16777216.0); }

int main(void) {
    printf("libb4c_hip ABI version %d\n", b4c_abi_version());
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));

    /* ---- 1. [MASK]-position index generation: counts, offsets (exclusive scan), row-major flat indices ---- */
    enum { B = 37, S = 203 };
    int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * B * S);
    for (int i = 0; i < B * S; ++i) ids[i] = (rng_next() % 100 < 7) ? 1 : 2 + (int64_t)(rng_next() % 48);
    for (int s = 0; s < S; ++s) { ids[4 * S + s] = 7; ids[9 * S + s] = 1; }        /* a row without and a row of only matches */
    int32_t want_counts[B], want_off[B + 1], *want_flat = (int32_t *)malloc(sizeof(int32_t) * B * S);
    int R = 0, want_max = 0;
    for (int b = 0; b < B; ++b) {
        want_off[b] = R;
        for (int s = 0; s < S; ++s)
            if (ids[b * S + s] == 1) want_flat[R++] = b * S + s;
        want_counts[b] = R - want_off[b];
        if (want_counts[b] > want_max) want_max = want_counts[b];
    }
    want_off[B] = R;
    int64_t *d_ids; int32_t *d_counts, *d_off, *d_flat, *d_max;
    CHECK_HIP(hipMalloc((void **)&d_ids, sizeof(int64_t) * B * S));
    CHECK_HIP(hipMalloc((void **)&d_counts, sizeof(int32_t) * B));
    CHECK_HIP(hipMalloc((void **)&d_off, sizeof(int32_t) * (B + 1)));
    CHECK_HIP(hipMalloc((void **)&d_flat, sizeof(int32_t) * B * S));
    CHECK_HIP(hipMalloc((void **)&d_max, sizeof(int32_t)));
    CHECK_HIP(hipMemcpy(d_ids, ids, sizeof(int64_t) * B * S, hipMemcpyHostToDevice));
    CHECK_B4C(b4c_mask_positions(d_ids, B, S, 1, d_counts, d_off, d_flat, B * S, d_max, NULL, st));
    int32_t got_counts[B], got_off[B + 1], got_max, *got_flat = (int32_t *)malloc(sizeof(int32_t) * B * S);
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(got_counts, d_counts, sizeof(got_counts), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(got_off, d_off, sizeof(got_off), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(got_flat, d_flat, sizeof(int32_t) * B * S, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&got_max, d_max, sizeof(int32_t), hipMemcpyDeviceToHost));
    int bad = got_max != want_max;
    for (int b = 0; b < B; ++b) bad |= got_counts[b] != want_counts[b];
    for (int b = 0; b <= B; ++b) bad |= got_off[b] != want_off[b];
    for (int i = 0; i < R; ++i) bad |= got_flat[i] != want_flat[i];
    printf("mask_positions: %d matches in %d x %d ids, longest row %d: %s\n", R, B, S, got_max, bad ? "MISMATCH" : "bit-exact");
    if (bad) return 1;

    /* ---- 2. z = x + y; out = LayerNorm(z) * gamma + beta (biased variance, eps inside the rsqrt), fp32 ---- */
    enum { ROWS = 301, D = 128 };
    float *x = (float *)malloc(sizeof(float) * ROWS * D), *y = (float *)malloc(sizeof(float) * ROWS * D);
    float gamma[D], beta[D];
    for (int i = 0; i < ROWS * D; ++i) { x[i] = 2.f * rng_unit() - 1.f; y[i] = 0.5f * (2.f * rng_unit() - 1.f); }
    for (int j = 0; j < D; ++j) { gamma[j] = 0.5f + rng_unit(); beta[j] = 0.2f * (2.f * rng_unit() - 1.f); }
    float *d_x, *d_y, *d_g, *d_b, *d_z, *d_o, *d_s;
    CHECK_HIP(hipMalloc((void **)&d_x, sizeof(float) * ROWS * D));
    CHECK_HIP(hipMalloc((void **)&d_y, sizeof(float) * ROWS * D));
    CHECK_HIP(hipMalloc((void **)&d_z, sizeof(float) * ROWS * D));
    CHECK_HIP(hipMalloc((void **)&d_o, sizeof(float) * ROWS * D));
    CHECK_HIP(hipMalloc((void **)&d_s, sizeof(float) * ROWS * 2));
    CHECK_HIP(hipMalloc((void **)&d_g, sizeof(gamma)));
    CHECK_HIP(hipMalloc((void **)&d_b, sizeof(beta)));
    CHECK_HIP(hipMemcpy(d_x, x, sizeof(float) * ROWS * D, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_y, y, sizeof(float) * ROWS * D, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_g, gamma, sizeof(gamma), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_b, beta, sizeof(beta), hipMemcpyHostToDevice));
    CHECK_B4C(b4c_add_dropout_layernorm_fwd(d_x, d_y, d_g, d_b, d_z, d_o, d_s, ROWS, D, 1e-6f, 0.f, 0, B4C_F32, st));
    float *out = (float *)malloc(sizeof(float) * ROWS * D);
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(out, d_o, sizeof(float) * ROWS * D, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int r = 0; r < ROWS; ++r) {
        double mean = 0.0, var = 0.0;
        for (int j = 0; j < D; ++j) mean += (double)x[r * D + j] + (double)y[r * D + j];
        mean /= D;
        for (int j = 0; j < D; ++j) { const double c = (double)x[r * D + j] + (double)y[r * D + j] - mean; var += c * c; }
        const double rstd = 1.0 / sqrt(var / D + 1e-6);
        for (int j = 0; j < D; ++j) {
            const double want = ((double)x[r * D + j] + (double)y[r * D + j] - mean) * rstd * gamma[j] + beta[j];
            const double err = fabs(want - (double)out[r * D + j]);
            if (err > worst) worst = err;
        }
    }
    printf("add + LayerNorm (fp32, %d x %d): largest deviation from the fp64 loop %.3g (bound 1e-5)\n", ROWS, D, worst);
    if (!(worst < 1e-5)) return 1;
    /* a bad argument comes back as an error code with a message, not as a crash */
    if (b4c_add_dropout_layernorm_fwd(d_x, d_y, d_g, d_b, d_z, d_o, d_s, ROWS, 12, 1e-6f, 0.f, 0, B4C_F32, st) == 0) {
        fprintf(stderr, "d = 12 (not a multiple of 8) was accepted\n");
        return 1;
    }
    printf("rejected call says: %s\n", b4c_last_error());
    return 0;
}
