// Vocabulary-wide row kernels of the MLM head: softmax, masked sparse cross-entropy (the
// reference's clip -> log -> log-softmax form, and the fused training form that turns logits into
// their gradient in place), top-k item ids with HitRate / NDCG terms.
// One workgroup per row; a row of V <= 65536 logits lives in registers (8 x 8 x 1024 lanes) for
// the fused kernel, so HBM sees each logit once in and once out.
#include <math.h>

#include "common.h"

#define KERAS_EPS 1e-7f

// block-wide reductions over NW waves; result broadcast to every thread.  `buf` >= NW floats.
template <int NW> __device__ __forceinline__ float block_sum(float v, float *buf) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += buf[i];
    return t;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float *buf) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
#pragma unroll
    for (int i = 0; i < NW; ++i) t = fmaxf(t, buf[i]);
    return t;
}

// ------------------------------------------------------------------------------------------
// softmax over V (materialised probabilities: the drop-in surface of SoftMaxHead)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const T *__restrict__ x, int ld_in, T *__restrict__ y, int ld_out,
                                                           int64_t R, int V) {
    __shared__ float buf[4];
    const int tid = threadIdx.x;
    const int nch = (V + 7) >> 3;
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const T *xr = x + row * ld_in;
        T *yr = y + row * ld_out;
        float m = -INFINITY;
        for (int c = tid; c < nch; c += 256) {
            float v[8];
            Vec8<T>::load(xr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c * 8 + k < V) m = fmaxf(m, v[k]);
        }
        m = block_max<4>(m, buf);
        float s = 0.f;
        for (int c = tid; c < nch; c += 256) {
            float v[8];
            Vec8<T>::load(xr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c * 8 + k < V) s += expf(v[k] - m);
        }
        s = block_sum<4>(s, buf);
        const int nch_out = ld_out >> 3;
        for (int c = tid; c < nch_out; c += 256) {
            float v[8];
            if (c < nch) Vec8<T>::load(xr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (c * 8 + k < V) ? expf(v[k] - m) / s : 0.f;
            Vec8<T>::store(yr + c * 8, v);
        }
    }
}

// bf16, V <= 65536: the row stays in registers as raw bf16 pairs (NCH x 16 B per thread, 512 threads), so HBM sees
// every logit once in and every probability once out (the streaming kernel above reads the row three times).
template <int NCH>
__global__ void __launch_bounds__(512, 4) softmax_rows_bf16_kernel(const bf16_t *__restrict__ x, int ld_in, bf16_t *__restrict__ y,
                                                                   int ld_out, int64_t R, int V) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    __shared__ float buf[8];
    const int tid = threadIdx.x;
    const int nch_in = (V + 7) >> 3, nch_out = ld_out >> 3;
    const float LOG2E = 1.4426950408889634f;
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const bf16_t *xr = x + row * ld_in;
        bf16_t *yr = y + row * ld_out;
        u32x4 raw[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 512 + tid;
            raw[i] = (c < nch_in) ? *reinterpret_cast<const u32x4 *>(xr + c * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
#define SM_ELEM(i, k) __uint_as_float((k & 1) ? (raw[i][k >> 1] & 0xFFFF0000u) : (raw[i][k >> 1] << 16))
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int base = (i * 512 + tid) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (base + k < V) mx = fmaxf(mx, SM_ELEM(i, k));
        }
        mx = wave_max(mx);
        __syncthreads();
        if ((tid & 63) == 0) buf[tid >> 6] = mx;
        __syncthreads();
        mx = buf[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) mx = fmaxf(mx, buf[w]);
        const float mb = mx * LOG2E;
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int base = (i * 512 + tid) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (base + k < V) z += __builtin_amdgcn_exp2f(SM_ELEM(i, k) * LOG2E - mb);
        }
        z = wave_sum(z);
        __syncthreads();
        if ((tid & 63) == 0) buf[tid >> 6] = z;
        __syncthreads();
        z = buf[0] + buf[1] + buf[2] + buf[3] + buf[4] + buf[5] + buf[6] + buf[7];
        const float invz = 1.0f / z;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 512 + tid;
            const int base = c * 8;
            if (c < nch_out) {
                float p[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) p[k] = (base + k < V) ? __builtin_amdgcn_exp2f(SM_ELEM(i, k) * LOG2E - mb) * invz : 0.f;
                Vec8<bf16_t>::store_nt(yr + base, p);
            }
        }
#undef SM_ELEM
        __syncthreads();
    }
}

extern "C" int b4c_softmax_rows(const void *logits, int ld_in, void *probs, int ld_out, int64_t R, int V, int dtype,
                                void *stream) {
    B4C_REQUIRE(logits && probs && R >= 0 && V > 0, "softmax_rows: bad argument");
    B4C_REQUIRE(ld_in % 8 == 0 && ld_out % 8 == 0 && ld_in >= V && ld_out >= V, "softmax_rows: pitches must be multiples of 8 and >= V");
    if (R == 0) return B4C_OK;
    const int grid = (int)(R < 4096 ? R : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_BF16 && ld_out <= 8 * 16 * 512) {
        const int nch = (int)ceil_div64(ld_out / 8, 512);
        const int g5 = (int)(R < 3072 ? R : 3072);
#define SM16_LAUNCH(N) softmax_rows_bf16_kernel<N><<<g5, 512, 0, st>>>((const bf16_t *)logits, ld_in, (bf16_t *)probs, ld_out, R, V)
        if (nch <= 1) SM16_LAUNCH(1);
        else if (nch <= 2) SM16_LAUNCH(2);
        else if (nch <= 4) SM16_LAUNCH(4);
        else if (nch <= 8) SM16_LAUNCH(8);
        else if (nch <= 13) SM16_LAUNCH(13);
        else SM16_LAUNCH(16);
#undef SM16_LAUNCH
        return b4c_check_launch("softmax_rows_bf16");
    }
    if (dtype == B4C_F32) softmax_rows_kernel<float><<<grid, 256, 0, st>>>((const float *)logits, ld_in, (float *)probs, ld_out, R, V);
    else if (dtype == B4C_BF16) softmax_rows_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t *)logits, ld_in, (bf16_t *)probs, ld_out, R, V);
    else B4C_REQUIRE(false, "softmax_rows: dtype %d", dtype);
    return b4c_check_launch("softmax_rows");
}

// ------------------------------------------------------------------------------------------
// masked sparse CE on probabilities (reference dataflow)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) sparse_ce_probs_kernel(const T *__restrict__ p, int ld, const float *__restrict__ labels,
                                                              float *__restrict__ item_loss, float *__restrict__ n_valid,
                                                              int64_t R, int V, int variant) {
    __shared__ float buf[4];
    const int tid = threadIdx.x;
    const int nch = (V + 7) >> 3;
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const float lab = labels[row];
        if (lab == -1.0f) {   // LABEL_PAD: masked out, not counted (losses.py:50-51, 69)
            if (tid == 0) item_loss[row] = 0.f;
            continue;
        }
        const int y = (int)lab;
        if (y < 0 || y >= V) {  // TF: InvalidArgument on CPU / NaN on GPU
            if (tid == 0) { item_loss[row] = NAN; atomicAdd(n_valid, 1.0f); }
            continue;
        }
        const T *pr = p + row * ld;
        float loss;
        if (variant == B4C_CE_PLAIN) {
            loss = -logf((float)pr[y]);
        } else {
            float s = 0.f;
            for (int c = tid; c < nch; c += 256) {
                float v[8];
                Vec8<T>::load(pr + c * 8, v);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c * 8 + k < V) s += fminf(fmaxf(v[k], KERAS_EPS), 1.0f - KERAS_EPS);
            }
            s = block_sum<4>(s, buf);
            const float py = fminf(fmaxf((float)pr[y], KERAS_EPS), 1.0f - KERAS_EPS);
            loss = logf(s) - logf(py);
        }
        if (tid == 0) { item_loss[row] = loss; atomicAdd(n_valid, 1.0f); }
    }
}

extern "C" int b4c_sparse_ce_from_probs(const void *probs, int ld, const float *labels, float *item_loss,
                                        float *n_valid, int64_t R, int V, int variant, int dtype, void *stream) {
    B4C_REQUIRE(probs && labels && item_loss && n_valid && R >= 0 && V > 0, "sparse_ce_from_probs: bad argument");
    B4C_REQUIRE(ld % 8 == 0 && ld >= V, "sparse_ce_from_probs: pitch");
    B4C_REQUIRE(variant == B4C_CE_TF || variant == B4C_CE_PLAIN, "sparse_ce_from_probs: variant %d", variant);
    if (R == 0) return B4C_OK;
    const int grid = (int)(R < 4096 ? R : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_F32) sparse_ce_probs_kernel<float><<<grid, 256, 0, st>>>((const float *)probs, ld, labels, item_loss, n_valid, R, V, variant);
    else if (dtype == B4C_BF16) sparse_ce_probs_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t *)probs, ld, labels, item_loss, n_valid, R, V, variant);
    else B4C_REQUIRE(false, "sparse_ce_from_probs: dtype %d", dtype);
    return b4c_check_launch("sparse_ce_from_probs");
}

// ------------------------------------------------------------------------------------------
// fused training form: logits -> loss, dlogits in place.  1024 threads / row, row in registers.
//   p = softmax(x);  tf variant (loss = log sum_j clip(p_j) - log clip(p_y)):
//     u_j = [eps <= p_j <= 1-eps],  S = sum clip(p),  Pu = sum u p,
//     dL/dp_j = u_j (1/S - [j=y]/clip(p_y)),   G = sum_j p_j dL/dp_j = Pu/S - u_y p_y/clip(p_y)
//     dL/dx_j = p_j (dL/dp_j - G)
//   plain variant: dL/dx_j = p_j - [j=y]
// ------------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ void __launch_bounds__(1024) softmax_ce_fused_kernel(T *__restrict__ x, int ld, const int32_t *__restrict__ labels,
                                                                float *__restrict__ item_loss, const float *__restrict__ grad_scale,
                                                                int64_t R, int V, int variant) {
    __shared__ float buf[16];
    __shared__ float s_py;
    const int tid = threadIdx.x;
    const int nch_ld = ld >> 3;
    const float gs = grad_scale[0];
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        T *xr = x + row * ld;
        const int y = labels[row];
        const bool valid = y >= 0 && y < V;
        float v[NCH][8];
        if (!valid) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = i * 1024 + tid;
                if (c < nch_ld) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[i][k] = 0.f;
                    Vec8<T>::store(xr + c * 8, v[i]);
                }
            }
            if (tid == 0) item_loss[row] = (y >= V) ? NAN : 0.f;
            continue;
        }
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 1024 + tid;
            if (c < nch_ld) {
                Vec8<T>::load(xr + c * 8, v[i]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (c * 8 + k >= V) v[i][k] = -INFINITY;
                    m = fmaxf(m, v[i][k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[i][k] = -INFINITY;
            }
        }
        m = block_max<16>(m, buf);
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[i][k] = expf(v[i][k] - m);   // exp(-inf) = 0 for pad columns
                z += v[i][k];
            }
        z = block_sum<16>(z, buf);
        const float invz = 1.0f / z;
        float S = 0.f, Pu = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 1024 + tid;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float p = v[i][k] * invz;
                v[i][k] = p;
                const int j = c * 8 + k;
                if (j < V) {
                    S += fminf(fmaxf(p, KERAS_EPS), 1.0f - KERAS_EPS);
                    if (p >= KERAS_EPS && p <= 1.0f - KERAS_EPS) Pu += p;
                    if (j == y) s_py = p;
                }
            }
        }
        if (variant == B4C_CE_TF) {
            S = block_sum<16>(S, buf);
            Pu = block_sum<16>(Pu, buf);
        } else {
            __syncthreads();
        }
        const float py = s_py;
        const float pyc = fminf(fmaxf(py, KERAS_EPS), 1.0f - KERAS_EPS);
        const float uy = (py >= KERAS_EPS && py <= 1.0f - KERAS_EPS) ? 1.f : 0.f;
        float loss, G = 0.f, invS = 0.f;
        if (variant == B4C_CE_TF) {
            loss = logf(S) - logf(pyc);
            invS = 1.0f / S;
            G = Pu * invS - uy * py / pyc;
        } else {
            loss = -logf(py);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 1024 + tid;
            if (c < nch_ld) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int j = c * 8 + k;
                    const float p = v[i][k];
                    float g;
                    if (variant == B4C_CE_TF) {
                        const float u = (p >= KERAS_EPS && p <= 1.0f - KERAS_EPS) ? 1.f : 0.f;
                        g = p * (u * (invS - (j == y ? 1.0f / pyc : 0.f)) - G);
                    } else {
                        g = p - (j == y ? 1.f : 0.f);
                    }
                    v[i][k] = (j < V) ? g * gs : 0.f;
                }
                Vec8<T>::store(xr + c * 8, v[i]);
            }
        }
        if (tid == 0) item_loss[row] = loss;
        __syncthreads();  // s_py / buf reuse by the next row
    }
}

// bf16 throughput form: 512 threads per row, the row stays in registers as RAW bf16 pairs (NCH x 16 B
// per thread, <= 128 VGPRs so 2 workgroups share a CU and their load / compute / store phases overlap), exponentials
// are recomputed per pass with v_exp_f32.  When no probability leaves [1e-7, 1-1e-7] the clip is the
// identity (S = Pu = 1, G = 0) and the TF form reduces to p - onehot: that row-uniform fast path skips
// the S / Pu pass.
template <int NCH>
__global__ void __launch_bounds__(512, 4) softmax_ce_bf16_kernel(bf16_t *__restrict__ x, int ld, const int32_t *__restrict__ labels,
                                                              float *__restrict__ item_loss, const float *__restrict__ grad_scale,
                                                              int64_t R, int V, int variant) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    __shared__ float buf[8];
    __shared__ float buf2[8];
    __shared__ float s_ey;
    const int tid = threadIdx.x;
    const int nch_ld = ld >> 3;
    const float gs = grad_scale[0];
    const float LOG2E = 1.4426950408889634f;
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        bf16_t *xr = x + row * ld;
        const int y = labels[row];
        const bool valid = y >= 0 && y < V;
        u32x4 raw[NCH];
        if (!valid) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = i * 512 + tid;
                if (c < nch_ld) *reinterpret_cast<u32x4 *>(xr + c * 8) = (u32x4){0u, 0u, 0u, 0u};
            }
            if (tid == 0) item_loss[row] = (y >= V) ? NAN : 0.f;
            continue;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 512 + tid;
            raw[i] = (c < nch_ld) ? *reinterpret_cast<const u32x4 *>(xr + c * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
        // element k of chunk i: dword k>>1, low half for even k
#define CE_ELEM(i, k) __uint_as_float((k & 1) ? (raw[i][k >> 1] & 0xFFFF0000u) : (raw[i][k >> 1] << 16))
        float mx = -INFINITY, mn = INFINITY;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int base = (i * 512 + tid) * 8;
            if (base + 8 <= V) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float v = CE_ELEM(i, k); mx = fmaxf(mx, v); mn = fminf(mn, v); }
            } else if (base < V) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (base + k < V) { const float v = CE_ELEM(i, k); mx = fmaxf(mx, v); mn = fminf(mn, v); }
            }
        }
        mx = wave_max(mx);
        mn = -wave_max(-mn);
        __syncthreads();
        if ((tid & 63) == 0) { buf[tid >> 6] = mx; buf2[tid >> 6] = mn; }
        __syncthreads();
        mx = buf[0]; mn = buf2[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) { mx = fmaxf(mx, buf[w]); mn = fminf(mn, buf2[w]); }
        const float mb = mx * LOG2E;
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int base = (i * 512 + tid) * 8;
            if (base < V) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float e = __builtin_amdgcn_exp2f(CE_ELEM(i, k) * LOG2E - mb);
                    if (base + 8 <= V || base + k < V) {
                        z += e;
                        if (base + k == y) s_ey = e;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep one chunk's temporaries live at a time
        }
        z = wave_sum(z);
        __syncthreads();
        if ((tid & 63) == 0) buf[tid >> 6] = z;
        __syncthreads();
        z = buf[0] + buf[1] + buf[2] + buf[3] + buf[4] + buf[5] + buf[6] + buf[7];
        const float invz = 1.0f / z;
        const float py = s_ey * invz;
        const float pmin = __builtin_amdgcn_exp2f(mn * LOG2E - mb) * invz, pmax = invz;   // exp2(0) / z
        const bool clipped = (variant == B4C_CE_TF) && (pmin < KERAS_EPS || pmax > 1.0f - KERAS_EPS);
        float invS = 1.f, G = 0.f, inv_pyc = 1.0f / py, loss = -logf(py);
        if (clipped) {   // row-uniform
            float S = 0.f, Pu = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int base = (i * 512 + tid) * 8;
                if (base < V) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (base + 8 <= V || base + k < V) {
                            const float p = __builtin_amdgcn_exp2f(CE_ELEM(i, k) * LOG2E - mb) * invz;
                            const float pc = __builtin_amdgcn_fmed3f(p, KERAS_EPS, 1.0f - KERAS_EPS);
                            S += pc;
                            Pu += (pc == p) ? p : 0.f;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            S = wave_sum(S);
            Pu = wave_sum(Pu);
            __syncthreads();
            if ((tid & 63) == 0) { buf[tid >> 6] = S; buf2[tid >> 6] = Pu; }
            __syncthreads();
            S = buf[0] + buf[1] + buf[2] + buf[3] + buf[4] + buf[5] + buf[6] + buf[7];
            Pu = buf2[0] + buf2[1] + buf2[2] + buf2[3] + buf2[4] + buf2[5] + buf2[6] + buf2[7];
            const float pyc = __builtin_amdgcn_fmed3f(py, KERAS_EPS, 1.0f - KERAS_EPS);
            const float uy = (pyc == py) ? 1.f : 0.f;
            invS = 1.0f / S;
            G = Pu * invS - uy * py / pyc;
            inv_pyc = uy / pyc;
            loss = logf(S) - logf(pyc);
        }
        // d loss / d x_j = p_j (u_j / S - G) - [j == y] p_y u_y / clip(p_y)   (u = 1, S = 1, G = 0 when unclipped)
        const float ydelta = py * inv_pyc;    // == 1 when unclipped
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 512 + tid;
            const int base = c * 8;
            if (c < nch_ld) {
                float g[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float v = 0.f;
                    if (base + 8 <= V || base + k < V) {
                        const float p = __builtin_amdgcn_exp2f(CE_ELEM(i, k) * LOG2E - mb) * invz;
                        float u = 1.f;
                        if (clipped) u = (p >= KERAS_EPS && p <= 1.0f - KERAS_EPS) ? 1.f : 0.f;
                        v = p * (u * invS - G);
                        if (base + k == y) v -= ydelta;
                        v *= gs;
                    }
                    g[k] = v;
                }
                Vec8<bf16_t>::store(xr + base, g);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef CE_ELEM
        if (tid == 0) item_loss[row] = loss;
        __syncthreads();   // buf / s_ey are reused by the next row
    }
}

// Rows wider than the register-resident limit (V > 65,536: config 4's 100,000 items on the fp32 / materialised route):
// the same arithmetic streamed -- max, sum of exponentials, (TF variant) S / Pu, then the gradient in place.  The row
// (<= 1 MB) is re-read from L2 / Infinity Cache by the workgroup that just read it.
template <typename T>
__global__ void __launch_bounds__(1024) softmax_ce_stream_kernel(T *__restrict__ x, int ld, const int32_t *__restrict__ labels,
                                                                 float *__restrict__ item_loss, const float *__restrict__ grad_scale,
                                                                 int64_t R, int V, int variant) {
    __shared__ float buf[16];
    __shared__ float s_ey;
    const int tid = threadIdx.x;
    const int nch_ld = ld >> 3, nch = (V + 7) >> 3;
    const float gs = grad_scale[0];
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        T *xr = x + row * ld;
        const int y = labels[row];
        const bool valid = y >= 0 && y < V;
        if (!valid) {
            for (int c = tid; c < nch_ld; c += 1024) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
                Vec8<T>::store(xr + c * 8, v);
            }
            if (tid == 0) item_loss[row] = (y >= V) ? NAN : 0.f;
            continue;
        }
        float m = -INFINITY;
        for (int c = tid; c < nch; c += 1024) {
            float v[8];
            Vec8<T>::load(xr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c * 8 + k < V) m = fmaxf(m, v[k]);
        }
        m = block_max<16>(m, buf);
        float z = 0.f;
        for (int c = tid; c < nch; c += 1024) {
            float v[8];
            Vec8<T>::load(xr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c * 8 + k < V) {
                    const float e = expf(v[k] - m);
                    z += e;
                    if (c * 8 + k == y) s_ey = e;
                }
        }
        z = block_sum<16>(z, buf);
        const float invz = 1.0f / z;
        const float py = s_ey * invz;
        float loss = -logf(py), invS = 1.f, G = 0.f, inv_pyc = 1.0f / py;
        if (variant == B4C_CE_TF) {
            float S = 0.f, Pu = 0.f;
            for (int c = tid; c < nch; c += 1024) {
                float v[8];
                Vec8<T>::load(xr + c * 8, v);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c * 8 + k < V) {
                        const float p = expf(v[k] - m) * invz;
                        S += fminf(fmaxf(p, KERAS_EPS), 1.0f - KERAS_EPS);
                        if (p >= KERAS_EPS && p <= 1.0f - KERAS_EPS) Pu += p;
                    }
            }
            S = block_sum<16>(S, buf);
            Pu = block_sum<16>(Pu, buf);
            const float pyc = fminf(fmaxf(py, KERAS_EPS), 1.0f - KERAS_EPS);
            const float uy = (py >= KERAS_EPS && py <= 1.0f - KERAS_EPS) ? 1.f : 0.f;
            loss = logf(S) - logf(pyc);
            invS = 1.0f / S;
            G = Pu * invS - uy * py / pyc;
            inv_pyc = uy / pyc;
        }
        for (int c = tid; c < nch_ld; c += 1024) {
            float v[8];
            if (c < nch) Vec8<T>::load(xr + c * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = c * 8 + k;
                float g = 0.f;
                if (j < V) {
                    const float p = expf(v[k] - m) * invz;
                    if (variant == B4C_CE_TF) {
                        const float u = (p >= KERAS_EPS && p <= 1.0f - KERAS_EPS) ? 1.f : 0.f;
                        g = p * (u * invS - G) - (j == y ? py * inv_pyc : 0.f);
                    } else {
                        g = p - (j == y ? 1.f : 0.f);
                    }
                }
                v[k] = g * gs;
            }
            Vec8<T>::store(xr + c * 8, v);
        }
        if (tid == 0) item_loss[row] = loss;
        __syncthreads();
    }
}

extern "C" int b4c_softmax_ce_fwd_bwd(void *logits, int ld, const int32_t *labels, float *item_loss,
                                      const float *grad_scale, int64_t R, int V, int variant, int dtype, void *stream) {
    B4C_REQUIRE(logits && labels && item_loss && grad_scale && R >= 0 && V > 0, "softmax_ce_fwd_bwd: bad argument");
    B4C_REQUIRE(ld % 8 == 0 && ld >= V, "softmax_ce_fwd_bwd: pitch");
    B4C_REQUIRE(variant == B4C_CE_TF || variant == B4C_CE_PLAIN, "softmax_ce_fwd_bwd: variant %d", variant);
    if (R == 0) return B4C_OK;
    if (ld > 8 * 8 * 1024) {      // beyond the register-resident limit: streamed kernel
        const int gs_ = (int)(R < 2048 ? R : 2048);
        hipStream_t st_ = (hipStream_t)stream;
        if (dtype == B4C_F32) softmax_ce_stream_kernel<float><<<gs_, 1024, 0, st_>>>((float *)logits, ld, labels, item_loss, grad_scale, R, V, variant);
        else if (dtype == B4C_BF16) softmax_ce_stream_kernel<bf16_t><<<gs_, 1024, 0, st_>>>((bf16_t *)logits, ld, labels, item_loss, grad_scale, R, V, variant);
        else B4C_REQUIRE(false, "softmax_ce_fwd_bwd: dtype %d", dtype);
        return b4c_check_launch("softmax_ce_stream");
    }
    const int nch = (int)ceil_div64(ld / 8, 1024);
    const int grid = (int)(R < 2048 ? R : 2048);
    hipStream_t st = (hipStream_t)stream;
#define CE_LAUNCH(T, N) softmax_ce_fused_kernel<T, N><<<grid, 1024, 0, st>>>((T *)logits, ld, labels, item_loss, grad_scale, R, V, variant)
#define CE_DISPATCH(T)                        \
    if (nch <= 1) CE_LAUNCH(T, 1);            \
    else if (nch <= 2) CE_LAUNCH(T, 2);       \
    else if (nch <= 4) CE_LAUNCH(T, 4);       \
    else CE_LAUNCH(T, 8);
    if (dtype == B4C_F32) { CE_DISPATCH(float) }
    else if (dtype == B4C_BF16) {
        const int nch512 = (int)ceil_div64(ld / 8, 512);
        const int grid512 = (int)(R < 3072 ? R : 3072);
#define CE16_LAUNCH(N) softmax_ce_bf16_kernel<N><<<grid512, 512, 0, st>>>((bf16_t *)logits, ld, labels, item_loss, grad_scale, R, V, variant)
        if (nch512 <= 2) CE16_LAUNCH(2);
        else if (nch512 <= 4) CE16_LAUNCH(4);
        else if (nch512 <= 8) CE16_LAUNCH(8);
        else if (nch512 <= 13) CE16_LAUNCH(13);
        else CE16_LAUNCH(16);
#undef CE16_LAUNCH
    }
    else B4C_REQUIRE(false, "softmax_ce_fwd_bwd: dtype %d", dtype);
    return b4c_check_launch("softmax_ce_fwd_bwd");
}

// ------------------------------------------------------------------------------------------
// top-k: per-thread sorted lists -> LDS -> k rounds of block arg-max.  Order: larger value first,
// equal values -> lower index first (tf.math.top_k).  NaNs never enter a list.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

template <typename T, int KM>
__global__ void __launch_bounds__(256) topk_rows_lists_kernel(const T *__restrict__ x, int ld, int64_t R, int V, int k,
                                                        int32_t *__restrict__ topk_idx, const int32_t *__restrict__ labels,
                                                        float *__restrict__ hit, float *__restrict__ ndcg,
                                                        const int32_t *__restrict__ redo) {
    __shared__ float lv[256][KM + 1];
    __shared__ int li[256][KM + 1];
    __shared__ float wv[4];
    __shared__ int wi[4], wo[4];
    __shared__ int s_owner;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = (V + 7) >> 3;
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        if (redo && !redo[row]) continue;       // block-uniform: the threshold kernel has done this row
        const T *xr = x + row * ld;
        float tv[KM];
        int ti[KM];
#pragma unroll
        for (int q = 0; q < KM; ++q) { tv[q] = -INFINITY; ti[q] = 0x7fffffff; }
        for (int c = tid; c < nch; c += 256) {
            float v[8];
            Vec8<T>::load(xr + c * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = c * 8 + e;
                if (j < V && better(v[e], j, tv[KM - 1], ti[KM - 1])) {
                    tv[KM - 1] = v[e];
                    ti[KM - 1] = j;
#pragma unroll
                    for (int q = KM - 1; q > 0; --q) {
                        if (better(tv[q], ti[q], tv[q - 1], ti[q - 1])) {
                            const float fv = tv[q]; tv[q] = tv[q - 1]; tv[q - 1] = fv;
                            const int fi = ti[q]; ti[q] = ti[q - 1]; ti[q - 1] = fi;
                        }
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KM; ++q) { lv[tid][q] = tv[q]; li[tid][q] = ti[q]; }
        lv[tid][KM] = -INFINITY; li[tid][KM] = 0x7fffffff;
        int hp = 0;
        const int lab = labels ? labels[row] : -1;
        float h_acc = 0.f, n_acc = 0.f;
        for (int kk = 0; kk < k; ++kk) {
            float bv = lv[tid][hp];
            int bi = li[tid][hp], bo = tid;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o);
                const int oi = __shfl_xor(bi, o), oo = __shfl_xor(bo, o);
                if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; bo = oo; }
            }
            __syncthreads();
            if (lane == 0) { wv[wave] = bv; wi[wave] = bi; wo[wave] = bo; }
            __syncthreads();
            if (tid == 0) {
                float fv = wv[0]; int fi = wi[0], fo = wo[0];
                for (int w = 1; w < 4; ++w)
                    if (better(wv[w], wi[w], fv, fi)) { fv = wv[w]; fi = wi[w]; fo = wo[w]; }
                s_owner = fo;
                topk_idx[row * k + kk] = fi == 0x7fffffff ? -1 : fi;
                if (labels && fi == lab) {
                    h_acc += 1.f;
                    n_acc += 1.0f / (logf((float)(kk + 2)) / logf(2.0f));
                }
            }
            __syncthreads();
            if (tid == s_owner && hp < KM) ++hp;
        }
        if (tid == 0) {
            if (hit) hit[row] = h_acc;
            if (ndcg) ndcg[row] = n_acc;
        }
        __syncthreads();
    }
}

// Fast path: threshold selection.  Pass 1 finds every thread's best element; k rounds of block arg-max over those 256
// values give T = the k-th best of them, a lower bound of the row's k-th best (at least k elements are >= T).  Pass 2:
// the threads whose own maximum reaches T (about k of them) read their chunks again and put the elements that are not worse
// than T into LDS -- about k for continuous data -- and wave 0 picks the k best in order.  The row is read once; the
// per-thread sorted lists of the kernel above cost ~5,000 instructions per thread and row at k = 10 (562 GB/s).
// Rows with more than TOPK_CAP candidates (massive ties) are flagged in `redo` and handled by the list kernel.
#define TOPK_CAP 1024
#define TOPK_THREADS 512
#define TOPK_SLOTS 64          // threads per row that may hold candidates in the register-resident form
// lane-strided arg-max over n (value, index) pairs in LDS by ONE wave; taken entries carry the sentinel index
__device__ __forceinline__ void wave_argmax_lds(const float *v, const int *ix, int n, int lane, float &bv, int &bi, int &bp) {
    bv = -INFINITY; bi = 0x7fffffff; bp = -1;
    for (int q = lane; q < n; q += 64) {
        const float a = v[q];
        const int j = ix[q];
        if (j != 0x7fffffff && better(a, j, bv, bi)) { bv = a; bi = j; bp = q; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o), op = __shfl_xor(bp, o);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; bp = op; }
    }
}

// rank of lane's (v, j) among the first n lanes' pairs = number of pairs that are better (one wave, no LDS)
__device__ __forceinline__ int wave_rank(float v, int j, int n) {
    int rank = 0;
    for (int q = 0; q < n; ++q) {
        const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), q));
        const int oj = __builtin_amdgcn_readlane(j, q);
        rank += better(ov, oj, v, j) ? 1 : 0;
    }
    return rank;
}

// the k best of the cnt candidates in LDS, in order (wave 0), with the HitRate / NDCG terms of the row
__device__ __forceinline__ void topk_select(float *cv, int *ci, int cnt, int k, int64_t row, int tid, int lane, int wave,
                                            int32_t *__restrict__ topk_idx, const int32_t *__restrict__ labels,
                                            float *__restrict__ hit, float *__restrict__ ndcg, int32_t *__restrict__ redo) {
    if (cnt > TOPK_CAP) {
        if (tid == 0) redo[row] = 1;
    } else if (wave == 0) {
        if (lane == 0) redo[row] = 0;
        const int lab = labels ? labels[row] : -1;
        if (cnt <= 64) {        // the usual case: one candidate per lane, ordered by rank counting
            const float v = lane < cnt ? cv[lane] : -INFINITY;
            const int j = lane < cnt ? ci[lane] : 0x7fffffff;
            const int rank = wave_rank(v, j, cnt);
            if (lane < cnt && rank < k) topk_idx[row * k + rank] = j;
            if (lane >= cnt && lane < k) topk_idx[row * k + lane] = -1;          // fewer than k candidates (NaN rows)
            const bool found = labels && lane < cnt && rank < k && j == lab;
            const unsigned long long fb = __ballot(found);
            if (found) {
                if (hit) hit[row] = 1.f;
                if (ndcg) ndcg[row] = 1.0f / (logf((float)(rank + 2)) / logf(2.0f));
            } else if (fb == 0ull && lane == 0) {
                if (hit) hit[row] = 0.f;
                if (ndcg) ndcg[row] = 0.f;
            }
        } else {
            float h_acc = 0.f, n_acc = 0.f;
            for (int kk = 0; kk < k; ++kk) {
                float bv; int bi, bp;
                wave_argmax_lds(cv, ci, cnt, lane, bv, bi, bp);
                if (lane == 0) {
                    if (bp >= 0) ci[bp] = 0x7fffffff;
                    topk_idx[row * k + kk] = bi == 0x7fffffff ? -1 : bi;
                    if (labels && bi == lab) {
                        h_acc += 1.f;
                        n_acc += 1.0f / (logf((float)(kk + 2)) / logf(2.0f));
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) {
                if (hit) hit[row] = h_acc;
                if (ndcg) ndcg[row] = n_acc;
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(TOPK_THREADS) topk_rows_kernel(const T *__restrict__ x, int ld, int64_t R, int V, int k,
                                                                 int32_t *__restrict__ topk_idx, const int32_t *__restrict__ labels,
                                                                 float *__restrict__ hit, float *__restrict__ ndcg,
                                                                 int32_t *__restrict__ redo) {
    __shared__ float cv[TOPK_CAP];
    __shared__ int ci[TOPK_CAP];
    __shared__ float mv[TOPK_THREADS];
    __shared__ int s_cnt;
    __shared__ float s_tv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = (V + 7) >> 3;
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const T *xr = x + row * ld;
        // pass 1: per-thread maximum (values only; fmaxf drops NaNs).  Four loads in flight per thread.
        float m = -INFINITY;
        for (int c0 = tid; c0 < nch; c0 += 4 * TOPK_THREADS) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * TOPK_THREADS;
                Vec8<T>::load(xr + (c < nch ? c : c0) * 8, v[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * TOPK_THREADS;
                if (c + 1 < nch) {           // whole chunk inside the row (the last chunk may hold pad columns)
#pragma unroll
                    for (int e = 0; e < 8; ++e) m = fmaxf(m, v[u][e]);
                } else if (c < nch) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (c * 8 + e < V) m = fmaxf(m, v[u][e]);
                }
            }
        }
        mv[tid] = m;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        if (wave == 0) {
            // 64 maxima of disjoint element sets (lane l: threads l, l + 64, ...); the k-th largest of them is a lower
            // bound of the row's k-th largest value: at least k elements are >= it
            float g = mv[lane];
#pragma unroll
            for (int w = 1; w < TOPK_THREADS / 64; ++w) g = fmaxf(g, mv[lane + 64 * w]);
            const int rank = wave_rank(g, lane, 64);
            if (rank == k - 1) s_tv = g;      // ranks are a permutation of 0..63 (ties broken by lane)
        }
        __syncthreads();
        const float tv = s_tv;
        // Candidates = the elements >= tv.  A thread whose own maximum is below tv holds none, so only the few threads
        // with m >= tv (about k of the 512) read their chunks again: the row is NOT read a second time.
        if (m >= tv)
        for (int c0 = tid; c0 < nch; c0 += 4 * TOPK_THREADS) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * TOPK_THREADS;
                Vec8<T>::load(xr + (c < nch ? c : c0) * 8, v[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * TOPK_THREADS;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = c * 8 + e;
                    if (c < nch && j < V && v[u][e] >= tv) {
                        const int pos = atomicAdd(&s_cnt, 1);
                        if (pos < TOPK_CAP) { cv[pos] = v[u][e]; ci[pos] = j; }
                    }
                }
            }
        }
        __syncthreads();
        topk_select(cv, ci, s_cnt, k, row, tid, lane, wave, topk_idx, labels, hit, ndcg, redo);
        __syncthreads();
    }
}

// bf16, V <= 65536: the row stays in registers as raw bf16 pairs (NCH x 16 B per thread, all loads in flight at once), so the
// candidate pass reads nothing: per row one sweep of loads, the threshold (wave 0), a register scan by the ~k threads whose
// maximum reaches it, the selection (wave 0).  Against the re-reading form above the fixed cost per row drops from ~11 us
// to the two serial sections.
template <int NCH>
__global__ void __launch_bounds__(TOPK_THREADS) topk_rows_bf16_reg_kernel(const bf16_t *__restrict__ x, int ld, int64_t R, int V, int k,
                                                                          int32_t *__restrict__ topk_idx, const int32_t *__restrict__ labels,
                                                                          float *__restrict__ hit, float *__restrict__ ndcg,
                                                                          int32_t *__restrict__ redo) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    __shared__ float cv[TOPK_CAP];
    __shared__ int ci[TOPK_CAP];
    __shared__ float mv[TOPK_THREADS];
    __shared__ u32x4 stage[TOPK_SLOTS * NCH];
    __shared__ int stage_tid[TOPK_SLOTS];
    __shared__ int s_cnt, s_nslot;
    __shared__ float s_tv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = (V + 7) >> 3;
    for (int64_t row = blockIdx.x; row < R; row += gridDim.x) {
        const bf16_t *xr = x + row * ld;
        u32x4 raw[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * TOPK_THREADS + tid;
            // chunks past the row read as the most negative bf16 pattern (-inf): never a candidate
            raw[i] = (c < nch) ? *reinterpret_cast<const u32x4 *>(xr + c * 8) : (u32x4){0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u};
        }
#define TK_ELEM(i, e) __uint_as_float(((e) & 1) ? (raw[i][(e) >> 1] & 0xFFFF0000u) : (raw[i][(e) >> 1] << 16))
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int base = (i * TOPK_THREADS + tid) * 8;
            if (base + 8 <= V) {
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, TK_ELEM(i, e));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (base + e < V) m = fmaxf(m, TK_ELEM(i, e));
            }
        }
        mv[tid] = m;
        if (tid == 0) { s_cnt = 0; s_nslot = 0; }
        __syncthreads();
        if (wave == 0) {
            float g = mv[lane];
#pragma unroll
            for (int w = 1; w < TOPK_THREADS / 64; ++w) g = fmaxf(g, mv[lane + 64 * w]);
            const int rank = wave_rank(g, lane, 64);
            if (rank == k - 1) s_tv = g;
        }
        __syncthreads();
        const float tv = s_tv;
        // the ~k threads whose maximum reaches tv park their registers in LDS; everybody then scans those few hundred
        // elements with an ordinary loop (a fully unrolled per-element branch over the registers costs 249 VGPRs)
        if (m >= tv) {
            const int slot = atomicAdd(&s_nslot, 1);
            if (slot < TOPK_SLOTS) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) stage[slot * NCH + i] = raw[i];
                stage_tid[slot] = tid;
            }
        }
        __syncthreads();
        const int nslot = s_nslot;
        if (nslot <= TOPK_SLOTS) {
            const unsigned short *st16 = reinterpret_cast<const unsigned short *>(stage);
            for (int q = tid; q < nslot * NCH * 8; q += TOPK_THREADS) {
                const int slot = q / (NCH * 8), rr = q - slot * (NCH * 8);
                const int j = ((rr >> 3) * TOPK_THREADS + stage_tid[slot]) * 8 + (rr & 7);
                const float v = __uint_as_float((unsigned)st16[q] << 16);
                if (j < V && v >= tv) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    if (pos < TOPK_CAP) { cv[pos] = v; ci[pos] = j; }
                }
            }
        } else if (tid == 0) {
            s_cnt = TOPK_CAP + 1;        // more threads at the threshold than slots (mass ties): the list kernel redoes the row
        }
#undef TK_ELEM
        __syncthreads();
        topk_select(cv, ci, s_cnt, k, row, tid, lane, wave, topk_idx, labels, hit, ndcg, redo);
        __syncthreads();
    }
}

extern "C" int b4c_topk_rows(const void *scores, int ld, int64_t R, int V, int k, int32_t *topk_idx,
                             const int32_t *labels, float *hit, float *ndcg, int dtype, void *stream) {
    return b4c_topk_rows_ws(scores, ld, R, V, k, topk_idx, labels, hit, ndcg, nullptr, dtype, stream);
}

// `redo` (int32 [R], scratch) enables the threshold kernel; rows it could not finish (more than TOPK_CAP candidates:
// massive ties) are redone by the per-thread-list kernel.  redo == NULL: list kernel for every row.
extern "C" int b4c_topk_rows_ws(const void *scores, int ld, int64_t R, int V, int k, int32_t *topk_idx,
                                const int32_t *labels, float *hit, float *ndcg, int32_t *redo, int dtype, void *stream) {
    B4C_REQUIRE(scores && topk_idx && R >= 0 && V > 0, "topk_rows: bad argument");
    B4C_REQUIRE(k >= 1 && k <= B4C_MAX_TOPK && k <= V, "topk_rows: k=%d must be in [1, min(%d, V)]", k, B4C_MAX_TOPK);
    B4C_REQUIRE(ld % 8 == 0 && ld >= V, "topk_rows: pitch");
    if (R == 0) return B4C_OK;
    const int grid = (int)(R < 4096 ? R : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (redo) {
        const int g2 = (int)(R < 2048 ? R : 2048);
        if (dtype == B4C_F32) topk_rows_kernel<float><<<g2, TOPK_THREADS, 0, st>>>((const float *)scores, ld, R, V, k, topk_idx, labels, hit, ndcg, redo);
        else if (dtype == B4C_BF16) {
            const int per_thread = ((V + 7) / 8 + TOPK_THREADS - 1) / TOPK_THREADS;      // 16-B chunks per thread
            const bool al = (((uintptr_t)scores) & 15) == 0;
#define TOPK_REG(N) topk_rows_bf16_reg_kernel<N><<<g2, TOPK_THREADS, 0, st>>>((const bf16_t *)scores, ld, R, V, k, topk_idx, labels, hit, ndcg, redo)
            if (al && per_thread <= 4) TOPK_REG(4);
            else if (al && per_thread <= 8) TOPK_REG(8);
            else if (al && per_thread <= 13) TOPK_REG(13);
            else if (al && per_thread <= 16) TOPK_REG(16);
            else topk_rows_kernel<bf16_t><<<g2, TOPK_THREADS, 0, st>>>((const bf16_t *)scores, ld, R, V, k, topk_idx, labels, hit, ndcg, redo);
#undef TOPK_REG
        }
        else B4C_REQUIRE(false, "topk_rows: dtype %d", dtype);
    }
#define TOPK_LAUNCH(T, KM) topk_rows_lists_kernel<T, KM><<<grid, 256, 0, st>>>((const T *)scores, ld, R, V, k, topk_idx, labels, hit, ndcg, redo)
#define TOPK_DISPATCH(T)                  \
    if (k <= 1) TOPK_LAUNCH(T, 1);        \
    else if (k <= 4) TOPK_LAUNCH(T, 4);   \
    else if (k <= 8) TOPK_LAUNCH(T, 8);   \
    else TOPK_LAUNCH(T, 16);
    if (dtype == B4C_F32) { TOPK_DISPATCH(float) }
    else if (dtype == B4C_BF16) { TOPK_DISPATCH(bf16_t) }
    else B4C_REQUIRE(false, "topk_rows: dtype %d", dtype);
    return b4c_check_launch("topk_rows");
}
