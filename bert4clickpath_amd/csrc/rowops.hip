// HBM-bound row kernels: embedding stage, residual+dropout+LayerNorm, [MASK]-row index
// generation / gather / scatter, weight packing, Adam.  gfx950, wave64, 16-byte accesses.
#include <stdarg.h>
#include <string.h>

#include "common.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void b4c_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int b4c_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        b4c_set_error("%s: %s", what, hipGetErrorString(e));
        return B4C_ELAUNCH;
    }
    return B4C_OK;
}
extern "C" const char *b4c_last_error(void) { return g_err; }
extern "C" int b4c_abi_version(void) { return 12; }
extern "C" int b4c_keep(uint64_t seed, uint64_t e, float rate) { return b4c_keep_elem(seed, e, rate) ? 1 : 0; }

// ------------------------------------------------------------------------------------------
// embedding stage
// ------------------------------------------------------------------------------------------
struct EmbedArgs {
    const int64_t *ids[B4C_MAX_FEATURES];
    float *table[B4C_MAX_FEATURES];   // const for fwd; gradient tables for bwd
    int64_t rows[B4C_MAX_FEATURES];
    int col0[B4C_MAX_FEATURES + 1];   // column offset of each feature in the d_model-wide row (all 0 when the features are summed)
    int fd[B4C_MAX_FEATURES];         // width of each feature's table
    int n;
    int sum;                          // 1: every feature is d_model wide and the rows are ADDED (config 4 of the north star)
};

// one thread = 8 consecutive output columns of one token (a 16-B bf16 / 32-B fp32 store)
template <typename T>
__global__ void __launch_bounds__(256) embed_fwd_kernel(EmbedArgs a, const float *__restrict__ pe, float scale,
                                                        T *__restrict__ out, int ld_out, uint8_t *__restrict__ key_pad,
                                                        int64_t T_tok, int S, int d, float rate, uint64_t seed,
                                                        const int32_t *__restrict__ token_src) {
    const int cpr = d >> 3;  // 8-column chunks per row
    const int64_t total = T_tok * cpr;
    const float inv_keep = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / cpr;
        const int c = (int)(i - t * cpr) << 3;
        // packed layout: output row t is the dense position token_src[t] = b * S + s (its ids and its positional row)
        const int64_t ts = token_src ? (int64_t)token_src[t] : t;
        const int s = (int)(ts % S);
        if (c == 0 && key_pad) key_pad[t] = (a.ids[0][ts] == 0) ? 1 : 0;
        float v[8], p[8];
        if (a.sum) {        // rows of all features added in feature order (fp32), then scaled
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
            for (int f = 0; f < a.n; ++f) {
                int64_t id = a.ids[f][ts];
                id = id < 0 ? 0 : (id >= a.rows[f] ? a.rows[f] - 1 : id);
                float w[8];
                Vec8<float>::load(a.table[f] + id * d + c, w);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += w[k];
            }
        } else {
            int f = 0;
#pragma unroll
            for (int k = 1; k < B4C_MAX_FEATURES; ++k)
                if (k < a.n && c >= a.col0[k]) f = k;
            int64_t id = a.ids[f][ts];
            id = id < 0 ? 0 : (id >= a.rows[f] ? a.rows[f] - 1 : id);
            // every feature dim is a multiple of 8 (the host checks): a chunk never straddles two features
            Vec8<float>::load(a.table[f] + id * a.fd[f] + (c - a.col0[f]), v);
        }
        Vec8<float>::load(pe + (int64_t)s * d + c, p);
        const uint32_t km = rate > 0.f ? b4c_keep8(seed, (uint64_t)(t * d + c), b4c_keep_threshold(rate)) : 0xFFu;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float x = v[k] * scale + p[k];
            if (rate > 0.f) x = ((km >> k) & 1u) ? x * inv_keep : 0.f;
            v[k] = x;
        }
        Vec8<T>::template store_sel<B4C_NT(B4C_NT_EMBED)>(out + t * ld_out + c, v);
    }
}

// scatter-add into the fp32 gradient tables; one lane = one column so that a wave-instruction adds
// contiguous bytes (MI355X float-atomic rate shape).  Exact zeros (pad rows) are skipped.  Rows with
// id < EMB_HOT -- the reserved tokens ([MASK] alone receives one add per masked item, [CLS]/[SEP] one per
// sequence) and the first vocabulary entries -- are accumulated in an LDS copy per workgroup and flushed
// once: a single contended HBM row otherwise runs the whole kernel at the 14x-slower same-row atomic rate.
#define EMB_HOT 64
template <typename T>
__global__ void __launch_bounds__(256) embed_bwd_kernel(EmbedArgs a, float scale, const T *__restrict__ dout, int ld,
                                                        int64_t T_tok, int d, float rate, uint64_t seed, int64_t tok_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float hot[];   // [EMB_HOT][d]
    for (int i = threadIdx.x; i < EMB_HOT * d; i += 256) hot[i] = 0.f;
    __syncthreads();
    const float mul = scale * (rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f);
    const int64_t t0 = blockIdx.x * tok_per_wg;
    const int64_t t1 = (t0 + tok_per_wg < T_tok) ? t0 + tok_per_wg : T_tok;
    for (int64_t i = t0 * d + threadIdx.x; i < t1 * d; i += 256) {
        const int64_t t = i / d;
        const int c = (int)(i - t * d);
        float g = (float)dout[t * ld + c];
        if (g == 0.f) continue;
        if (rate > 0.f && !b4c_keep_elem(seed, (uint64_t)i, rate)) continue;
        int f0 = 0;
#pragma unroll
        for (int k = 1; k < B4C_MAX_FEATURES; ++k)
            if (k < a.n && c >= a.col0[k]) f0 = k;
        const int f1 = a.sum ? a.n : f0 + 1;      // summed features all receive the gradient of the one d_model-wide row
        for (int f = a.sum ? 0 : f0; f < f1; ++f) {
            int64_t id = a.ids[f][t];
            id = id < 0 ? 0 : (id >= a.rows[f] ? a.rows[f] - 1 : id);
            // the LDS copy of the hot rows is per column of the d_model-wide row: with summed features it serves the last one
            if (id < EMB_HOT && f == f0) atomicAdd(&hot[id * d + c], g * mul);
            else atomicAdd(a.table[f] + id * a.fd[f] + (c - a.col0[f]), g * mul);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < EMB_HOT * d; i += 256) {
        const float v = hot[i];
        if (v == 0.f) continue;
        const int id = i / d, c = i - id * d;
        int f = 0;
#pragma unroll
        for (int k = 1; k < B4C_MAX_FEATURES; ++k)
            if (k < a.n && c >= a.col0[k]) f = k;
        if (id < a.rows[f]) atomicAdd(a.table[f] + (int64_t)id * a.fd[f] + (c - a.col0[f]), v);
    }
}

static int fill_embed_args(EmbedArgs &a, int n_feat, const int64_t *const *h_ids, float *const *h_tables,
                           const int *h_dims, const int64_t *h_rows, int d_model) {
    B4C_REQUIRE(n_feat >= 1 && n_feat <= B4C_MAX_FEATURES, "embed: n_feat %d out of range", n_feat);
    memset(&a, 0, sizeof(a));
    a.n = n_feat;
    // two or more features that are each d_model wide: their rows are added, not concatenated
    a.sum = n_feat >= 2;
    for (int f = 0; f < n_feat; ++f) a.sum = a.sum && h_dims[f] == d_model;
    int off = 0;
    for (int f = 0; f < n_feat; ++f) {
        B4C_REQUIRE(h_dims[f] > 0 && h_dims[f] % 8 == 0, "embed: feature dim %d must be a positive multiple of 8", h_dims[f]);
        B4C_REQUIRE(h_ids[f] && h_tables[f] && h_rows[f] > 0, "embed: null pointer / empty table for feature %d", f);
        a.ids[f] = h_ids[f];
        a.table[f] = h_tables[f];
        a.rows[f] = h_rows[f];
        a.fd[f] = h_dims[f];
        a.col0[f] = a.sum ? 0 : off;
        off += h_dims[f];
    }
    for (int f = n_feat; f <= B4C_MAX_FEATURES; ++f) a.col0[f] = a.sum ? d_model : off;
    B4C_REQUIRE(a.sum || off == d_model, "embed: sum of feature dims %d != d_model %d (and not every feature is d_model wide)", off, d_model);
    return B4C_OK;
}

static inline int grid_for(int64_t work_items, int block) {
    int64_t g = ceil_div64(work_items, block);
    const int64_t cap = 256 * 8;  // 256 CUs x 8 blocks, grid-stride the rest
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" int b4c_embed_concat_pe_fwd(int n_feat, const int64_t *const *h_ids, const float *const *h_tables,
                                       const int *h_dims, const int64_t *h_rows, const float *pe, float scale,
                                       void *out, int ld_out, uint8_t *key_pad, int B, int S, int d_model,
                                       float dropout_rate, uint64_t seed, int dtype, void *stream) {
    return b4c_embed_concat_pe_fwd_packed(n_feat, h_ids, h_tables, h_dims, h_rows, pe, scale, out, ld_out, key_pad, B, S, d_model,
                                          dropout_rate, seed, nullptr, (int64_t)B * S, dtype, stream);
}

extern "C" int b4c_embed_concat_pe_fwd_packed(int n_feat, const int64_t *const *h_ids, const float *const *h_tables,
                                              const int *h_dims, const int64_t *h_rows, const float *pe, float scale,
                                              void *out, int ld_out, uint8_t *key_pad, int B, int S, int d_model,
                                              float dropout_rate, uint64_t seed, const int32_t *token_src, int64_t n_tokens,
                                              int dtype, void *stream) {
    EmbedArgs a;
    int rc = fill_embed_args(a, n_feat, h_ids, (float *const *)h_tables, h_dims, h_rows, d_model);
    if (rc) return rc;
    B4C_REQUIRE(pe && out && B > 0 && S > 0 && ld_out >= d_model && ld_out % 8 == 0, "embed_fwd: bad shape");
    B4C_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f, "embed_fwd: dropout_rate %f", dropout_rate);
    B4C_REQUIRE(n_tokens >= 0 && n_tokens <= (int64_t)B * S, "embed_fwd: n_tokens %lld out of range", (long long)n_tokens);
    const int64_t T_tok = token_src ? n_tokens : (int64_t)B * S;
    if (T_tok == 0) return B4C_OK;
    const int grid = grid_for(T_tok * (d_model / 8), 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_F32)
        embed_fwd_kernel<float><<<grid, 256, 0, st>>>(a, pe, scale, (float *)out, ld_out, key_pad, T_tok, S, d_model, dropout_rate, seed, token_src);
    else if (dtype == B4C_BF16)
        embed_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>(a, pe, scale, (bf16_t *)out, ld_out, key_pad, T_tok, S, d_model, dropout_rate, seed, token_src);
    else
        B4C_REQUIRE(false, "embed_fwd: dtype %d", dtype);
    return b4c_check_launch("embed_fwd");
}

extern "C" int b4c_embed_concat_pe_bwd(int n_feat, const int64_t *const *h_ids, float *const *h_dtables,
                                       const int *h_dims, const int64_t *h_rows, float scale, const void *dout,
                                       int ld_dout, int B, int S, int d_model, float dropout_rate, uint64_t seed,
                                       int dtype, void *stream) {
    EmbedArgs a;
    int rc = fill_embed_args(a, n_feat, h_ids, h_dtables, h_dims, h_rows, d_model);
    if (rc) return rc;
    B4C_REQUIRE(dout && B > 0 && S > 0 && ld_dout >= d_model, "embed_bwd: bad shape");
    const int64_t T_tok = (int64_t)B * S;
    B4C_REQUIRE(d_model <= 512, "embed_bwd: d_model %d > 512 (LDS hot-row cache)", d_model);
    int grid = (int)(T_tok < 1024 ? T_tok : 1024);
    const int64_t tok_per_wg = ceil_div64(T_tok, grid);
    grid = (int)ceil_div64(T_tok, tok_per_wg);
    const size_t shm = (size_t)EMB_HOT * d_model * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (shm > 64 * 1024) {
        (void)hipFuncSetAttribute((const void *)embed_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        (void)hipFuncSetAttribute((const void *)embed_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    }
    if (dtype == B4C_F32)
        embed_bwd_kernel<float><<<grid, 256, shm, st>>>(a, scale, (const float *)dout, ld_dout, T_tok, d_model, dropout_rate, seed, tok_per_wg);
    else if (dtype == B4C_BF16)
        embed_bwd_kernel<bf16_t><<<grid, 256, shm, st>>>(a, scale, (const bf16_t *)dout, ld_dout, T_tok, d_model, dropout_rate, seed, tok_per_wg);
    else
        B4C_REQUIRE(false, "embed_bwd: dtype %d", dtype);
    return b4c_check_launch("embed_bwd");
}

// Sorted form of the same scatter-add: `order` lists the token indices of one feature sorted by id, so a wave
// that walks 64 consecutive entries meets every id as one run, sums the run in registers and writes each
// gradient row once.  Only a run that may continue in a neighbouring wave's range (same id just before / after
// the range) is added with atomics; everything else is a plain read-modify-write.  ~105 M float atomics
// (the unsorted kernel runs at the atomic rate, 0.65 ms at C2) become ~2 per distinct id.
struct EmbedOrder {
    const int32_t *order[B4C_MAX_FEATURES];
    float *part[B4C_MAX_FEATURES];       // deterministic form: [waves][2][fd] partial sums of the runs that cross a wave boundary
};
// DET: a run that continues in a neighbouring wave's range is NOT added with float atomics (whose order, and so the sum's last
// bit, changes from run to run): the wave stores its share -- slot 0: the run comes in from the wave before, slot 1: the run
// goes on into the wave after -- and embed_bwd_runs_kernel adds the shares of one run in wave order.
template <typename T, bool DET>
__global__ void __launch_bounds__(256) embed_bwd_sorted_kernel(EmbedArgs a, EmbedOrder ord, float scale, const T *__restrict__ dout,
                                                               int ld, int64_t T_tok, int d, float rate, uint64_t seed) {
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (p0 >= T_tok) return;
    const int np = (int)((T_tok - p0 < 64) ? T_tok - p0 : 64);
    const int c0 = a.col0[f], fd = a.fd[f];
    const int32_t *order = ord.order[f];
    const int64_t *ids = a.ids[f];
    const int64_t nrows = a.rows[f];
    auto id_at = [&](int64_t p) -> int64_t {
        int64_t id = ids[order[p]];
        return id < 0 ? 0 : (id >= nrows ? nrows - 1 : id);
    };
    const int my_tok = lane < np ? order[p0 + lane] : 0;
    int64_t my_id = -1;
    if (lane < np) {
        my_id = ids[my_tok];
        my_id = my_id < 0 ? 0 : (my_id >= nrows ? nrows - 1 : my_id);
    }
    // the ids just outside the range (wave-uniform)
    const int64_t prev_id = p0 > 0 ? id_at(p0 - 1) : -2, next_id = p0 + np < T_tok ? id_at(p0 + np) : -3;
    const float mul = scale * (rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f);
    const uint32_t thr = b4c_keep_threshold(rate);
    float *table = a.table[f];
    for (int cb = 0; cb < fd; cb += 128) {          // 128 columns per pass: the lane owns columns cb + 2 lane, + 1
        const int col = cb + 2 * lane;
        const bool on = col < fd;
        float acc0 = 0.f, acc1 = 0.f;
        int64_t cur = __shfl(my_id, 0);
        auto flush = [&](int64_t id) {
            if (DET && (id == prev_id || id == next_id)) {
                if (on) {
                    float *dst = ord.part[f] + ((p0 >> 6) * 2 + (id == prev_id ? 0 : 1)) * (int64_t)fd + col;
                    dst[0] = acc0;
                    dst[1] = acc1;
                }
                return;
            }
            if (on && (acc0 != 0.f || acc1 != 0.f)) {
                float *row = table + id * fd + col;
                if (id == prev_id || id == next_id) {
                    atomicAdd(row, acc0);
                    atomicAdd(row + 1, acc1);
                } else {
                    row[0] += acc0;
                    row[1] += acc1;
                }
            }
        };
        constexpr int NR = 8;                        // rows in flight (16 measured slower)
        const int colc = on ? col : 0;               // loads are unconditional on clamped indices (a conditional load is
        for (int j0 = 0; j0 < np; j0 += NR) {        // waited for at its join: one row in flight instead of NR)
            float g0[NR], g1[NR];
            int tk[NR];
            int64_t idj[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int j = j0 + u;
                tk[u] = __shfl(my_tok, j & 63);      // lanes past np hold token 0: a valid row
                idj[u] = __shfl(my_id, j & 63);
                const T *src = dout + (int64_t)tk[u] * ld + c0 + colc;
                g0[u] = (float)src[0];
                g1[u] = (float)src[1];
            }
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                if (j0 + u >= np || !on) g0[u] = g1[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                if (j0 + u >= np) break;             // wave-uniform
                if (idj[u] != cur) {                 // wave-uniform
                    flush(cur);
                    acc0 = acc1 = 0.f;
                    cur = idj[u];
                }
                if (g0[u] != 0.f || g1[u] != 0.f) {
                    if (rate > 0.f) {
                        const uint64_t e = (uint64_t)tk[u] * d + c0 + col;      // e and e + 1 share a 4-element block
                        const uint64_t hsh = b4c_rand64(seed, e >> 2);
                        const int sh = 16 * (int)(e & 3);
                        if (((uint32_t)(hsh >> sh) & 0xFFFFu) < thr) g0[u] = 0.f;
                        if (((uint32_t)(hsh >> (sh + 16)) & 0xFFFFu) < thr) g1[u] = 0.f;
                    }
                    acc0 += g0[u] * mul;
                    acc1 += g1[u] * mul;
                }
            }
        }
        flush(cur);
    }
}

// One wave per wave boundary b (between the 64-entry ranges b and b + 1) and feature: if a run of one id crosses it AND starts in
// range b, this wave owns the run: tail share of range b, then the head shares of the ranges after it while the run goes on, added
// in that order into the id's gradient row (a plain read-modify-write: one owner per run, and no other run has this id).
__global__ void __launch_bounds__(256) embed_bwd_runs_kernel(EmbedArgs a, EmbedOrder ord, int64_t T_tok) {
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int64_t nw = (T_tok + 63) >> 6;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b + 1 >= nw) return;
    const int fd = a.fd[f];
    const int32_t *order = ord.order[f];
    const int64_t *ids = a.ids[f];
    const int64_t nrows = a.rows[f];
    auto id_at = [&](int64_t p) -> int64_t {
        int64_t id = ids[order[p]];
        return id < 0 ? 0 : (id >= nrows ? nrows - 1 : id);
    };
    const int64_t pb = b << 6;                              // first entry of range b
    const int64_t id = id_at(pb + 63);
    if (id_at(pb + 64) != id) return;                       // nothing crosses this boundary
    if (id_at(pb) == id && b > 0 && id_at(pb - 1) == id) return;      // the run came in from before range b: not its first range
    // the last range the run reaches into: the entries are sorted by id, so "first entry of range w is this id" is monotone in w
    // -- a binary search instead of a walk (the hottest id of a Zipf batch fills hundreds of ranges, and every step of a walk
    // is two dependent loads)
    int64_t lo = b + 1, hi = nw - 1;                        // range b + 1 starts with this id; find the largest w that does
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (id_at(mid << 6) == id) lo = mid; else hi = mid - 1;
    }
    const int64_t e = lo;
    const float *part = ord.part[f];
    float *table = a.table[f];
    for (int cb = 0; cb < fd; cb += 128) {
        const int col = cb + 2 * lane;
        if (col >= fd) continue;
        const float *t = part + (b * 2 + 1) * (int64_t)fd + col;
        float s0 = t[0], s1 = t[1];
        for (int64_t w = b + 1; w <= e; ++w) {               // (independent loads: the order of the adds is what is fixed)
            const float *h = part + (w * 2) * (int64_t)fd + col;
            s0 += h[0];
            s1 += h[1];
        }
        float *row = table + id * fd + col;
        row[0] += s0;
        row[1] += s1;
    }
}

extern "C" int64_t b4c_embed_concat_pe_bwd_sorted_workspace_bytes(int n_feat, const int *h_dims, int B, int S) {
    int64_t total = 0;
    const int64_t nw = ((int64_t)B * S + 63) / 64;
    for (int f = 0; f < n_feat; ++f) total += nw * 2 * ((h_dims[f] + 1) / 2 * 2) * 4;
    return total;
}

extern "C" int b4c_embed_concat_pe_bwd_sorted(int n_feat, const int64_t *const *h_ids, const int32_t *const *h_order,
                                              float *const *h_dtables, const int *h_dims, const int64_t *h_rows, float scale,
                                              const void *dout, int ld_dout, int B, int S, int d_model, float dropout_rate,
                                              uint64_t seed, int dtype, void *stream) {
    return b4c_embed_concat_pe_bwd_sorted_ws(n_feat, h_ids, h_order, h_dtables, h_dims, h_rows, scale, dout, ld_dout, B, S, d_model,
                                             dropout_rate, seed, nullptr, 0, dtype, stream);
}

extern "C" int b4c_embed_concat_pe_bwd_sorted_ws(int n_feat, const int64_t *const *h_ids, const int32_t *const *h_order,
                                                 float *const *h_dtables, const int *h_dims, const int64_t *h_rows, float scale,
                                                 const void *dout, int ld_dout, int B, int S, int d_model, float dropout_rate,
                                                 uint64_t seed, void *workspace, int64_t workspace_bytes, int dtype, void *stream) {
    EmbedArgs a;
    int rc = fill_embed_args(a, n_feat, h_ids, h_dtables, h_dims, h_rows, d_model);
    if (rc) return rc;
    B4C_REQUIRE(dout && h_order && B > 0 && S > 0 && ld_dout >= d_model && ld_dout % 2 == 0, "embed_bwd_sorted: bad shape");
    B4C_REQUIRE((int64_t)B * S < (1ll << 31), "embed_bwd_sorted: more than 2^31 tokens");
    EmbedOrder ord = {};
    const int64_t T_tok = (int64_t)B * S;
    const bool det = workspace != nullptr;
    B4C_REQUIRE(!det || workspace_bytes >= b4c_embed_concat_pe_bwd_sorted_workspace_bytes(n_feat, h_dims, B, S),
                "embed_bwd_sorted: workspace too small for the deterministic form");
    float *wsp = (float *)workspace;
    for (int f = 0; f < n_feat; ++f) {
        B4C_REQUIRE(h_order[f], "embed_bwd_sorted: null order for feature %d", f);
        ord.order[f] = h_order[f];
        B4C_REQUIRE(!det || h_dims[f] % 2 == 0, "embed_bwd_sorted: odd feature width %d", h_dims[f]);
        ord.part[f] = wsp;
        if (det) wsp += ((T_tok + 63) / 64) * 2 * h_dims[f];
    }
    dim3 grid((unsigned)ceil_div64(T_tok, 256), (unsigned)n_feat);
    hipStream_t st = (hipStream_t)stream;
#define EMB_SORTED_LAUNCH(TT) do { if (det) embed_bwd_sorted_kernel<TT, true><<<grid, 256, 0, st>>>(a, ord, scale, (const TT *)dout, ld_dout, T_tok, d_model, dropout_rate, seed); \
        else embed_bwd_sorted_kernel<TT, false><<<grid, 256, 0, st>>>(a, ord, scale, (const TT *)dout, ld_dout, T_tok, d_model, dropout_rate, seed); } while (0)
    if (dtype == B4C_F32)
        EMB_SORTED_LAUNCH(float);
    else if (dtype == B4C_BF16)
        EMB_SORTED_LAUNCH(bf16_t);
    else
        B4C_REQUIRE(false, "embed_bwd_sorted: dtype %d", dtype);
#undef EMB_SORTED_LAUNCH
    if (det && T_tok > 64) {
        dim3 g2((unsigned)ceil_div64((T_tok + 63) / 64, 4), (unsigned)n_feat);
        embed_bwd_runs_kernel<<<g2, 256, 0, st>>>(a, ord, T_tok);
    }
    return b4c_check_launch("embed_bwd_sorted");
}

// ------------------------------------------------------------------------------------------
// residual + dropout + LayerNorm.  A row of d elements is owned by G lanes (8 elements per lane
// per pass); G is a power of two so rows never straddle a wave.  d <= 1024.
// ------------------------------------------------------------------------------------------
#define LN_MAX_PASS 2  // d <= 64 lanes * 8 elems * 2 passes = 1024 (keeps the per-lane row slice in registers)

template <typename T, int G>
__global__ void __launch_bounds__(256) add_ln_fwd_kernel(const T *__restrict__ x, const T *__restrict__ y,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         T *__restrict__ z, T *__restrict__ out, float *__restrict__ stats,
                                                         int64_t rows, int d, float eps, float rate, uint64_t seed) {
    const int lane_in_row = threadIdx.x & (G - 1);
    const int rows_per_block = 256 / G;
    const int npass = (d + G * 8 - 1) / (G * 8);
    const float inv_keep = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
    const float inv_d = 1.0f / (float)d;
    for (int64_t row = blockIdx.x * (int64_t)rows_per_block + threadIdx.x / G; row < rows;
         row += (int64_t)gridDim.x * rows_per_block) {
        float v[LN_MAX_PASS][8];
        float sum = 0.f;
#pragma unroll
        for (int p = 0; p < LN_MAX_PASS; ++p) {
            if (p < npass) {
                const int c = (p * G + lane_in_row) * 8;
                if (c < d) {
                    float a[8], b[8];
                    Vec8<T>::load(x + row * d + c, a);
                    Vec8<T>::load(y + row * d + c, b);
                    const uint32_t km = rate > 0.f ? b4c_keep8(seed, (uint64_t)(row * d + c), b4c_keep_threshold(rate)) : 0xFFu;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float yy = b[k];
                        if (rate > 0.f) yy = ((km >> k) & 1u) ? yy * inv_keep : 0.f;
                        v[p][k] = a[k] + yy;
                        sum += v[p][k];
                    }
                    if (z) Vec8<T>::store(z + row * d + c, v[p]);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[p][k] = 0.f;
                }
            }
        }
        const float mean = group_sum<G>(sum) * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int p = 0; p < LN_MAX_PASS; ++p) {
            if (p < npass) {
                const int c = (p * G + lane_in_row) * 8;
                if (c < d) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float dlt = v[p][k] - mean;
                        sq += dlt * dlt;
                    }
                }
            }
        }
        const float var = group_sum<G>(sq) * inv_d;
        const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int p = 0; p < LN_MAX_PASS; ++p) {
            if (p < npass) {
                const int c = (p * G + lane_in_row) * 8;
                if (c < d) {
                    float g[8], b[8], o[8];
                    Vec8<float>::load(gamma + c, g);
                    Vec8<float>::load(beta + c, b);
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = (v[p][k] - mean) * rstd * g[k] + b[k];
                    Vec8<T>::store(out + row * d + c, o);
                }
            }
        }
        if (stats && lane_in_row == 0) {
            stats[row * 2] = mean;
            stats[row * 2 + 1] = rstd;
        }
    }
}

// backward: dz = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dout * gamma.
// dgamma/dbeta: per-thread partials over the block's rows -> LDS -> one atomic per column per block.
// DET (deterministic form, `partial` != NULL): the threads' partials meet in a fixed order -- through LDS [row group][2][d], summed
// group by group, the block's sums stored to partial[block][2][d]; ln_bwd_reduce_kernel then adds the blocks in block order.
template <typename T, int G, int NP, bool DET>
__global__ void __launch_bounds__(256, 4) add_ln_bwd_kernel(const T *__restrict__ dout, const T *__restrict__ z,
                                                         const float *__restrict__ stats, const float *__restrict__ gamma,
                                                         T *__restrict__ dz, T *__restrict__ dy, float *__restrict__ dgamma,
                                                         float *__restrict__ dbeta, int64_t rows, int d, float rate,
                                                         uint64_t seed, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [2][d]  (DET: [256 / G][2][d])
    const int lane_in_row = threadIdx.x & (G - 1);
    const int rows_per_block = 256 / G;
    const float inv_keep = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
    const float inv_d = 1.0f / (float)d;
    if (!DET) {
        for (int i = threadIdx.x; i < 2 * d; i += 256) red[i] = 0.f;
        __syncthreads();
    }
    // NP = passes of G * 8 columns a row needs (1 for d <= 128 at G = 16): a runtime pass count kept the second pass's
    // 64 registers allocated
    float pg[NP][8], pb[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) pg[p][k] = pb[p][k] = 0.f;

    for (int64_t row = blockIdx.x * (int64_t)rows_per_block + threadIdx.x / G; row < rows;
         row += (int64_t)gridDim.x * rows_per_block) {
        const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
        float gv[NP][8], xh[NP][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            {
                const int c = (p * G + lane_in_row) * 8;
                if (c < d) {
                    float go[8], zz[8], gm[8];
                    Vec8<T>::load(dout + row * d + c, go);
                    Vec8<T>::load(z + row * d + c, zz);
                    Vec8<float>::load(gamma + c, gm);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        xh[p][k] = (zz[k] - mean) * rstd;
                        gv[p][k] = go[k] * gm[k];
                        s1 += gv[p][k];
                        s2 += gv[p][k] * xh[p][k];
                        pg[p][k] += go[k] * xh[p][k];
                        pb[p][k] += go[k];
                    }
                }
            }
        }
        s1 = group_sum<G>(s1) * inv_d;
        s2 = group_sum<G>(s2) * inv_d;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            {
                const int c = (p * G + lane_in_row) * 8;
                if (c < d) {
                    float o[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = rstd * (gv[p][k] - s1 - xh[p][k] * s2);
                    Vec8<T>::template store_sel<B4C_NT(B4C_NT_LNBWD_DZ)>(dz + row * d + c, o);
                    if (rate > 0.f && dy) {
                        const uint32_t km = b4c_keep8(seed, (uint64_t)(row * d + c), b4c_keep_threshold(rate));
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = ((km >> k) & 1u) ? o[k] * inv_keep : 0.f;
                        Vec8<T>::template store_sel<B4C_NT(B4C_NT_LNBWD_DY)>(dy + row * d + c, o);
                    }
                }
            }
        }
    }
    if (DET) {
        float *mine = red + (threadIdx.x / G) * 2 * d;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = (p * G + lane_in_row) * 8;
            if (c < d) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { mine[c + k] = pg[p][k]; mine[d + c + k] = pb[p][k]; }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * d; i += 256) {
            float sum = 0.f;
            for (int grp = 0; grp < rows_per_block; ++grp) sum += red[grp * 2 * d + i];
            partial[(int64_t)blockIdx.x * 2 * d + i] = sum;
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        {
            const int c = (p * G + lane_in_row) * 8;
            if (c < d) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    atomicAdd(&red[c + k], pg[p][k]);
                    atomicAdd(&red[d + c + k], pb[p][k]);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < d; i += 256) {
        atomicAdd(dgamma + i, red[i]);
        atomicAdd(dbeta + i, red[d + i]);
    }
}

// dgamma[i] += sum over blocks of partial[block][0][i]; dbeta likewise.  One wave per column, in a FIXED order: lane l adds the
// blocks l, l + 64, l + 128, ... one after the other, then the 64 lane sums meet in wave_sum's fixed butterfly.
__global__ void __launch_bounds__(256) ln_bwd_reduce_kernel(const float *__restrict__ partial, int nblocks, int d, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= 2 * d) return;
    float sum = 0.f;
    for (int b = lane; b < nblocks; b += 64) sum += partial[(int64_t)b * 2 * d + i];
    sum = wave_sum(sum);
    if (lane == 0) { if (i < d) dgamma[i] += sum; else dbeta[i - d] += sum; }
}

static int ln_group(int d) {
    int g = 1;
    while (g * 8 < d && g < 64) g <<= 1;
    return g;
}

#define LN_DISPATCH_G(KERNEL, T, g, ...)                                         \
    switch (g) {                                                                 \
        case 1: KERNEL<T, 1> __VA_ARGS__; break;                                 \
        case 2: KERNEL<T, 2> __VA_ARGS__; break;                                 \
        case 4: KERNEL<T, 4> __VA_ARGS__; break;                                 \
        case 8: KERNEL<T, 8> __VA_ARGS__; break;                                 \
        case 16: KERNEL<T, 16> __VA_ARGS__; break;                               \
        case 32: KERNEL<T, 32> __VA_ARGS__; break;                               \
        default: KERNEL<T, 64> __VA_ARGS__; break;                               \
    }

extern "C" int b4c_add_dropout_layernorm_fwd(const void *x, const void *y, const float *gamma, const float *beta,
                                             void *z, void *out, float *stats, int64_t rows, int d, float eps,
                                             float dropout_rate, uint64_t seed, int dtype, void *stream) {
    B4C_REQUIRE(x && y && gamma && beta && out && rows > 0, "add_ln_fwd: null pointer / empty");
    B4C_REQUIRE(d > 0 && d % 8 == 0 && d <= 64 * 8 * LN_MAX_PASS, "add_ln_fwd: d=%d must be a multiple of 8, <= 1024", d);
    B4C_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f, "add_ln_fwd: dropout_rate %f", dropout_rate);
    const int g = ln_group(d);
    const int grid = grid_for(rows * g, 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_F32) {
        LN_DISPATCH_G(add_ln_fwd_kernel, float, g, <<<grid, 256, 0, st>>>((const float *)x, (const float *)y, gamma, beta, (float *)z, (float *)out, stats, rows, d, eps, dropout_rate, seed))
    } else if (dtype == B4C_BF16) {
        LN_DISPATCH_G(add_ln_fwd_kernel, bf16_t, g, <<<grid, 256, 0, st>>>((const bf16_t *)x, (const bf16_t *)y, gamma, beta, (bf16_t *)z, (bf16_t *)out, stats, rows, d, eps, dropout_rate, seed))
    } else
        B4C_REQUIRE(false, "add_ln_fwd: dtype %d", dtype);
    return b4c_check_launch("add_ln_fwd");
}

extern "C" int64_t b4c_add_dropout_layernorm_bwd_workspace_bytes(int64_t rows, int d) {
    if (rows <= 0 || d <= 0) return 0;
    return (int64_t)1024 * 2 * d * 4;
}

extern "C" int b4c_add_dropout_layernorm_bwd(const void *dout, const void *z, const float *stats, const float *gamma,
                                             void *dz, void *dy, float *dgamma, float *dbeta, int64_t rows, int d,
                                             float dropout_rate, uint64_t seed, int dtype, void *stream) {
    return b4c_add_dropout_layernorm_bwd_ws(dout, z, stats, gamma, dz, dy, dgamma, dbeta, rows, d, dropout_rate, seed, nullptr, 0, dtype, stream);
}

extern "C" int b4c_add_dropout_layernorm_bwd_ws(const void *dout, const void *z, const float *stats, const float *gamma,
                                                void *dz, void *dy, float *dgamma, float *dbeta, int64_t rows, int d,
                                                float dropout_rate, uint64_t seed, void *workspace, int64_t workspace_bytes, int dtype,
                                                void *stream) {
    B4C_REQUIRE(dout && z && stats && gamma && dz && dgamma && dbeta && rows > 0, "add_ln_bwd: null pointer / empty");
    B4C_REQUIRE(!workspace || workspace_bytes >= b4c_add_dropout_layernorm_bwd_workspace_bytes(rows, d), "add_ln_bwd: workspace too small");
    B4C_REQUIRE(d > 0 && d % 8 == 0 && d <= 64 * 8 * LN_MAX_PASS, "add_ln_bwd: d=%d must be a multiple of 8, <= 1024", d);
    B4C_REQUIRE(dropout_rate == 0.f || dy, "add_ln_bwd: dy required when dropout_rate > 0");
    const int g = ln_group(d);
    int grid = grid_for(rows * g, 256);
    if (grid > 1024) grid = 1024;  // fewer blocks -> fewer dgamma/dbeta atomics (six per CU measured slower than four)
    const bool det = workspace != nullptr;
    float *partial = (float *)workspace;
    const size_t shm = (det ? (size_t)(256 / g) : 1) * 2 * (size_t)d * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int npass = (d + g * 8 - 1) / (g * 8);      // 1, or 2 when d > 512
#define LN_BWD_LAUNCH(TT, GG, NPP) do { if (det) add_ln_bwd_kernel<TT, GG, NPP, true><<<grid, 256, shm, st>>>((const TT *)dout, (const TT *)z, stats, gamma, (TT *)dz, (TT *)dy, dgamma, dbeta, rows, d, dropout_rate, seed, partial); \
        else add_ln_bwd_kernel<TT, GG, NPP, false><<<grid, 256, shm, st>>>((const TT *)dout, (const TT *)z, stats, gamma, (TT *)dz, (TT *)dy, dgamma, dbeta, rows, d, dropout_rate, seed, partial); } while (0)
#define LN_BWD_DISPATCH(TT)                                                                              \
    switch (g) {                                                                                         \
        case 1: LN_BWD_LAUNCH(TT, 1, 1); break;                                                          \
        case 2: LN_BWD_LAUNCH(TT, 2, 1); break;                                                          \
        case 4: LN_BWD_LAUNCH(TT, 4, 1); break;                                                          \
        case 8: LN_BWD_LAUNCH(TT, 8, 1); break;                                                          \
        case 16: LN_BWD_LAUNCH(TT, 16, 1); break;                                                        \
        case 32: LN_BWD_LAUNCH(TT, 32, 1); break;                                                        \
        default: if (npass == 1) LN_BWD_LAUNCH(TT, 64, 1); else LN_BWD_LAUNCH(TT, 64, 2); break;         \
    }
    if (dtype == B4C_F32) {
        LN_BWD_DISPATCH(float)
    } else if (dtype == B4C_BF16) {
        LN_BWD_DISPATCH(bf16_t)
    } else
        B4C_REQUIRE(false, "add_ln_bwd: dtype %d", dtype);
    if (det) ln_bwd_reduce_kernel<<<(2 * d + 3) / 4, 256, 0, st>>>(partial, grid, d, dgamma, dbeta);
    return b4c_check_launch("add_ln_bwd");
}

// ------------------------------------------------------------------------------------------
// [MASK] positions: count per row (one wave per row), single-block scan, ordered write
// ------------------------------------------------------------------------------------------
template <bool NE>
__global__ void __launch_bounds__(256) mask_count_kernel(const int64_t *__restrict__ ids, int B, int S, int64_t value,
                                                         int32_t *__restrict__ counts) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B) return;
    int c = 0;
    for (int s = lane; s < S; s += 64) c += ((ids[(int64_t)wave * S + s] == value) != NE);
    c = (int)wave_sum((float)c);  // S < 2^24 so the float sum is exact
    if (lane == 0) counts[wave] = c;
}

// clamp >= 0: offsets are clamped to it (a caller-given row count -- the packed layout's token count, the sync-free Cloze
// path's B x max_masked_per_row -- must never index past the tensors that were sized by it) and maxcount[0] is written as -(max) - 1
// when the true total differs from it (EXACT) or exceeds it (!EXACT): a poison flag the host folds into the loss without a
// read-back; `poison` (optional) is set to -1 in the same case
template <bool EXACT>
__global__ void __launch_bounds__(1024) mask_scan_kernel(const int32_t *__restrict__ counts, int B,
                                                         int32_t *__restrict__ offsets, int32_t *__restrict__ maxcount,
                                                         int32_t clamp, int32_t *__restrict__ poison) {
    __shared__ int32_t part[1024];
    __shared__ int32_t pmax[1024];
    const int tid = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int lo = tid * per, hi = min(B, lo + per);
    int32_t s = 0, m = 0;
    for (int i = lo; i < hi; ++i) { s += counts[i]; m = max(m, counts[i]); }
    part[tid] = s; pmax[tid] = m;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan / running max
        int32_t v = tid >= o ? part[tid - o] : 0;
        int32_t w = tid >= o ? pmax[tid - o] : 0;
        __syncthreads();
        part[tid] += v; pmax[tid] = max(pmax[tid], w);
        __syncthreads();
    }
    int32_t run = part[tid] - s;
    for (int i = lo; i < hi; ++i) { offsets[i] = (clamp >= 0 && run > clamp) ? clamp : run; run += counts[i]; }
    if (tid == 1023) {
        const int32_t total = part[1023];
        offsets[B] = (clamp >= 0 && total > clamp) ? clamp : total;
        const bool bad = clamp >= 0 && (EXACT ? total != clamp : total > clamp);
        if (maxcount) maxcount[0] = bad ? -pmax[1023] - 1 : pmax[1023];
        if (poison && bad) poison[0] = -1;
    }
}

// `inverse` (optional, int32 [B*S]): inverse[b*S + s] = rank of the hit in row-major order, -1 elsewhere
template <bool NE>
__global__ void __launch_bounds__(256) mask_write_kernel(const int64_t *__restrict__ ids, int B, int S, int64_t value,
                                                         const int32_t *__restrict__ offsets, int32_t *__restrict__ flat_idx,
                                                         int32_t cap, int32_t *__restrict__ inverse) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B) return;
    int32_t base = offsets[wave];
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool hit = s < S && ((ids[(int64_t)wave * S + s] == value) != NE);
        const unsigned long long bal = __ballot(hit);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && base + before < cap) flat_idx[base + before] = wave * S + s;
        if (inverse && s < S) inverse[wave * S + s] = (hit && base + before < cap) ? base + before : -1;
        base += __popcll(bal);
    }
}

extern "C" int b4c_mask_positions(const int64_t *ids, int B, int S, int64_t value, int32_t *counts, int32_t *offsets,
                                  int32_t *flat_idx, int32_t cap, int32_t *maxcount, int32_t *poison, void *stream) {
    B4C_REQUIRE(ids && counts && offsets && flat_idx && B > 0 && S > 0 && cap >= 0, "mask_positions: bad argument");
    B4C_REQUIRE((int64_t)B * S < (1ll << 31), "mask_positions: B*S must fit int32");
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)ceil_div64((int64_t)B * 64, 256);
    mask_count_kernel<false><<<grid, 256, 0, st>>>(ids, B, S, value, counts);
    // more matches than `cap` rows: the offsets stay inside the cap (consumers size their row tensors by it) and the flags say so
    mask_scan_kernel<false><<<1, 1024, 0, st>>>(counts, B, offsets, maxcount, cap, poison);
    mask_write_kernel<false><<<grid, 256, 0, st>>>(ids, B, S, value, offsets, flat_idx, cap, nullptr);
    return b4c_check_launch("mask_positions");
}

// Packed (padding-free) token layout: the positions whose id is NOT `pad_value`, row-major.
//   counts[B] = real tokens per sequence, cu_seqlens[B+1] = their exclusive scan (cu[B] = T_real), token_src[cap] = b*S + s
//   of every real token (entries >= T_real untouched), packed_of[B*S] = packed row of a dense position or -1.
extern "C" int b4c_nonpad_positions(const int64_t *ids, int B, int S, int64_t pad_value, int32_t *counts, int32_t *cu_seqlens,
                                    int32_t *token_src, int32_t cap, int32_t *packed_of, int32_t *maxcount, void *stream) {
    B4C_REQUIRE(ids && counts && cu_seqlens && token_src && B > 0 && S > 0 && cap >= 0, "nonpad_positions: bad argument");
    B4C_REQUIRE((int64_t)B * S < (1ll << 31), "nonpad_positions: B*S must fit int32");
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)ceil_div64((int64_t)B * 64, 256);
    mask_count_kernel<true><<<grid, 256, 0, st>>>(ids, B, S, pad_value, counts);
    mask_scan_kernel<true><<<1, 1024, 0, st>>>(counts, B, cu_seqlens, maxcount, cap, nullptr);
    mask_write_kernel<true><<<grid, 256, 0, st>>>(ids, B, S, pad_value, cu_seqlens, token_src, cap, packed_of);
    return b4c_check_launch("nonpad_positions");
}

// out[i] = idx[i] >= 0 ? map[idx[i]] : -1   (dense [MASK] positions -> rows of the packed encoder output)
__global__ void __launch_bounds__(256) remap_index_kernel(const int32_t *__restrict__ idx, const int32_t *__restrict__ map,
                                                          int32_t *__restrict__ out, int64_t n) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i < n) out[i] = idx[i] >= 0 ? map[idx[i]] : -1;
}
extern "C" int b4c_remap_index(const int32_t *idx, const int32_t *map, int32_t *out, int64_t n, void *stream) {
    B4C_REQUIRE(idx && map && out && n >= 0, "remap_index: bad argument");
    if (n == 0) return B4C_OK;
    remap_index_kernel<<<(int)ceil_div64(n, 256), 256, 0, (hipStream_t)stream>>>(idx, map, out, n);
    return b4c_check_launch("remap_index");
}

__global__ void __launch_bounds__(256) padded_index_kernel(const int32_t *__restrict__ counts, const int32_t *__restrict__ offsets,
                                                           const int32_t *__restrict__ flat_idx, int B, int M,
                                                           int32_t *__restrict__ padded) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (int64_t)B * M) return;
    const int b = (int)(i / M), m = (int)(i % M);
    padded[i] = m < counts[b] ? flat_idx[offsets[b] + m] : -1;
}

extern "C" int b4c_padded_index(const int32_t *counts, const int32_t *offsets, const int32_t *flat_idx, int B, int M,
                                int32_t *padded_idx, void *stream) {
    B4C_REQUIRE(counts && offsets && flat_idx && padded_idx && B > 0 && M > 0, "padded_index: bad argument");
    padded_index_kernel<<<(int)ceil_div64((int64_t)B * M, 256), 256, 0, (hipStream_t)stream>>>(counts, offsets, flat_idx, B, M, padded_idx);
    return b4c_check_launch("padded_index");
}

template <typename T, bool SCATTER>
__global__ void __launch_bounds__(256) move_rows_kernel(const T *__restrict__ src, int ld_src, const int32_t *__restrict__ idx,
                                                        T *__restrict__ dst, int ld_dst, int64_t n, int width) {
    const int cpr = width >> 3;
    const int64_t total = n * cpr;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) << 3;
        const int32_t j = idx[r];
        float v[8];
        if (SCATTER) {
            if (j < 0) continue;
            Vec8<T>::load(src + r * ld_src + c, v);
            Vec8<T>::store(dst + (int64_t)j * ld_dst + c, v);
        } else {
            if (j >= 0) Vec8<T>::load(src + (int64_t)j * ld_src + c, v);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
            }
            Vec8<T>::store(dst + r * ld_dst + c, v);
        }
    }
}

extern "C" int b4c_gather_rows(const void *in, int ld_in, const int32_t *idx, void *out, int ld_out, int64_t n_out,
                               int width, int dtype, void *stream) {
    B4C_REQUIRE(in && idx && out && n_out >= 0 && width > 0 && width % 8 == 0 && ld_in >= width && ld_out >= width &&
                    ld_in % 8 == 0 && ld_out % 8 == 0, "gather_rows: bad shape");
    if (n_out == 0) return B4C_OK;
    const int grid = grid_for(n_out * (width / 8), 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_F32) move_rows_kernel<float, false><<<grid, 256, 0, st>>>((const float *)in, ld_in, idx, (float *)out, ld_out, n_out, width);
    else if (dtype == B4C_BF16) move_rows_kernel<bf16_t, false><<<grid, 256, 0, st>>>((const bf16_t *)in, ld_in, idx, (bf16_t *)out, ld_out, n_out, width);
    else B4C_REQUIRE(false, "gather_rows: dtype %d", dtype);
    return b4c_check_launch("gather_rows");
}

extern "C" int b4c_scatter_rows(const void *src, int ld_src, const int32_t *idx, void *dst, int ld_dst, int64_t n_src,
                                int64_t n_dst, int width, int dtype, void *stream) {
    B4C_REQUIRE(src && idx && dst && n_src >= 0 && n_dst > 0 && width > 0 && width % 8 == 0 && ld_src >= width &&
                    ld_dst >= width && ld_src % 8 == 0 && ld_dst % 8 == 0, "scatter_rows: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const size_t esz = dtype == B4C_F32 ? 4 : 2;
    B4C_REQUIRE(dtype == B4C_F32 || dtype == B4C_BF16, "scatter_rows: dtype %d", dtype);
    if (hipMemsetAsync(dst, 0, (size_t)n_dst * ld_dst * esz, st) != hipSuccess) return b4c_check_launch("scatter_rows memset");
    if (n_src == 0) return B4C_OK;
    const int grid = grid_for(n_src * (width / 8), 256);
    if (dtype == B4C_F32) move_rows_kernel<float, true><<<grid, 256, 0, st>>>((const float *)src, ld_src, idx, (float *)dst, ld_dst, n_src, width);
    else move_rows_kernel<bf16_t, true><<<grid, 256, 0, st>>>((const bf16_t *)src, ld_src, idx, (bf16_t *)dst, ld_dst, n_src, width);
    return b4c_check_launch("scatter_rows");
}

// ------------------------------------------------------------------------------------------
// weight packing: fp32 Keras kernel [K][N] -> T compute copy, optionally transposed (32x32 LDS tile)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) pack_weight_kernel(const float *__restrict__ src, int K, int N, T *__restrict__ dst,
                                                          int ld_dst, int transpose) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (!transpose) {
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r, n = n0 + tx;
            if (k < K && n < N) dst[(int64_t)k * ld_dst + n] = (T)src[(int64_t)k * N + n];
        }
        return;
    }
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        tile[r][tx] = (k < K && n < N) ? src[(int64_t)k * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) dst[(int64_t)n * ld_dst + k] = (T)tile[tx][r];
    }
}

extern "C" int b4c_pack_weight(const float *src, int K, int N, void *dst, int ld_dst, int transpose, int dtype,
                               void *stream) {
    B4C_REQUIRE(src && dst && K > 0 && N > 0, "pack_weight: bad argument");
    B4C_REQUIRE(ld_dst >= (transpose ? K : N), "pack_weight: ld_dst %d too small", ld_dst);
    dim3 grid((N + 31) / 32, (K + 31) / 32);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_F32) pack_weight_kernel<float><<<grid, 256, 0, st>>>(src, K, N, (float *)dst, ld_dst, transpose);
    else if (dtype == B4C_BF16) pack_weight_kernel<bf16_t><<<grid, 256, 0, st>>>(src, K, N, (bf16_t *)dst, ld_dst, transpose);
    else B4C_REQUIRE(false, "pack_weight: dtype %d", dtype);
    return b4c_check_launch("pack_weight");
}

// every dense layer's compute copies in ONE launch: grid = (max tiles, descriptors); a tile = 64 (k) x 64 (n) of the fp32
// master: read as float4 along n, written as 8-byte bf16 quads along n (wc: same orientation) and along k (wt: through an
// LDS transpose).  (The first form moved 32 x 32 tiles with 2-byte stores: 0.11 ms per step for 112 MB.)
template <typename T>
__global__ void __launch_bounds__(256) pack_batched_kernel(const b4c_pack_desc *__restrict__ desc) {
    __shared__ float tile[64][65];
    const b4c_pack_desc d = desc[blockIdx.y];
    const int tiles_n = (d.N + 63) / 64, tiles_k = (d.K + 63) / 64;
    if ((int)blockIdx.x >= tiles_n * tiles_k) return;
    const int tid = threadIdx.x;
    const int k0 = (blockIdx.x / tiles_n) * 64, n0 = (blockIdx.x % tiles_n) * 64;
    T *wt = reinterpret_cast<T *>(d.wt), *wc = reinterpret_cast<T *>(d.wc);
    const bool vec_src = (d.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.src) & 15) == 0);
    // 64 rows (k) x 16 quads (n): thread -> (row = tid / 16 + 16 i, quad = tid % 16)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 4) + 16 * i, q4 = (tid & 15) * 4;
        const int k = k0 + r, n = n0 + q4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (k < d.K) {
            if (vec_src && n + 3 < d.N) {
                const f32x4 s4 = *reinterpret_cast<const f32x4 *>(d.src + (int64_t)k * d.N + n);
                v[0] = s4[0]; v[1] = s4[1]; v[2] = s4[2]; v[3] = s4[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < d.N) v[j] = d.src[(int64_t)k * d.N + n + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[r][q4 + j] = v[j];
        if (wc && k < d.K) {
            T *dst = wc + (int64_t)k * d.ld_c + d.col_off + n;
            if (n + 3 < d.N && ((d.ld_c | d.col_off) % 4 == 0)) {
                T w4[4] = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                if (sizeof(T) == 2) { unsigned long long u; __builtin_memcpy(&u, w4, 8); *reinterpret_cast<unsigned long long *>(dst) = u; }
                else { dst[0] = w4[0]; dst[1] = w4[1]; dst[2] = w4[2]; dst[3] = w4[3]; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < d.N) dst[j] = (T)v[j];
            }
        }
    }
    __syncthreads();
    if (wt) {
        // 64 rows (n) x 16 quads (k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 4) + 16 * i, q4 = (tid & 15) * 4;
            const int n = n0 + r, k = k0 + q4;
            if (n < d.N) {
                T *dst = wt + (int64_t)(d.col_off + n) * d.ld_t + k;
                T w4[4] = {(T)tile[q4][r], (T)tile[q4 + 1][r], (T)tile[q4 + 2][r], (T)tile[q4 + 3][r]};
                if (k + 3 < d.K && d.ld_t % 4 == 0) {
                    if (sizeof(T) == 2) { unsigned long long u; __builtin_memcpy(&u, w4, 8); *reinterpret_cast<unsigned long long *>(dst) = u; }
                    else { dst[0] = w4[0]; dst[1] = w4[1]; dst[2] = w4[2]; dst[3] = w4[3]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < d.K) dst[j] = w4[j];
                }
            }
        }
    }
    if (blockIdx.x == 0 && d.bias_src && d.bias_dst)
        for (int n = threadIdx.x; n < d.N; n += 256) d.bias_dst[d.col_off + n] = d.bias_src[n];
}

extern "C" int b4c_pack_weights_batched(const b4c_pack_desc *d_desc, int n_desc, int max_tiles, int dtype, void *stream) {
    B4C_REQUIRE(d_desc && n_desc > 0 && max_tiles > 0 && n_desc <= 65535, "pack_weights_batched: bad argument");
    dim3 grid(max_tiles, n_desc);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == B4C_F32) pack_batched_kernel<float><<<grid, 256, 0, st>>>(d_desc);
    else if (dtype == B4C_BF16) pack_batched_kernel<bf16_t><<<grid, 256, 0, st>>>(d_desc);
    else B4C_REQUIRE(false, "pack_weights_batched: dtype %d", dtype);
    return b4c_check_launch("pack_weights_batched");
}

// ------------------------------------------------------------------------------------------
// Adam over a flat fp32 arena
// ------------------------------------------------------------------------------------------
// one element, one step: the ONLY place the update is written down (adam_kernel and adam_rows_kernel must agree bit for bit)
// Every rounding is spelled out and contraction is off: left to the compiler, `m * b1 + g * (1 - b1)` became fma(m, b1, g (1 - b1))
// in one kernel and fma(g, 1 - b1, m b1) in the other (one ulp apart now and then: found by tests/test_gpu_lazy_adam.py).
__device__ __forceinline__ void adam_elem(float &pp, float gk, float &mm, float &vv, float lr_t, float b1, float b2, float eps) {
#pragma clang fp contract(off)
    const float gm = gk * (1.f - b1);
    const float gv = (gk * gk) * (1.f - b2);
    mm = __builtin_fmaf(mm, b1, gm);
    vv = __builtin_fmaf(vv, b2, gv);
    const float num = lr_t * mm;
    const float den = sqrtf(vv) + eps;
    pp = pp - num / den;
}

__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, float lr_t, float b1, float b2,
                                                   float eps, float gmul) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 pp = reinterpret_cast<f32x4 *>(p)[i], gg = reinterpret_cast<const f32x4 *>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4 *>(m)[i], vv = reinterpret_cast<f32x4 *>(v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = pp[k], mk = mm[k], vk = vv[k];
            adam_elem(pk, gg[k] * gmul, mk, vk, lr_t, b1, b2, eps);
            pp[k] = pk; mm[k] = mk; vv[k] = vk;
        }
        reinterpret_cast<f32x4 *>(p)[i] = pp;
        reinterpret_cast<f32x4 *>(m)[i] = mm;
        reinterpret_cast<f32x4 *>(v)[i] = vv;
    }
    // tail
    for (int64_t i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float pk = p[i], mk = m[i], vk = v[i];
        adam_elem(pk, g[i] * gmul, mk, vk, lr_t, b1, b2, eps);
        p[i] = pk; m[i] = mk; v[i] = vk;
    }
}

// ------------------------------------------------------------------------------------------
// Adam for row-sparse tables (config 5: two 2M-row tables of which < 0.1 % of the rows carry a gradient in a step).
// Keras' Adam is dense-equivalent: the moments of EVERY row decay every step and the row moves by its momentum
// (SURVEY 8c iii), which as one kernel is a pass over 28 B per parameter per step.  Here a row keeps the number of the last
// step it is current through (stamp[row]); a row is brought up to date only when somebody is about to read it or when it
// receives a gradient, by replaying the zero-gradient steps it missed -- the SAME fp32 operations in the same order as
// adam_kernel executes them with g = 0, with each missed step's own lr_t out of lr_hist[] -- so the table is, bit for bit,
// what the dense kernel would have made of it.  A row whose moments are all zero (never touched) replays to itself.
//   mode 0 (catch up): rows are brought to step t (all zero-gradient steps).  Before the forward pass reads them.
//   mode 1 (step):     rows are brought to step t - 1, then take step t with their gradient row; the gradient row is
//                      zeroed behind it (the table's gradient is all zeros between steps: no 3 GB fill per step).
// ids != NULL: one wave per entry of ids; a row that occurs several times is claimed once (atomicMax on its stamp).
// ids == NULL: rows [row_lo, row_lo + n): each exactly once -- the rotating catch-up that bounds every row's staleness,
//              the full catch-up in front of a checkpoint, and the dense fallback of a data-parallel step.
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256) adam_rows_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                        float *__restrict__ v, int32_t *__restrict__ stamp,
                                                        const int64_t *__restrict__ ids, int64_t n, int64_t row_lo, int64_t rows,
                                                        int width, const float *__restrict__ lr_hist, int t, float b1, float b2,
                                                        float eps, float gmul) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t i = blockIdx.x * 4ll + wave;
    if (i >= n) return;
    int64_t r;
    int old;
    if (ids) {
        r = ids[i];
        r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);          // clamped as the gather kernels clamp ids
        // cheap look first: once the row is claimed the other occurrences of a hot id leave without an atomic
        old = __hip_atomic_load(&stamp[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old < t) {
            if (lane == 0) old = atomicMax(&stamp[r], t);
            old = __builtin_amdgcn_readfirstlane(old);
        }
        if (old >= t) return;                                  // current already, or another wave has the row
    } else {
        r = row_lo + i;
        old = stamp[r];
        if (old >= t || (MODE == 0 && old == 0)) return;       // (stamp 0: never touched, all moments zero -- nothing to replay)
    }
    const int t_replay = MODE == 0 ? t : t - 1;                // zero-gradient steps old + 1 .. t_replay
    for (int c = lane * 4; c < width; c += 256) {
        const int64_t o = r * (int64_t)width + c;
        const f32x4 p4 = *reinterpret_cast<f32x4 *>(p + o), m4 = *reinterpret_cast<f32x4 *>(m + o), v4 = *reinterpret_cast<f32x4 *>(v + o);
        f32x4 gg = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 1) gg = *reinterpret_cast<const f32x4 *>(g + o);
        float pp[4] = {p4[0], p4[1], p4[2], p4[3]}, mm[4] = {m4[0], m4[1], m4[2], m4[3]}, vv[4] = {v4[0], v4[1], v4[2], v4[3]};
        // a row that never received a gradient has m = v = 0 and every zero-gradient step leaves it as it is
        const bool live = (mm[0] != 0.f) | (mm[1] != 0.f) | (mm[2] != 0.f) | (mm[3] != 0.f) | (vv[0] != 0.f) | (vv[1] != 0.f) |
                          (vv[2] != 0.f) | (vv[3] != 0.f);
        if (old > 0 && live) {
            for (int s = old + 1; s <= t_replay; ++s) {
                const float lr_s = lr_hist[s];
#pragma unroll
                for (int k = 0; k < 4; ++k) adam_elem(pp[k], 0.f, mm[k], vv[k], lr_s, b1, b2, eps);
            }
        }
        if (MODE == 1) {
            const float lr_t = lr_hist[t];
#pragma unroll
            for (int k = 0; k < 4; ++k) adam_elem(pp[k], gg[k] * gmul, mm[k], vv[k], lr_t, b1, b2, eps);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4 *>(g + o) = z;
        }
        const f32x4 po = {pp[0], pp[1], pp[2], pp[3]}, mo = {mm[0], mm[1], mm[2], mm[3]}, vo = {vv[0], vv[1], vv[2], vv[3]};
        *reinterpret_cast<f32x4 *>(p + o) = po;
        *reinterpret_cast<f32x4 *>(m + o) = mo;
        *reinterpret_cast<f32x4 *>(v + o) = vo;
    }
    if (!ids && lane == 0) stamp[r] = t;
}

extern "C" int b4c_adam_rows(float *p, float *g, float *m, float *v, int32_t *stamp, const int64_t *ids, int64_t n, int64_t row_lo,
                             int64_t rows, int width, const float *lr_hist, int t, float beta1, float beta2, float eps,
                             float grad_mul, int mode, void *stream) {
    B4C_REQUIRE(p && m && v && stamp && lr_hist && rows > 0 && width > 0 && t >= 0, "adam_rows: bad argument");
    B4C_REQUIRE(mode == 0 || (mode == 1 && g), "adam_rows: mode %d (0 = catch up, 1 = step; the step needs the gradient table)", mode);
    B4C_REQUIRE(width % 4 == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                "adam_rows: width %d must be a multiple of 4 and the tables 16-byte aligned", width);
    B4C_REQUIRE(ids || (row_lo >= 0 && row_lo + n <= rows), "adam_rows: rows [%lld, %lld) outside the table of %lld rows",
                (long long)row_lo, (long long)(row_lo + n), (long long)rows);
    if (n <= 0) return 0;
    const int64_t blocks = (n + 3) / 4;
    B4C_REQUIRE(blocks < (1ll << 31), "adam_rows: %lld rows in one call", (long long)n);
    if (mode == 0) adam_rows_kernel<0><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, stamp, ids, n, row_lo, rows, width, lr_hist, t, beta1, beta2, eps, grad_mul);
    else adam_rows_kernel<1><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, stamp, ids, n, row_lo, rows, width, lr_hist, t, beta1, beta2, eps, grad_mul);
    return b4c_check_launch("adam_rows");
}

extern "C" int b4c_adam_step(float *p, const float *g, float *m, float *v, int64_t n, float lr_t, float beta1,
                             float beta2, float eps, float grad_mul, void *stream) {
    B4C_REQUIRE(p && g && m && v && n > 0, "adam_step: bad argument");
    B4C_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: pointers must be 16-byte aligned");
    adam_kernel<<<grid_for(n / 4 + 1, 256), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr_t, beta1, beta2, eps, grad_mul);
    return b4c_check_launch("adam_step");
}
