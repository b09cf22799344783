"""Host-side operators over libb4c_hip.so: raw launch wrappers plus the block-level
``torch.autograd.Function``s the modules are made of.  PyTorch is used for device
memory, streams and the autograd tape only -- every FLOP and byte below runs in the
hand-written HIP kernels.  No CPU path exists: CPU tensors raise ``B4CError``."""
import ctypes
import os

import torch

from . import _lib as L
from ._lib import B4CError

LN_EPS = 1e-6

_weights_epoch = 0      # bumped by optimizers that write parameters through raw pointers


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1


def rup8(n):
    return (n + 7) // 8 * 8


# ---- optional launch recorder (bench.py): HIP events on the launch stream around EVERY launch of the hot-path
# kernels, with the algorithmic bytes / flops of that launch, so the dominant kernel family and its roofline
# fraction are measured live inside the timed region -------------------------------------------------------
_rec = None        # None = off; else dict family -> [ms_events..., bytes, flops, launches]
# host-side facts about the batch in flight that the recorder's ALGORITHMIC counts need and the launch sites cannot see
# (the sequence lengths live on the device): bench.py sets them per step.
#   token_rows: rows of the token-sized tensors (T): GEMMs with fewer than half of them are booked as `gemm_nt_rows`
#   sum_len_sq: sum over sequences of len^2 -- the real work of the packed attention (T x S is an upper bound)
#   sum_q_len:  sum over sequences of (masked query rows x len) -- the masked-query attention of the last layer
rec_hints = {}


def set_record_hints(**kw):
    rec_hints.clear()
    rec_hints.update(kw)



def start_recording():
    global _rec
    _rec = {}


def pause_recording():
    """Stop bracketing launches WITHOUT synchronising; hand the result to stop_recording() later."""
    global _rec
    rec, _rec = _rec, None
    return rec


def stop_recording(rec=None):
    """-> {family: {'ms': total, 'launches': n, 'bytes': algorithmic bytes, 'flops': algorithmic flops}} (synchronises)."""
    global _rec
    if rec is None:
        rec, _rec = _rec, None
    torch.cuda.synchronize()
    out = {}
    for fam, items in (rec or {}).items():
        out[fam] = {'ms': sum(a.elapsed_time(b) for a, b, _, _ in items), 'launches': len(items),
                    'bytes': sum(x[2] for x in items), 'flops': sum(x[3] for x in items)}
    return out


# B4C_FAMILY_LOG=<path> (scratch/pmc_traffic.py): EVERY launch that goes through a recorder site is noted, in host order, with
# the family the recorder books it under and its algorithmic bytes -- whether the recorder is on or not.  The PMC script pairs the
# i-th dispatch of a kernel with the i-th note of that kernel's families, so "which launch belongs to which family" has one
# source: this file's `_record(...)` call sites (token-sized and row-sized launches of one kernel are told apart here only).
family_log = [] if os.environ.get('B4C_FAMILY_LOG') else None


def dump_family_log():
    if family_log is not None:
        import json
        with open(os.environ['B4C_FAMILY_LOG'], 'w') as f:
            json.dump(family_log, f)


class _record:
    def __init__(self, family, nbytes, flops=0):
        self.on = _rec is not None
        self.family, self.nbytes, self.flops = family, nbytes, flops
        if family_log is not None:
            family_log.append((family, int(nbytes)))

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.on and _rec is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _rec.setdefault(self.family, []).append((self.a, b, self.nbytes, self.flops))


_timers = {}


def enable_timer(name):
    _timers[name] = []


def timer_results_ms(name):
    """Mean / count of the recorded launches (call after a device synchronize)."""
    ev = _timers.get(name) or []
    ms = [a.elapsed_time(b) for a, b in ev]
    return (sum(ms) / len(ms) if ms else None), len(ms)


def reset_timer(name):
    if name in _timers:
        _timers[name] = []


class _timed:
    def __init__(self, name):
        self.rec = _timers.get(name)

    def __enter__(self):
        if self.rec is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.rec is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.rec.append((self.a, b))


def dt_code(dtype):
    if dtype == torch.float32:
        return L.F32
    if dtype == torch.bfloat16:
        return L.BF16
    raise B4CError('unsupported compute dtype %s (float32 or bfloat16)' % dtype)


def _p(t):
    return None if t is None else t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise B4CError('bert4clickpath_amd runs on the HIP device only; got a CPU tensor '
                           '(there is no CPU fallback -- move inputs and the model to "cuda")')


# --------------------------------------------------------------------------------------
# raw launches
# --------------------------------------------------------------------------------------
def _feature_arrays(ids_list, tables):
    n = len(ids_list)
    ids_arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ids_list])
    tab_arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tables])
    dims = (ctypes.c_int * n)(*[int(t.shape[1]) for t in tables])
    rows = (ctypes.c_int64 * n)(*[int(t.shape[0]) for t in tables])
    return n, ids_arr, tab_arr, dims, rows


class Packed:
    """Padding-free token layout of one batch (include/b4c.h "packed token layout"): the T real tokens of the (B, S) batch in
    row-major order.  cu [B+1] int32 (sequence b = packed rows cu[b] .. cu[b+1]), tok_src [>= T] int32 (dense position
    b*S + s of packed row t), packed_of [B*S] int32 (inverse, -1 at pads), max_len (upper bound of the longest sequence)."""
    __slots__ = ('cu', 'tok_src', 'packed_of', 'B', 'S', 'T', 'max_len', 'ids_packed')

    def __init__(self, cu, tok_src, packed_of, B, S, T, max_len):
        self.cu, self.tok_src, self.packed_of, self.B, self.S, self.T, self.max_len = cu, tok_src, packed_of, B, S, T, max_len
        self.ids_packed = None


def nonpad_positions(ids, cap, pad_value=0):
    """-> (counts [B], cu [B+1], tok_src [cap], packed_of [B*S], maxcount [1]) (all int32): b4c_nonpad_positions."""
    _cuda(ids)
    B, S = ids.shape
    dev = ids.device
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    cu = torch.empty(B + 1, dtype=torch.int32, device=dev)
    tok_src = zeros(max(cap, 1), dtype=torch.int32, device=dev)      # rows past the true count (wrong cap) read token 0
    packed_of = torch.empty(B * S, dtype=torch.int32, device=dev)
    mx = torch.empty(1, dtype=torch.int32, device=dev)
    L.check(L.lib().b4c_nonpad_positions(_p(ids), B, S, pad_value, _p(counts), _p(cu), _p(tok_src), cap, _p(packed_of), _p(mx),
                                         _st()), 'nonpad_positions')
    return counts, cu, tok_src, packed_of, mx


def remap_index(idx, mapping):
    out = torch.empty_like(idx)
    L.check(L.lib().b4c_remap_index(_p(idx), _p(mapping), _p(out), idx.shape[0], _st()), 'remap_index')
    return out


def embed_concat_pe_fwd(ids_list, tables, pe, scale, rate, seed, dtype, packed=None, combine='concat'):
    """combine='sum' (two or more features of one width): the gathered rows are added instead of concatenated -- the
    library takes that form when every table is d_model wide."""
    _cuda(pe, *ids_list, *tables)
    B, S = ids_list[0].shape
    if combine == 'sum':
        d = int(tables[0].shape[1])
        if len(tables) < 2 or any(int(t.shape[1]) != d for t in tables):
            raise B4CError("combine='sum' needs two or more embedding tables of one width")
    elif combine == 'concat':
        d = sum(int(t.shape[1]) for t in tables)
    else:
        raise B4CError("combine must be 'concat' or 'sum', got %r" % (combine,))
    for i in ids_list:
        if i.dtype != torch.int64 or not i.is_contiguous() or tuple(i.shape) != (B, S):
            raise B4CError('embedding ids must be contiguous int64 (B,S) tensors of one shape')
    for t in tables:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise B4CError('embedding tables must be contiguous float32')
    if pe.shape[0] < S or pe.shape[1] != d:
        raise B4CError('positional table (%d,%d) too small for S=%d d=%d' % (pe.shape[0], pe.shape[1], S, d))
    n, ids_arr, tab_arr, dims, rows = _feature_arrays(ids_list, tables)
    if packed is not None:      # rows of the real tokens only
        T_tok = packed.T
        out = torch.empty(1, T_tok, d, dtype=dtype, device=pe.device)
        key_pad = torch.empty(T_tok, dtype=torch.uint8, device=pe.device)
        with _record('embed_fwd', T_tok * d * (4 + out.element_size())):
            L.check(L.lib().b4c_embed_concat_pe_fwd_packed(n, ids_arr, tab_arr, dims, rows, _p(pe), scale, _p(out), d, _p(key_pad),
                                                           B, S, d, rate, seed, _p(packed.tok_src), T_tok, dt_code(dtype), _st()),
                    'embed_concat_pe_fwd_packed')
        return out, key_pad
    out = torch.empty(B, S, d, dtype=dtype, device=pe.device)
    key_pad = torch.empty(B, S, dtype=torch.uint8, device=pe.device)
    with _record('embed_fwd', B * S * d * (4 + out.element_size())):
        L.check(L.lib().b4c_embed_concat_pe_fwd(n, ids_arr, tab_arr, dims, rows, _p(pe), scale, _p(out), d, _p(key_pad),
                                                B, S, d, rate, seed, dt_code(dtype), _st()), 'embed_concat_pe_fwd')
    return out, key_pad


library_sort = True      # b4c_sort_ids (6 - 9 launches) instead of torch.sort (14 launches through rocPRIM)


# ---- scratch memory the library's entry points take from the caller -----------------------------------------------------
# One growing buffer per (purpose, device, launch stream): launches of ONE stream use it one after the other, so the stream
# order keeps them apart; two models that train on streams of their own get buffers of their own (a buffer shared across
# streams would be written by two kernels at once).  Nothing here belongs to a training step: the buffers hold no results.
_workspaces = {}


def _workspace(kind, device, need, floor=0):
    """uint8 scratch tensor of at least `need` bytes for launches of `kind` on `device`'s current stream"""
    key = (kind, device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(int(need), int(floor), 1), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _sort_order(ids, n_rows):
    """Token indices sorted by id (int32).  Radix-sort cost grows with the key width, and ids are row indices of a
    table: 16-bit keys for tables of up to 65,536 rows (biased into int16), else 32-bit (63 / 178 / 224 us for 16 / 32 / 64
    bits at 819,200 tokens).  Out-of-range ids are clamped exactly as the kernels clamp them."""
    if library_sort and ids.is_cuda:
        flat = ids.reshape(-1)
        if flat.dtype != torch.int64 or not flat.is_contiguous():
            flat = flat.to(torch.int64).contiguous()
        n = flat.shape[0]
        order = torch.empty(n, dtype=torch.int32, device=ids.device)
        if n == 0:
            return order
        need = L.lib().b4c_sort_ids_workspace_bytes(n, n_rows)
        ws = _workspace('sort', ids.device, need, 1 << 20)
        L.check(L.lib().b4c_sort_ids(_p(flat), n, n_rows, _p(order), ws.data_ptr(), ws.numel(), _st()), 'sort_ids')
        return order
    flat = ids.view(-1).clamp(0, n_rows - 1)
    if n_rows <= 65536:
        keys = (flat - 32768).to(torch.int16)
    else:
        keys = flat.to(torch.int32)
    return torch.sort(keys, stable=True)[1].to(torch.int32)


sorted_embed_bwd = True     # sort the tokens of every feature by id (one radix sort per call) and sum runs in registers

# ops.deterministic (B4C_DETERMINISTIC=1): ONE switch for a bit-reproducible training step -- every reduction that otherwise
# finishes through float atomics (whose arrival order, and with it the last bit of the sum, changes from run to run) takes a
# fixed-order form instead:
#   the vocabulary projection's dW / db      one workgroup per vocabulary tile walks every token; label term through a stable sort
#   the embedding tables' gradient rows      runs of one id that cross the kernel's 64-entry ranges are summed in range order
#                                            (b4c_embed_concat_pe_bwd_sorted_ws); the unsorted atomic kernel is never taken
#   LayerNorm dgamma / dbeta                 per-workgroup sums in a fixed order, added block by block (b4c_add_dropout_layernorm_bwd_ws)
# (the dense weight gradients already are: ops.tn_deterministic).  Not covered: the sampled-softmax head's row scatter (config 5).
# Two identical runs of bench.py's step then give torch.equal arenas (tests/test_gpu_deterministic.py); cost: DESIGN.md.
deterministic = os.environ.get('B4C_DETERMINISTIC', '0') == '1'


def embed_concat_pe_bwd(ids_list, tables, dout, scale, rate, seed, into=None, order=None):
    B, S = ids_list[0].shape
    d = dout.shape[-1]
    dtabs = into if into is not None else [torch.zeros_like(t) for t in tables]
    n, ids_arr, tab_arr, dims, rows = _feature_arrays(ids_list, dtabs)
    if (sorted_embed_bwd and B * S >= 4096) or deterministic:
        if order is None:
            order = [_sort_order(i, int(t.shape[0])) for i, t in zip(ids_list, tables)]
        ord_arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in order])
        ws = None
        if deterministic:
            ws = _workspace('embed_bwd', dout.device, L.lib().b4c_embed_concat_pe_bwd_sorted_workspace_bytes(n, dims, B, S))
        with _record('embed_bwd', B * S * d * (4 + dout.element_size())):
            L.check(L.lib().b4c_embed_concat_pe_bwd_sorted_ws(n, ids_arr, ord_arr, tab_arr, dims, rows, scale, _p(dout), d, B, S, d,
                                                              rate, seed, _p(ws), ws.numel() if ws is not None else 0,
                                                              dt_code(dout.dtype), _st()), 'embed_concat_pe_bwd_sorted')
        return dtabs
    with _record('embed_bwd', B * S * d * (4 + dout.element_size())):
        L.check(L.lib().b4c_embed_concat_pe_bwd(n, ids_arr, tab_arr, dims, rows, scale, _p(dout), d, B, S, d, rate, seed,
                                                dt_code(dout.dtype), _st()), 'embed_concat_pe_bwd')
    return dtabs


def row_pitch(n):
    """Row pitch (elements) of a materialised [rows, n] vocabulary-wide tensor: n rounded up to 128 elements, so that every
    row starts on a 256-B boundary.  With V = 50,000 a bf16 row is 100,000 B and starts 32 B further into a 128-B line
    than the row above it: a pure store kernel with the projection's tile shape then writes 3.8 TB/s, with aligned rows
    5.5 TB/s (scratch/store_bw2.hip).  Narrow outputs keep their natural pitch."""
    return n if n < 2048 else (n + 127) // 128 * 128


def empty_rows(R, n, dtype, device):
    """An uninitialised [R, n] tensor on the row_pitch(n) pitch (a view of a [R, pitch] allocation; contiguous when the
    pitch is n)."""
    ld = row_pitch(n)
    buf = torch.empty(R, ld, dtype=dtype, device=device)
    return buf if ld == n else buf[:, :n]


def zero_(t):
    """zeros into a contiguous tensor through the library (hipMemsetAsync on the launch stream), in place"""
    if t.is_cuda and t.is_contiguous():
        if t.numel():
            L.check(L.lib().b4c_zero(_p(t), t.numel() * t.element_size(), _st()), 'zero')
        return t
    return t.zero_()


def zeros(*shape, dtype=torch.float32, device=None):
    """torch.zeros through the library's fill on a GPU (no PyTorch kernel in the training step's trace)"""
    t = torch.empty(*shape, dtype=dtype, device=device)
    return zero_(t) if t.is_cuda else t.zero_()


def chain_ids(seqs, cls, sep):
    """[cls, sep, seq_0, sep, seq_1, sep, ...] along axis 1 for (B, L_i) int64 id tensors on a GPU (b4c_chain_ids)"""
    B = seqs[0].shape[0]
    seqs = [s if (s.stride(-1) == 1 or s.shape[1] == 0) else s.contiguous() for s in seqs]
    S = 2 + sum(int(s.shape[1]) + 1 for s in seqs)
    out = torch.empty(B, S, dtype=torch.int64, device=seqs[0].device)
    n = len(seqs)
    ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in seqs])
    lens = (ctypes.c_int * n)(*[int(s.shape[1]) for s in seqs])
    pitches = (ctypes.c_int * n)(*[int(s.stride(0)) if s.shape[1] else 0 for s in seqs])
    L.check(L.lib().b4c_chain_ids(ptrs, lens, pitches, n, B, int(cls), int(sep), _p(out), S, _st()), 'chain_ids')
    return out


def _rows_ok(g, dtype):
    """g as the kernels take it: `dtype`, unit column stride, 16-B aligned rows -- copied only when it is not"""
    if g.dtype != dtype:
        g = g.to(dtype)
    if g.dim() != 2 or g.stride(1) != 1 or (g.stride(0) * g.element_size()) % 16 or g.data_ptr() % 16:
        g = g.contiguous()
    return g


def gemm_nt(a, bt, n, bias=None, act=L.ACT_NONE, gate=None, residual=None, out_dtype=None, out=None):
    """a: [M, Kp], bt: [>=n, Kp] -> [M, n] (written into `out`, a row-pitched [M, n] view, when given)"""
    M, K = a.shape
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty(M, n, dtype=out_dtype, device=a.device)
    elif out.dtype != out_dtype or tuple(out.shape) != (M, n) or out.stride(1) != 1:
        raise B4CError('gemm_nt: out must be a [%d, %d] %s view with unit column stride' % (M, n, out_dtype))
    if M == 0:
        return out
    es = a.element_size()
    nbytes = M * K * es + n * K * es + M * n * out.element_size() + (M * n * es if gate is not None else 0) + \
        (M * n * es if residual is not None else 0)
    # the materialised vocabulary projection (wide kernel: bf16, K <= 128, N >= 2048, plain epilogue) is its own family
    fam = 'vocab_proj' if (n >= 2048 and K <= 128 and a.dtype == torch.bfloat16 and out_dtype == torch.bfloat16 and
                           act == L.ACT_NONE and gate is None and residual is None) else 'gemm_nt'
    # token-sized launches ([T] rows: HBM-bound at 60 - 130 us each) and row-sized ones ([R] rows of the head trunk -- 10 - 60 us
    # each, matrix work -- and of the rows-only last layer, a few us each) are two families: one number would describe neither
    if fam == 'gemm_nt' and 2 * M < rec_hints.get('token_rows', 0):
        fam = 'gemm_nt_rows'
    with _record(fam, nbytes, 2 * M * n * K):
        L.check(L.lib().b4c_gemm_nt(_p(a), a.stride(0), _p(bt), bt.stride(0), _p(out), out.stride(0), M, n, K, _p(bias), act,
                                    _p(gate), gate.stride(0) if gate is not None else 0,
                                    _p(residual), residual.stride(0) if residual is not None else 0,
                                    dt_code(a.dtype), dt_code(out_dtype), _st()), 'gemm_nt')
    return out


def gemm_tn(a, g, K, N, want_bias=True, into=None):
    """dW[K,N] (+)= a[:, :K]^T @ g[:, :N] (fp32), db[N] (+)= colsum(g).
    into = (list of dW_i [K, N/len], list of db_i [N/len]): accumulate into existing fp32 tensors (column
    segments of equal width) instead of allocating zeros; returns (None, None) then."""
    M = a.shape[0]
    if M > 0 and (_rec is not None or family_log is not None):
        with _record('gemm_tn' if 2 * M >= rec_hints.get('token_rows', 0) else 'gemm_tn_rows',
                     M * (K + N) * a.element_size() + K * N * 4, 2 * M * K * N):
            return _gemm_tn_impl(a, g, K, N, want_bias, into)
    return _gemm_tn_impl(a, g, K, N, want_bias, into)


tn_deterministic = True     # False: no workspace, partial tiles are added with float atomics


def _tn_workspace(a, M, K, N):
    if not tn_deterministic or M == 0:
        return None, 0
    need = L.lib().b4c_gemm_tn_workspace_bytes(M, K, N, dt_code(a.dtype))
    if need == 0:
        return None, 0
    ws = _workspace('gemm_tn', a.device, need, 32 << 20)      # the split partial tiles of gemm_tn
    return ws.data_ptr(), ws.numel()


def _gemm_tn_impl(a, g, K, N, want_bias, into):
    M = a.shape[0]
    wsp, wsb = _tn_workspace(a, M, K, N)
    if into is not None:
        dWs, dbs = into
        if M == 0:
            return None, None
        n = len(dWs)
        if n == 1:
            L.check(L.lib().b4c_gemm_tn(_p(a), a.stride(0), _p(g), g.stride(0), _p(dWs[0]), N, _p(dbs[0]), M, K, N,
                                        dt_code(a.dtype), wsp, wsb, _st()), 'gemm_tn')
        else:
            wa = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dWs])
            ba = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dbs])
            L.check(L.lib().b4c_gemm_tn_seg(_p(a), a.stride(0), _p(g), g.stride(0), n, wa, ba, N // n, M, K,
                                            dt_code(a.dtype), wsp, wsb, _st()), 'gemm_tn_seg')
        return None, None
    dW = torch.zeros(K, N, dtype=torch.float32, device=a.device)
    db = torch.zeros(N, dtype=torch.float32, device=a.device) if want_bias else None
    if M == 0:
        return dW, db
    L.check(L.lib().b4c_gemm_tn(_p(a), a.stride(0), _p(g), g.stride(0), _p(dW), N, _p(db), M, K, N,
                                dt_code(a.dtype), wsp, wsb, _st()), 'gemm_tn')
    return dW, db


# ---- the whole backward of a Dense layer in one pass over its gradient (csrc/gemm_dxdw.hip) -----------------------------------
# Inside the C2 step (456 k token rows; main kernel + the fixed-order reduction of the workgroups' partial sums):
#   Q | K | V projection (three column blocks)   ~150 us against 111 + 147 us for b4c_gemm_nt + its share of the grouped b4c_gemm_tn
#   K | V of the masked-query last layer (two)    ~100 us against 84 + 91
#   attention output projection (one, no residual) ~70 us against 50 + 50
# B4C_FUSED_DXDW: 0 = off, 1 = Q | K | V only, 2 = + K | V, 3 (default) = + the output projection.
fused_dxdw = int(os.environ.get('B4C_FUSED_DXDW', '3'))


def dxdw_supported(x, g, n_seg):
    """bf16, 128-wide layer input, gradient of 128 x n_seg columns (n_seg 1 to 3), rows the kernels' 16-B accesses can take"""
    return x.dtype == torch.bfloat16 and g.dtype == torch.bfloat16 and x.shape[1] == 128 and n_seg in (1, 2, 3) and \
        g.shape[1] == 128 * n_seg and x.stride(0) % 8 == 0 and g.stride(0) % 8 == 0 and x.shape[0] >= 4096


def gemm_dxdw(x, g, wc, dWs, dbs, residual=None):
    """dX = g wc^T (+ residual), dW_s += x^T g_s, db_s += colsum(g_s) with g read once (b4c_gemm_dxdw).
    x [M, 128], g [M, 128 n_seg], wc [128, >= 128 n_seg] (the dX operand of gemm_nt), dWs / dbs: fp32 gradient tensors."""
    M, n_seg = x.shape[0], len(dWs)
    dx = torch.empty(M, 128, dtype=x.dtype, device=x.device)
    ws = _workspace('gemm_dxdw', x.device, L.lib().b4c_gemm_dxdw_workspace_bytes(M, n_seg))
    wa = (ctypes.c_void_p * n_seg)(*[t.data_ptr() for t in dWs])
    ba = (ctypes.c_void_p * n_seg)(*[(t.data_ptr() if t is not None else None) for t in dbs])
    es = 2
    with _record('gemm_dxdw', M * (128 * (1 + n_seg) + 128 + (128 if residual is not None else 0)) * es, 4 * M * 128 * 128 * n_seg):
        L.check(L.lib().b4c_gemm_dxdw(_p(x), x.stride(0), _p(g), g.stride(0), _p(wc), wc.stride(0), _p(residual),
                                      residual.stride(0) if residual is not None else 0, _p(dx), dx.stride(0), n_seg, wa, ba,
                                      dWs[0].stride(0), M, ws.data_ptr(), ws.numel(), _st()), 'gemm_dxdw')
    return dx


# ---- the feed-forward block's whole backward in one pass (csrc/ffn_bwd.hip): LayerNorm + dropout backward, both Dense layers' dX,
# dW and db.  B4C_FUSED_FFN_BWD=0 keeps the five kernels.
fused_ffn_bwd = os.environ.get('B4C_FUSED_FFN_BWD', '1') != '0'


def ffn_bwd_shape_ok(x, h, z):
    """the model shapes b4c_ffn_bwd takes: bf16, d_model = 128, dff <= 128 (padded to 8)"""
    return x.dtype == torch.bfloat16 and h.dtype == torch.bfloat16 and z.dtype == torch.bfloat16 and x.shape[1] == 128 and \
        h.shape[1] <= 128 and h.shape[1] % 8 == 0 and x.stride(0) % 8 == 0 and h.stride(0) % 8 == 0 and z.is_contiguous()


def ffn_bwd_supported(x, h, z):
    """... and enough rows for a persistent kernel (below 4,096 the five kernels are launched)"""
    return ffn_bwd_shape_ok(x, h, z) and x.shape[0] >= 4096


def ffn_bwd(dout, z, stats, gamma, rate, seed, h, x, wc2, wc1, F, dW1, db1, dW2, db2, dgamma, dbeta):
    """dX of the feed-forward block; dW1 / db1 / dW2 / db2 / dgamma / dbeta += (b4c_ffn_bwd).  wc2 [Fp][>= 128], wc1 [128][>= Fp]:
    the dX operands of gemm_nt for the second / first Dense layer."""
    M, Fp = x.shape[0], h.shape[1]
    dx = torch.empty(M, 128, dtype=x.dtype, device=x.device)
    ws = _workspace('ffn_bwd', x.device, L.lib().b4c_ffn_bwd_workspace_bytes(M))
    es = 2
    with _record('ffn_bwd' if 2 * M >= rec_hints.get('token_rows', 0) else 'ffn_bwd_rows', M * ((4 * 128 + Fp) * es + 8),
                 8 * M * 128 * Fp):
        L.check(L.lib().b4c_ffn_bwd(_p(dout), _p(z), _p(stats), _p(gamma), rate, seed, _p(h), h.stride(0), _p(x), x.stride(0),
                                    _p(wc2), wc2.stride(0), _p(wc1), wc1.stride(0), F, Fp, _p(dx), dx.stride(0),
                                    _p(dW1), dW1.stride(0), _p(db1), _p(dW2), dW2.stride(0), _p(db2), _p(dgamma), _p(dbeta),
                                    M, ws.data_ptr(), ws.numel(), _st()), 'ffn_bwd')
    return dx


# ---- the feed-forward block's forward in one pass (csrc/ffn_fwd.hip).  B4C_FUSED_FFN_FWD=0 keeps gemm_nt + gemm_nt_add_ln.
fused_ffn_fwd = os.environ.get('B4C_FUSED_FFN_FWD', '1') != '0'


def ffn_fwd_supported(x, Fp):
    return x.dtype == torch.bfloat16 and x.shape[1] == 128 and Fp <= 128 and Fp % 8 == 0 and x.stride(0) % 8 == 0 and x.shape[0] >= 4096


def ffn_fwd(x, wt1, b1, wt2, b2, gamma, beta, F, Fp, rate, seed, save=True):
    """-> h [M, Fp], z (None unless save), out, stats (None unless save) of LayerNorm(x + dropout(relu(x W1 + b1) W2 + b2)) (b4c_ffn_fwd).
    wt1 [Fp][128], wt2 [128][Fp]: the forward operands of gemm_nt for the two layers; b1 [Fp], b2 [128] fp32."""
    M = x.shape[0]
    h = torch.empty(M, Fp, dtype=x.dtype, device=x.device)
    z = torch.empty(M, 128, dtype=x.dtype, device=x.device) if save else None
    out = torch.empty(M, 128, dtype=x.dtype, device=x.device)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device) if save else None
    with _record('ffn_fwd' if 2 * M >= rec_hints.get('token_rows', 0) else 'ffn_fwd_rows',
                 M * ((128 * (3 if save else 2) + Fp) * 2 + (8 if save else 0)), 4 * M * 128 * Fp):
        L.check(L.lib().b4c_ffn_fwd(_p(x), x.stride(0), _p(wt1), wt1.stride(0), _p(b1), _p(wt2), wt2.stride(0), _p(b2), _p(gamma), _p(beta),
                                    F, Fp, _p(h), h.stride(0), _p(z), _p(out), _p(stats), M, LN_EPS, rate, seed, _st()), 'ffn_fwd')
    return h, z, out, stats


# ---- the attention block's tail in one pass (csrc/attn_out_bwd.hip): LayerNorm + dropout backward and the output projection's dX /
# dW / db.  B4C_FUSED_ATTN_OUT_BWD=0 keeps add_ln_bwd + the projection's own kernels.
fused_attn_out_bwd = os.environ.get('B4C_FUSED_ATTN_OUT_BWD', '1') != '0'


def attn_out_bwd_supported(o, z):
    return o.dtype == torch.bfloat16 and z.dtype == torch.bfloat16 and o.shape[1] == 128 and z.shape[1] == 128 and \
        o.stride(0) % 8 == 0 and z.is_contiguous() and o.shape[0] >= 4096


def attn_out_bwd(dout, z, stats, gamma, rate, seed, o, wc, dW, db, dgamma, dbeta):
    """-> dz [M, 128] (residual branch), d_o [M, 128];  dW / db / dgamma / dbeta += (b4c_attn_out_bwd).  wc [128][>= 128]: the dX
    operand of gemm_nt for the output projection."""
    M = o.shape[0]
    dz = torch.empty(M, 128, dtype=o.dtype, device=o.device)
    d_o = torch.empty(M, 128, dtype=o.dtype, device=o.device)
    ws = _workspace('attn_out_bwd', o.device, L.lib().b4c_attn_out_bwd_workspace_bytes(M))
    with _record('attn_out_bwd' if 2 * M >= rec_hints.get('token_rows', 0) else 'attn_out_bwd_rows', M * (5 * 128 * 2 + 8), 4 * M * 128 * 128):
        L.check(L.lib().b4c_attn_out_bwd(_p(dout), _p(z), _p(stats), _p(gamma), rate, seed, _p(o), o.stride(0), _p(wc), wc.stride(0),
                                         _p(dz), _p(d_o), d_o.stride(0), _p(dW), dW.stride(0), _p(db), _p(dgamma), _p(dbeta),
                                         M, ws.data_ptr(), ws.numel(), _st()), 'attn_out_bwd')
    return dz, d_o


# ---- grouped weight gradients: the dW GEMMs of an encoder layer are off the critical path (nothing in backward
# consumes them), so in arena mode they are queued and launched together (b4c_gemm_tn_group: one main + one reduce
# kernel per layer instead of four of each, and ~6x less partial-tile traffic).
grouped_dw = True


def queue_dw(c, a, g, K, N, dWs, dbs, params):
    """dW_i += a^T g (column segments), db_i += colsum(g) -- now or with the next flush_pending_dw(c).  `c`: the ArenaContext
    of the step in flight (the queue lives there: a step that fails leaves nothing behind for the next one)."""
    ok = c is not None and grouped_dw and a.dtype == torch.bfloat16 and a.stride(0) % 8 == 0 and g.stride(0) % 8 == 0 and \
        a.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0 and a.shape[0] >= 4096
    if not ok:
        gemm_tn(a, g, K, N, into=(dWs, dbs))
        _ready(*params)
        return
    q = c.pending_dw
    if q and (q[0][0].shape[0] != a.shape[0] or len(q) == 8):
        flush_pending_dw(c)
    q.append((a, g, K, N, dWs, dbs, params))


def flush_pending_dw(c):
    if c is None or not c.pending_dw:
        return
    items = list(c.pending_dw)
    del c.pending_dw[:]
    M = items[0][0].shape[0]
    descs = (L.TNDesc * len(items))()
    nbytes = flops = 0
    for d, (a, g, K, N, dWs, dbs, _) in zip(descs, items):
        d.A, d.G, d.lda, d.ldg, d.K = a.data_ptr(), g.data_ptr(), a.stride(0), g.stride(0), K
        d.n_seg, d.seg_width, d.ldw = len(dWs), N // len(dWs), dWs[0].stride(0)
        for i, (w, b) in enumerate(zip(dWs, dbs)):
            d.dW[i] = w.data_ptr()
            d.db[i] = b.data_ptr() if b is not None else None
        nbytes += M * (K + N) * a.element_size() + K * N * 4
        flops += 2 * M * K * N
    need = L.lib().b4c_gemm_tn_group_workspace_bytes(descs, len(items), M)
    ws = _workspace('gemm_tn', items[0][0].device, need, 32 << 20)
    with _record('gemm_tn' if 2 * M >= rec_hints.get('token_rows', 0) else 'gemm_tn_rows', nbytes, flops):
        L.check(L.lib().b4c_gemm_tn_group(descs, len(items), M, L.BF16, ws.data_ptr(), ws.numel(), _st()), 'gemm_tn_group')
    for it in items:
        _ready(*it[6])


# ---- per-arena host state ------------------------------------------------------------------------------------------
# optim.FlatArena re-homes parameters and their gradients into two flat buffers; the weight-gradient kernels then add
# straight into param.grad (autograd receives None for those inputs).  Everything the host keeps about such a training
# step -- who is told when a gradient is complete, and the side-stream work of the backward pass in flight -- lives in
# the arena's ArenaContext, reachable from every parameter of the arena as `p._b4c_ctx`.  Nothing here is process
# state: two models, with or without an arena, train side by side in one process (tests/test_gpu_context.py).
flash_ce = True          # bf16 training: vocabulary projection + CE without the (R x V) logits (csrc/vocab_ce.hip)


class ArenaContext:
    def __init__(self):
        self.grad_ready_cb = None     # parallel.GradReducer: called with each parameter whose gradient has just been produced
        # weight-gradient GEMMs of the layer in flight, queued for one grouped launch (queue_dw / flush_pending_dw): the
        # entries hold that step's activations and gradient tensors
        self.pending_dw = []
        # side-stream work (the vocabulary head's background dW sweep, see "Vocabulary-head weight gradient BESIDE ...")
        self.queue = []               # closures to run on the side stream, in order: pieces of sweeps, label terms, event records
        self.pending = []             # (event recorded on the side stream, parameters whose gradient it completes, main stream)
        self.slots = 0                # launch opportunities left in the current plan (one now, one per attention backward expected)
        self.counting = False         # inside a backward pass that feeds the queue
        self.kicks = 0                # attention-backward launches seen in this pass
        self.kicks_expected = 0       # ... in the previous backward pass: the plan of the next one
        self.side_launched = None     # device whose side stream has been given work since the last join (None: nothing in flight)
        # the model's feed-forward blocks take the fused backward (FFNBlockFn.forward sets it; the head's backward then runs its dW
        # sweep in the foreground: _background_dw_for)
        self.fused_blocks = False

    def reset(self):
        """Drop what a failed step left behind: queued weight-gradient GEMMs and side-stream closures hold that step's tensors
        and would add a stale gradient into the next one; side-stream kernels that step ALREADY launched (the first piece of the
        background sweep goes out before anything can fail) add into the gradient arena with float atomics, so the stream that
        is about to zero the arena waits for them first.  Called at the start of every step: FlatArena.zero_grad,
        GradReducer.begin_backward."""
        del self.pending_dw[:]
        del self.queue[:]
        if self.side_launched is not None:
            torch.cuda.current_stream(self.side_launched).wait_stream(_side_stream(self.side_launched))
            self.side_launched = None
        del self.pending[:]
        self.slots = self.kicks = 0
        self.counting = False


def arena_context(*params):
    """The ArenaContext shared by all of `params` whose .grad is a live view of that arena's gradient buffer, else None."""
    ctx = None
    for p in params:
        c = getattr(p, '_b4c_ctx', None)
        g = p.grad
        if c is None or (ctx is not None and c is not ctx):
            return None
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or g.device != p.device:
            return None
        ctx = c
    return ctx


def _inplace_ok(*params):
    return arena_context(*params) is not None


def _ready(*params):
    for p in params:
        c = getattr(p, '_b4c_ctx', None)
        if c is not None and c.grad_ready_cb is not None:
            c.grad_ready_cb(p)


def attn_mq_fwd(q, kv, cu, moff, B, max_len, H, dh, key_pad=None):
    """Attention of a few query rows per sequence against all of its keys (b4c_attn_mq_fwd).
    q [R, H*dh]; kv [T, 2*H*dh] (k | v); cu [B+1] token offsets; moff [B+1] query-row offsets -> (o [R, H*dh], lse [R, H])."""
    R = q.shape[0]
    # rows outside every [moff[b], moff[b+1]) range (the unused tail of the sync-free form) are not written: zeros, not garbage
    o = zeros(R, H * dh, dtype=q.dtype, device=q.device)
    lse = zeros(R, H, dtype=torch.float32, device=q.device)
    if R == 0:
        return o, lse
    es = q.element_size()
    # algorithmic work: every query row against the keys of its own sequence (host hint; R x max_len is an upper bound)
    with _record('attn_mq_fwd', kv.shape[0] * 2 * H * dh * es + 2 * R * H * dh * es, 4 * rec_hints.get('sum_q_len', R * max_len) * H * dh):
        L.check(L.lib().b4c_attn_mq_fwd(_p(q), q.stride(0), _p(kv), kv.stride(0), _p(key_pad), _p(cu), _p(moff), _p(o), o.stride(0),
                                        _p(lse), B, max_len, H, dh, dt_code(q.dtype), _st()), 'attn_mq_fwd')
    return o, lse


def attn_mq_bwd(q, kv, cu, moff, o, d_o, lse, B, max_len, H, dh, key_pad=None):
    """-> (dq [R, H*dh], dkv [T, 2*H*dh]); every token row of dkv is written (zeros where no query reads the sequence)."""
    R = q.shape[0]
    dq = zeros(q.shape[0], q.shape[1], dtype=q.dtype, device=q.device)
    if R == 0:                 # no query row anywhere: no key or value receives a gradient from this layer
        return dq, torch.zeros_like(kv)
    dkv = torch.empty_like(kv)
    if kv.shape[0] == 0:
        return dq, dkv
    es = q.element_size()
    with _record('attn_mq_bwd', kv.shape[0] * 4 * H * dh * es + 4 * R * H * dh * es, 10 * rec_hints.get('sum_q_len', R * max_len) * H * dh):
        L.check(L.lib().b4c_attn_mq_bwd(_p(q), q.stride(0), _p(kv), kv.stride(0), _p(key_pad), _p(cu), _p(moff), _p(o), o.stride(0),
                                        _p(d_o), d_o.stride(0), _p(lse), _p(dq), dq.stride(0), _p(dkv), dkv.stride(0), B, max_len,
                                        H, dh, dt_code(q.dtype), _st()), 'attn_mq_bwd')
    return dq, dkv


def attn_fwd(qkv, key_pad, B, S, H, dh, cu=None):
    """cu (int32 [B+1]): packed layout -- sequence b owns rows cu[b] .. cu[b+1] of qkv, S = upper bound of the longest one."""
    d = H * dh
    T_tok = qkv.shape[0]
    o = torch.empty(T_tok, d, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=qkv.device)
    # algorithmic work of the packed layout: sum over sequences of len^2 (host hint; T x S is an upper bound and what the
    # padded layout computes)
    pairs = rec_hints.get('sum_len_sq', T_tok * S) if cu is not None else T_tok * S
    with _record('attn_fwd', T_tok * 4 * d * qkv.element_size(), 4 * pairs * d):
        if cu is None:
            L.check(L.lib().b4c_attn_fwd(_p(qkv), qkv.stride(0), _p(key_pad), _p(o), d, _p(lse), B, S, H, dh,
                                         dt_code(qkv.dtype), _st()), 'attn_fwd')
        else:
            L.check(L.lib().b4c_attn_fwd_varlen(_p(qkv), qkv.stride(0), _p(key_pad), _p(cu), _p(o), d, _p(lse), B, S, H, dh,
                                                dt_code(qkv.dtype), _st()), 'attn_fwd_varlen')
    return o, lse


def attn_bwd(qkv, key_pad, o, d_o, lse, B, S, H, dh, cu=None, actx=None):
    """actx: the ArenaContext of the training step in flight (its background work gets a launch opportunity behind this kernel)"""
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    need = L.lib().b4c_attn_bwd_workspace_bytes(B, S, H, dh, dt_code(qkv.dtype))     # > 0 only for bf16 256 < S <= 512
    ws = _workspace('attn_bwd', qkv.device, need) if need else None
    T_tok = qkv.shape[0]
    pairs = rec_hints.get('sum_len_sq', T_tok * S) if cu is not None else T_tok * S
    with _record('attn_bwd', T_tok * 8 * H * dh * qkv.element_size(), 10 * pairs * H * dh):
        if cu is None:
            L.check(L.lib().b4c_attn_bwd_ws(_p(qkv), qkv.stride(0), _p(key_pad), _p(o), o.stride(0), _p(d_o), d_o.stride(0),
                                            _p(lse), _p(delta), _p(dqkv), dqkv.stride(0), B, S, H, dh, _p(ws), need,
                                            dt_code(qkv.dtype), _st()), 'attn_bwd')
        else:
            L.check(L.lib().b4c_attn_bwd_varlen(_p(qkv), qkv.stride(0), _p(key_pad), _p(cu), _p(o), o.stride(0), _p(d_o),
                                                d_o.stride(0), _p(lse), _p(delta), _p(dqkv), dqkv.stride(0), B, S, H, dh, _p(ws),
                                                need, dt_code(qkv.dtype), _st()), 'attn_bwd_varlen')
    _background_kick(actx)      # the next piece of the vocabulary head's dW sweep starts when this kernel has left the CUs
    return dqkv


def add_dropout_layernorm_fwd(x, y, gamma, beta, rate, seed, save=True):
    rows, d = x.shape
    z = torch.empty_like(x) if save else None
    out = torch.empty_like(x)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device) if save else None
    if rows == 0:
        return z, out, stats
    with _record('add_ln_fwd', rows * d * x.element_size() * (4 if save else 3)):
        L.check(L.lib().b4c_add_dropout_layernorm_fwd(_p(x), _p(y), _p(gamma), _p(beta), _p(z), _p(out), _p(stats), rows, d,
                                                      LN_EPS, rate, seed, dt_code(x.dtype), _st()), 'add_dropout_layernorm_fwd')
    return z, out, stats


fused_ln = True      # bf16, d_model <= 256: the GEMM in front of "x + dropout(y) -> LayerNorm" does it in its epilogue


def gemm_ln_supported(a, n):
    return fused_ln and a.dtype == torch.bfloat16 and n <= 256 and n % 8 == 0


def gemm_nt_add_ln(a, bt, bias, x, gamma, beta, rate, seed, save=True):
    """LayerNorm(x + dropout(a @ bt^T + bias)) in one kernel (== gemm_nt + add_dropout_layernorm_fwd, bit for bit).
    a: [M, Kp], bt: [>=n, Kp], x: [M, n] -> (z, out, stats) as add_dropout_layernorm_fwd."""
    M, K = a.shape
    n = x.shape[1]
    z = torch.empty_like(x) if save else None
    out = torch.empty_like(x)
    stats = torch.empty(M, 2, dtype=torch.float32, device=x.device) if save else None
    if M == 0:
        return z, out, stats
    es = a.element_size()
    with _record('gemm_nt_ln' if 2 * M >= rec_hints.get('token_rows', 0) else 'gemm_nt_ln_rows',
                 M * K * es + n * K * es + M * n * es * (3 if save else 2), 2 * M * n * K):
        L.check(L.lib().b4c_gemm_nt_add_ln(_p(a), a.stride(0), _p(bt), bt.stride(0), _p(bias), _p(x), x.stride(0), _p(gamma),
                                           _p(beta), _p(z), _p(out), _p(stats), M, n, K, LN_EPS, rate, seed,
                                           dt_code(a.dtype), _st()), 'gemm_nt_add_ln')
    return z, out, stats


def add_dropout_layernorm_bwd(dout, z, stats, gamma, rate, seed, into=None):
    rows, d = z.shape
    dz = torch.empty_like(z)
    dy = torch.empty_like(z) if rate > 0 else None
    if into is not None:
        dgamma, dbeta = into
    else:
        dgamma = torch.zeros(d, dtype=torch.float32, device=z.device)
        dbeta = torch.zeros(d, dtype=torch.float32, device=z.device)
    if rows == 0:          # no row at all (a batch without a [MASK]): nothing to add to dgamma / dbeta
        return dz, (dy if dy is not None else dz), dgamma, dbeta
    ws = _workspace('ln_bwd', z.device, L.lib().b4c_add_dropout_layernorm_bwd_workspace_bytes(rows, d)) if deterministic else None
    with _record('add_ln_bwd' if 2 * rows >= rec_hints.get('token_rows', 0) else 'add_ln_bwd_rows',
                 rows * d * z.element_size() * (4 if rate > 0 else 3)):
        L.check(L.lib().b4c_add_dropout_layernorm_bwd_ws(_p(dout), _p(z), _p(stats), _p(gamma), _p(dz), _p(dy), _p(dgamma),
                                                         _p(dbeta), rows, d, rate, seed, _p(ws), ws.numel() if ws is not None else 0,
                                                         dt_code(z.dtype), _st()),
                'add_dropout_layernorm_bwd')
    return dz, (dy if dy is not None else dz), dgamma, dbeta


def mask_positions(ids, value, cap=None, poison=None):
    """-> counts[B], offsets[B+1], flat_idx[cap], maxcount[1] (all int32, device).  More matches than `cap`: offsets stay
    within cap, maxcount comes back as -(longest row) - 1 and the int32 flag `poison` (optional) is set to -1 (include/b4c.h)."""
    _cuda(ids)
    B, S = ids.shape
    cap = B * S if cap is None else cap
    dev = ids.device
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    offsets = torch.empty(B + 1, dtype=torch.int32, device=dev)
    flat = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
    mx = torch.empty(1, dtype=torch.int32, device=dev)
    L.check(L.lib().b4c_mask_positions(_p(ids), B, S, value, _p(counts), _p(offsets), _p(flat), cap, _p(mx), _p(poison), _st()),
            'mask_positions')
    return counts, offsets, flat, mx


def poison_rows(x, flag):
    """x (contiguous last dim) := NaN, in place, when the device flag (int32 [1]) is negative; untouched otherwise."""
    x2 = x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x
    if x2.numel() and (x2.data_ptr() != x.data_ptr() or x2.stride(1) != 1):
        raise B4CError('poison_rows: needs a tensor whose leading dimensions merge into rows (got strides %s)' % (x.stride(),))
    if x2.numel() and flag is not None:
        code = 2 if x2.dtype == torch.int32 else dt_code(x2.dtype)          # B4C_I32: rows of ids, poisoned with -1
        L.check(L.lib().b4c_poison_rows(_p(x2), x2.stride(0), x2.shape[0], x2.shape[1], _p(flag), code, _st()), 'poison_rows')
    return x


def padded_index(counts, offsets, flat, B, M):
    out = torch.empty(B * M, dtype=torch.int32, device=counts.device)
    L.check(L.lib().b4c_padded_index(_p(counts), _p(offsets), _p(flat), B, M, _p(out), _st()), 'padded_index')
    return out


def gather_rows(src, idx, n_out):
    width = src.shape[1]
    out = torch.empty(n_out, width, dtype=src.dtype, device=src.device)
    if n_out == 0:
        return out
    L.check(L.lib().b4c_gather_rows(_p(src), src.stride(0), _p(idx), _p(out), width, n_out, width, dt_code(src.dtype),
                                    _st()), 'gather_rows')
    return out


def scatter_rows(src, idx, n_dst):
    n_src, width = src.shape
    if n_src == 0:
        return torch.zeros(n_dst, width, dtype=src.dtype, device=src.device)
    out = torch.empty(n_dst, width, dtype=src.dtype, device=src.device)
    L.check(L.lib().b4c_scatter_rows(_p(src), src.stride(0), _p(idx), _p(out), width, n_src, n_dst, width,
                                     dt_code(src.dtype), _st()), 'scatter_rows')
    return out


def softmax_rows(logits, V):
    """probabilities with the shape and the row pitch of `logits` ([R, W >= V], pad columns are not written)"""
    R, ld = logits.shape[0], logits.stride(0)
    probs = torch.empty(R, ld, dtype=logits.dtype, device=logits.device)
    if ld != logits.shape[1]:
        probs = probs[:, :logits.shape[1]]
    if R == 0:
        return probs
    with _record('softmax_rows', 2 * R * ld * logits.element_size()):
        L.check(L.lib().b4c_softmax_rows(_p(logits), ld, _p(probs), ld, R, V, dt_code(logits.dtype), _st()), 'softmax_rows')
    return probs


def sparse_ce_from_probs(probs, labels_f32, V, variant=L.CE_TF):
    R, ld = probs.shape[0], probs.stride(0)
    item = torch.empty(R, dtype=torch.float32, device=probs.device)
    nval = torch.zeros(1, dtype=torch.float32, device=probs.device)
    if R == 0:
        return item, nval
    L.check(L.lib().b4c_sparse_ce_from_probs(_p(probs), ld, _p(labels_f32), _p(item), _p(nval), R, V, variant,
                                             dt_code(probs.dtype), _st()), 'sparse_ce_from_probs')
    return item, nval


def softmax_ce_fwd_bwd_(logits, labels_i32, grad_scale, V, variant=L.CE_TF):
    """In place: logits -> grad_scale * dloss/dlogits.  Returns per-row loss."""
    R, ld = logits.shape[0], logits.stride(0)
    item = torch.empty(R, dtype=torch.float32, device=logits.device)
    if R == 0:
        return item
    with _record('softmax_ce', 2 * R * ld * logits.element_size()):
        L.check(L.lib().b4c_softmax_ce_fwd_bwd(_p(logits), ld, _p(labels_i32), _p(item), _p(grad_scale), R, V, variant,
                                               dt_code(logits.dtype), _st()), 'softmax_ce_fwd_bwd')
    return item


def label_scale(labels_i32, V):
    """-> fp32 [2]: 1 / n_valid (0 if no row is valid) and n_valid, valid = 0 <= label < V  (one launch)."""
    out = torch.empty(2, dtype=torch.float32, device=labels_i32.device)
    L.check(L.lib().b4c_label_scale(_p(labels_i32), labels_i32.shape[0], V, _p(out), _st()), 'label_scale')
    return out


def sum_scaled(item, scale, poison=None):
    """-> 0-dim fp32: scale[0] * sum(item) in a fixed order; NaN when the int32 flag `poison`[0] is negative."""
    out = torch.empty((), dtype=torch.float32, device=item.device)
    L.check(L.lib().b4c_sum_scaled(_p(item), item.shape[0], _p(scale), _p(poison), _p(out), _st()), 'sum_scaled')
    return out


def relu_gate(g, act):
    """g where act > 0, else 0 (same shape / dtype, contiguous, element count a multiple of 8)."""
    g, act = g.contiguous(), act.contiguous()
    out = torch.empty_like(g)
    if g.numel() == 0:
        return out
    L.check(L.lib().b4c_relu_gate(_p(g), _p(act), _p(out), g.numel(), dt_code(g.dtype), _st()), 'relu_gate')
    return out


def vocab_ce_supported(h, K):
    """The logits-free path: bf16 head input of width 64 / 128 (include/b4c.h b4c_vocab_ce_fwd)."""
    return h.dtype == torch.bfloat16 and K in (64, 128) and h.shape[1] == K


def _vce_workspace(h, R, V, K):
    return _workspace('vocab_ce', h.device, L.lib().b4c_vocab_ce_workspace_bytes(R, V, K))


def vocab_ce_fwd(h, wt, bias, labels_i32, grad_scale, V, variant=L.CE_TF):
    """h [R, K] bf16, wt [>=V, K] bf16, bias fp32 [>=V] -> (item_loss [R], dh [R, K] bf16 already scaled by
    grad_scale, rowscal [R, 8] for vocab_ce_dw).  The (R x V) logits never exist in memory."""
    _cuda(h)
    R, K = h.shape
    item = torch.empty(R, dtype=torch.float32, device=h.device)
    dh = torch.empty(R, K, dtype=h.dtype, device=h.device)
    rowscal = torch.empty(R, 8, dtype=torch.float32, device=h.device)
    if R == 0:
        return item, dh, rowscal
    ws = _vce_workspace(h, R, V, K)
    # algorithmic work: the logits GEMM and the P.W GEMM (the clipped-row sweep is data dependent and not counted)
    with _record('vocab_ce_fwd', R * K * 2 * 2 + V * K * 2, 4 * R * V * K):
        L.check(L.lib().b4c_vocab_ce_fwd(_p(h), h.stride(0), _p(wt), wt.stride(0), _p(bias), _p(labels_i32), _p(grad_scale),
                                         _p(item), _p(dh), dh.stride(0), _p(rowscal), ws.data_ptr(), ws.numel(), R, V, K,
                                         variant, _st()), 'vocab_ce_fwd')
    return item, dh, rowscal


def vocab_ce_dw(h, wt, bias, labels_i32, rowscal, V, dW, db):
    """dW [K, V] fp32 += d loss / d kernel, db [V] += d loss / d bias (second half of vocab_ce_fwd)."""
    R, K = h.shape
    if R == 0:
        return
    ws = _vce_workspace(h, R, V, K)
    with _record('vocab_ce_dw', R * K * 2 + V * K * 2 + V * K * 4, 4 * R * V * K):
        L.check(L.lib().b4c_vocab_ce_dw(_p(h), h.stride(0), _p(wt), wt.stride(0), _p(bias), _p(labels_i32), _p(rowscal),
                                        _p(dW), dW.stride(0), _p(db), ws.data_ptr(), ws.numel(), R, V, K, int(deterministic_vocab_dw or deterministic), _st()),
                'vocab_ce_dw')


def vocab_ce_dw_sweep(h, wt, bias, rowscal, V, dW, db, tile_begin, tile_end, background_workgroups=0):
    """the dlogit part of vocab_ce_dw for the 128-id vocabulary tiles [tile_begin, tile_end); background_workgroups > 0:
    as a background kernel of at most that many one-wave-per-SIMD workgroups (runs beside another stream's kernels)"""
    R, K = h.shape
    if R == 0 or tile_end <= tile_begin:
        return
    frac = (tile_end - tile_begin) / float((V + 127) // 128)
    with _record('vocab_ce_dw_bg' if background_workgroups > 0 else 'vocab_ce_dw',
                 int(frac * (R * K * 2 + V * K * 2 + V * K * 4)), int(frac * 4 * R * V * K)):
        L.check(L.lib().b4c_vocab_ce_dw_sweep(_p(h), h.stride(0), _p(wt), wt.stride(0), _p(bias), _p(rowscal), _p(dW), dW.stride(0),
                                              _p(db), R, V, K, tile_begin, tile_end, background_workgroups, int(deterministic_vocab_dw or deterministic), _st()),
                'vocab_ce_dw_sweep')


def vocab_ce_dw_labels(h, labels_i32, rowscal, V, dW, db):
    """the label term of vocab_ce_dw (dW[:, y] -= yd h_row, db[y] -= yd)"""
    R, K = h.shape
    if R == 0:
        return
    ws = _vce_workspace(h, R, V, K)
    L.check(L.lib().b4c_vocab_ce_dw_labels(_p(h), h.stride(0), _p(labels_i32), _p(rowscal), _p(dW), dW.stride(0), _p(db),
                                           ws.data_ptr(), ws.numel(), R, V, K, int(deterministic_vocab_dw or deterministic), _st()), 'vocab_ce_dw_labels')


fused_softmax_proj = True      # Dense(V, softmax) of the bf16 path in one pass over (R x V): lse sweep + softmax epilogue


def vocab_softmax(h, wt, bias, Np, V):
    """probs [R, Np] (bf16, row_pitch(Np) pitch) = softmax over the first V columns of h wt^T + bias; columns V.. are 0.
    The logits never reach HBM: b4c_vocab_lse recomputes them for the row lse, b4c_gemm_nt_softmax writes probabilities."""
    R, K = h.shape
    probs = empty_rows(R, Np, h.dtype, h.device)
    if R == 0:
        return probs
    ws = _vce_workspace(h, R, V, K)
    lse2 = torch.empty(R, dtype=torch.float32, device=h.device)
    with _record('vocab_lse', R * K * 2 + V * K * 2, 2 * R * V * K):
        L.check(L.lib().b4c_vocab_lse(_p(h), h.stride(0), _p(wt), wt.stride(0), _p(bias), _p(lse2), ws.data_ptr(), ws.numel(),
                                      R, V, K, _st()), 'vocab_lse')
    with _record('vocab_proj', R * K * 2 + Np * K * 2 + R * Np * 2, 2 * R * Np * K):
        L.check(L.lib().b4c_gemm_nt_softmax(_p(h), h.stride(0), _p(wt), wt.stride(0), _p(probs), probs.stride(0), R, Np, K,
                                            _p(bias), _p(lse2), _st()), 'gemm_nt_softmax')
    if Np != V:
        probs[:, V:].zero_()
    return probs


fused_rank = True      # bf16 scoring path: ranks / top-k ids without the (R x V) scores in memory (b4c_vocab_rank, b4c_vocab_topk)


def _rank_workspace(h, R, V, K):
    return _workspace('vocab_rank', h.device, L.lib().b4c_vocab_rank_workspace_bytes(R, V, K))


def vocab_rank(h, wt, bias, labels_i32, V):
    """rank [R] int32 of the label among the V scores h wt^T + bias (items ranked before it; ties -> lower index first;
    negative: no valid label).  The scores never exist in memory."""
    _cuda(h)
    R, K = h.shape
    rank = torch.empty(R, dtype=torch.int32, device=h.device)
    if R == 0:
        return rank
    ws = _rank_workspace(h, R, V, K)
    with _record('vocab_rank', R * K * 2 + V * K * 2, 2 * R * V * K):
        L.check(L.lib().b4c_vocab_rank(_p(h), h.stride(0), _p(wt), wt.stride(0), _p(bias), _p(labels_i32), _p(rank), ws.data_ptr(),
                                       ws.numel(), R, V, K, _st()), 'vocab_rank')
    return rank


def rank_metrics(rank, k):
    """-> (hit [R], ndcg [R]) fp32: [rank < k], [rank < k] / log2(rank + 2); rows with a negative rank: 0"""
    R = rank.shape[0]
    hit = torch.empty(R, dtype=torch.float32, device=rank.device)
    ndcg = torch.empty(R, dtype=torch.float32, device=rank.device)
    if R:
        L.check(L.lib().b4c_rank_metrics(_p(rank), R, k, _p(hit), _p(ndcg), _st()), 'rank_metrics')
    return hit, ndcg


def vocab_topk(h, wt, bias, V, k, labels_i32=None):
    """-> (idx [R, k] int32, hit, ndcg (when labels are given), overflow int32 [1]): the k best item ids of every row of
    h wt^T + bias in order, scores never in memory.  overflow[0] rows (ids -1) have too many ties at the selection
    threshold: rank those on materialised scores."""
    _cuda(h)
    R, K = h.shape
    idx = torch.empty(R, k, dtype=torch.int32, device=h.device)
    hit = torch.empty(R, dtype=torch.float32, device=h.device) if labels_i32 is not None else None
    ndcg = torch.empty(R, dtype=torch.float32, device=h.device) if labels_i32 is not None else None
    overflow = torch.empty(1, dtype=torch.int32, device=h.device)
    if R == 0:
        return idx, hit, ndcg, overflow.zero_()
    ws = _rank_workspace(h, R, V, K)
    with _record('vocab_topk', 2 * (R * K * 2 + V * K * 2), 4 * R * V * K):
        L.check(L.lib().b4c_vocab_topk(_p(h), h.stride(0), _p(wt), wt.stride(0), _p(bias), k, _p(idx), _p(labels_i32), _p(hit),
                                       _p(ndcg), _p(overflow), ws.data_ptr(), ws.numel(), R, V, K, _st()), 'vocab_topk')
    return idx, hit, ndcg, overflow


class VocabSoftmaxFn(torch.autograd.Function):
    """Dense(V, softmax) (head.py:36) on the head's trunk output, probabilities materialised once, logits never:
    apply(h [R, K] bf16, pack, V, kernel, bias) -> probs [R, Np].  Backward: the softmax Jacobian on the saved
    probabilities (b4c_softmax_rows_bwd), then the projection's dW / db / dh as MLPFn does."""

    @staticmethod
    def forward(ctx, h, pack, V, kernel, bias):
        h = h.contiguous()
        training = any(ctx.needs_input_grad)       # (grad mode is off inside forward: ask what autograd will want)
        wt, _, b = pack.get(h.dtype, h.shape[1], training)
        probs = vocab_softmax(h, wt, b, pack.Np, V)
        if training:
            ctx.save_for_backward(h, probs)
            ctx.pack, ctx.V, ctx.params = pack, V, (kernel, bias)
        return probs

    @staticmethod
    def backward(ctx, g):
        h, probs = ctx.saved_tensors
        pack = ctx.pack
        kernel, bias = ctx.params
        g = _rows_ok(g, probs.dtype)
        dlogits = softmax_rows_bwd(probs, g, ctx.V)
        _, wc, _ = pack.get(h.dtype, h.shape[1], True)
        actx = arena_context(kernel, bias)
        if actx is not None:
            queue_dw(actx, h, dlogits, pack.K, pack.N, [kernel.grad], [bias.grad], (kernel, bias))
            dW = db = None
        else:
            dW, db = gemm_tn(h, dlogits, pack.K, pack.N)
        with _timed('vocab_proj_dx'):
            dh = gemm_nt(dlogits, wc, h.shape[1])
        flush_pending_dw(actx)
        return dh, None, None, dW, db


topk_threshold = True     # threshold-selection kernel (one HBM read per row); False: per-thread sorted lists only


def topk_rows(scores, V, k, labels_i32=None):
    _cuda(scores)
    R, ld = scores.shape[0], scores.stride(0)
    idx = torch.empty(R, k, dtype=torch.int32, device=scores.device)
    hit = torch.empty(R, dtype=torch.float32, device=scores.device) if labels_i32 is not None else None
    ndcg = torch.empty(R, dtype=torch.float32, device=scores.device) if labels_i32 is not None else None
    if R == 0:
        return idx, hit, ndcg
    redo = torch.empty(R, dtype=torch.int32, device=scores.device) if topk_threshold else None
    with _record('topk_rows', R * ld * scores.element_size()):
        L.check(L.lib().b4c_topk_rows_ws(_p(scores), ld, R, V, k, _p(idx), _p(labels_i32), _p(hit), _p(ndcg), _p(redo),
                                         dt_code(scores.dtype), _st()), 'topk_rows')
    return idx, hit, ndcg


def adam_step_(p, g, m, v, lr_t, beta1, beta2, eps, grad_mul=1.0):
    with _record('adam', p.numel() * 28):
        L.check(L.lib().b4c_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr_t, beta1, beta2, eps, grad_mul, _st()),
                'adam_step')


def adam_rows_(p, g, m, v, stamp, ids, n, row_lo, rows, width, lr_hist, t, beta1, beta2, eps, grad_mul, mode):
    """b4c_adam_rows over one row-sparse table of the arena (optim.LazyRows): mode 0 = bring rows to step t, 1 = take step t.
    Booked bytes: the rows NAMED (an upper bound of the distinct rows; rec_hints['adam_distinct_rows'] when the host knows it)."""
    if n <= 0:
        return
    nrows = min(n, rec_hints.get('adam_distinct_rows', n)) if ids is not None else n
    with _record('adam' if mode == 1 else 'adam_catch_up', nrows * width * (32 if mode == 1 else 24) + (n * 8 if ids is not None else 0)):
        L.check(L.lib().b4c_adam_rows(_p(p), _p(g), _p(m), _p(v), _p(stamp), _p(ids), n, row_lo, rows, width, _p(lr_hist), t,
                                      beta1, beta2, eps, grad_mul, mode, _st()), 'adam_rows')


def rand64_host(seed, ctr):
    """Host restatement of b4c_rand64 (csrc/common.h): Threefry-2x32, 12 rounds, key = seed, counter = ctr (uint64 array)."""
    import numpy as np
    M = np.uint64(0xFFFFFFFF)

    def rotl(x, r):
        return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & M

    ctr = np.asarray(ctr, dtype=np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    k2 = np.uint64(0x1BD11BDA) ^ k0 ^ k1
    x0 = ((ctr & M) + k0) & M
    x1 = ((ctr >> np.uint64(32)) + k1) & M
    inject = [(k1, k2, 1), (k2, k0, 2), (k0, k1, 3)]
    rounds = [(13, 15, 26, 6), (17, 29, 16, 24), (13, 15, 26, 6)]
    for rs, (a0, a1, j) in zip(rounds, inject):
        for r in rs:
            x0 = (x0 + x1) & M
            x1 = rotl(x1, r) ^ x0
        x0 = (x0 + a0) & M
        x1 = (x1 + a1 + np.uint64(j)) & M
    return x0 | (x1 << np.uint64(32))


def keep_mask(seed, n, rate):
    """Host regeneration of a dropout keep-mask (tests): element e kept iff b4c_keep(seed, e, rate) --
    counter = e >> 2, one 16-bit uniform per element (csrc/common.h)."""
    import numpy as np
    e = np.arange(n, dtype=np.uint64)
    h = rand64_host(seed, e >> np.uint64(2))
    u16 = (h >> (np.uint64(16) * (e & np.uint64(3)))) & np.uint64(0xFFFF)
    t = np.float32(rate) * np.float32(65536.0)
    thr = int(t)
    if np.float32(thr) < t:
        thr += 1
    return u16 >= np.uint64(thr)


# --------------------------------------------------------------------------------------
# packed compute copies of (fused) dense layers
# --------------------------------------------------------------------------------------
_pack_registry = []      # weak references to every PackedLinear: stale ones are refreshed together, in one launch


class PackedLinear:
    """Compute-dtype copies of one dense layer whose Keras kernels [K, N_i] are fused along N:
    ``wt`` [Np][Kp] (forward operand, K-contiguous) and ``wc`` [Kp][Np] (backward-dX operand),
    ``bias`` fp32 [Np].  Zero-padded to multiples of 8.  When the masters change (optimizer step,
    load_state_dict, .to()) ALL stale layers are re-packed by one b4c_pack_weights_batched launch."""

    def __init__(self, kernels, biases):
        import weakref
        self.kernels, self.biases = list(kernels), list(biases)
        self.K = int(self.kernels[0].shape[0])
        self.Ns = [int(k.shape[1]) for k in self.kernels]
        self.N = sum(self.Ns)
        self.Np = rup8(self.N)
        self._cache = {}
        _pack_registry.append(weakref.ref(self))

    def _key(self):
        return (_weights_epoch,) + tuple(t._version for t in self.kernels + self.biases) + \
            tuple(t.data_ptr() for t in self.kernels + self.biases)

    def _descs(self, ent, Kp):
        out, off = [], 0
        for k, b, n in zip(self.kernels, self.biases, self.Ns):
            kk = k.detach()
            if kk.dtype != torch.float32 or not kk.is_contiguous():
                raise B4CError('dense kernels must be contiguous float32 [in, out]')
            out.append(L.PackDesc(kk.data_ptr(), b.detach().data_ptr(), ent['wt'].data_ptr(),
                                  ent['wc'].data_ptr() if ent['wc'] is not None else None, ent['bias'].data_ptr(),
                                  self.K, n, Kp, self.Np, off, 0))
            off += n
        return out

    def get(self, dtype, Kp, need_wc):
        dev = self.kernels[0].device
        ent = self._cache.get((dtype, Kp, dev))
        if ent is None:
            ent = {'key': None, 'wt': torch.zeros(self.Np, Kp, dtype=dtype, device=dev), 'wc': None,
                   'bias': torch.zeros(self.Np, dtype=torch.float32, device=dev)}
            self._cache[(dtype, Kp, dev)] = ent
        if need_wc and ent['wc'] is None:
            ent['wc'] = torch.zeros(Kp, self.Np, dtype=dtype, device=dev)
            ent['key'] = None
        if ent['key'] != self._key():
            repack_stale(dtype, dev)
        return ent['wt'], ent['wc'], ent['bias']

    def split_grads(self, dW, db):
        """dW [K, N] / db [N] of the fused layer -> per-kernel gradients."""
        gk, gb, off = [], [], 0
        for n in self.Ns:
            gk.append(dW[:, off:off + n].contiguous() if len(self.Ns) > 1 else dW)
            gb.append(db[off:off + n].contiguous() if len(self.Ns) > 1 else db)
            off += n
        return gk, gb


_desc_cache = {}


def repack_stale(dtype, dev):
    """One launch refreshes every allocated compute copy (this dtype / device) whose masters changed."""
    descs, ents, tiles = [], [], []
    alive = []
    for ref in _pack_registry:
        pk = ref()
        if pk is None:
            continue
        alive.append(ref)
        key = None
        for (dt, Kp, dv), ent in pk._cache.items():
            if dt != dtype or dv != dev:
                continue
            key = key or pk._key()
            if ent['key'] == key:
                continue
            descs += pk._descs(ent, Kp)
            ents.append((ent, key))
            tiles += [((pk.K + 63) // 64) * ((n + 63) // 64) for n in pk.Ns]       # 64 x 64 tiles (csrc/rowops.hip)
    _pack_registry[:] = alive
    if not descs:
        return
    # The launch is a (largest tile count) x (descriptors) grid whose surplus workgroups exit at once; one vocabulary
    # projection (1,564 tiles of 64 x 64 at C2) beside thirty encoder matrices (4 tiles each) would make that 50,000 workgroups
    # for 1,900 tiles of work.  Descriptors are ordered by size and launched in groups of similar size (a factor of 8).
    order = sorted(range(len(descs)), key=lambda i: tiles[i])
    descs = [descs[i] for i in order]
    tiles = [tiles[i] for i in order]
    sig = tuple((d.src, d.wt, d.wc, d.bias_src) for d in descs)
    cached = _desc_cache.get((dtype, dev))
    if cached is None or cached[0] != sig:
        arr = (L.PackDesc * len(descs))(*descs)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        cached = (sig, host.to(dev))
        _desc_cache[(dtype, dev)] = cached
    esz = ctypes.sizeof(L.PackDesc)
    with torch.no_grad():
        i0 = 0
        while i0 < len(descs):
            i1 = i0 + 1
            while i1 < len(descs) and tiles[i1] <= 8 * max(tiles[i0], 1):
                i1 += 1
            L.check(L.lib().b4c_pack_weights_batched(cached[1].data_ptr() + i0 * esz, i1 - i0, max(tiles[i1 - 1], 1),
                                                     dt_code(dtype), _st()), 'pack_weights_batched')
            i0 = i1
    for ent, key in ents:
        ent['key'] = key


def _as2d(x):
    return x.reshape(-1, x.shape[-1])


# --------------------------------------------------------------------------------------
# autograd blocks
# --------------------------------------------------------------------------------------
class EmbedFn(torch.autograd.Function):
    """R6: gather + concat + *sqrt(d) + PE (+ input dropout).  apply(pe, scale, rate, seed, dtype, n, *ids, *tables);
    n may be (n, Packed): the packed layout, output (1, T, d); or (n, Packed | None, 'sum'): the features' rows are added."""

    @staticmethod
    def forward(ctx, pe, scale, rate, seed, dtype, n, *args):
        packed, combine = None, 'concat'
        if isinstance(n, tuple):
            if len(n) == 3:
                n, packed, combine = n
            else:
                n, packed = n
        ids, tables = list(args[:n]), list(args[n:])
        dense_ids = ids
        if packed is not None:
            # backward works on the packed ids (one gather per feature): B = 1, S = T rows
            pk_ids = []
            for i in ids:
                flat = i.reshape(-1)
                if flat.dtype != torch.int64 or not flat.is_contiguous():
                    flat = flat.to(torch.int64).contiguous()
                o = torch.empty(1, packed.T, dtype=torch.int64, device=flat.device)
                L.check(L.lib().b4c_gather_i64(_p(flat), _p(packed.tok_src), _p(o), packed.T, _st()), 'gather_i64')
                pk_ids.append(o)
            ids = pk_ids
        # a table under a row-lazy optimizer (optim.LazyRows): the rows about to be read are brought up to date first
        for j, (t, i) in enumerate(zip(tables, ids)):
            lz = getattr(t, '_b4c_lazy', None)
            if lz is not None:
                lz.catch_up(i, note=ctx.needs_input_grad[6 + n + j])
        out, key_pad = embed_concat_pe_fwd(dense_ids, [t.detach() for t in tables], pe, scale, rate, seed, dtype, packed, combine)
        ctx.save_for_backward(*ids, *tables)
        ctx.n, ctx.scale, ctx.rate, ctx.seed = n, scale, rate, seed
        ctx.mark_non_differentiable(key_pad)
        ctx.set_materialize_grads(False)       # (or autograd fills a zero "gradient" of key_pad's size every step)
        return out, key_pad

    @staticmethod
    def backward(ctx, dout, _):
        saved = ctx.saved_tensors
        ids, tables = list(saved[:ctx.n]), list(saved[ctx.n:])
        flush_pending_dw(getattr(tables[0], '_b4c_ctx', None))
        if dout is None:
            return (None,) * (6 + 2 * ctx.n)
        dout = dout.reshape(ids[0].shape[0], ids[0].shape[1], -1)
        if _inplace_ok(*tables):
            embed_concat_pe_bwd(ids, tables, dout.contiguous(), ctx.scale, ctx.rate, ctx.seed, into=[t.grad for t in tables])
            _ready(*tables)
            return (None,) * (6 + 2 * ctx.n)
        dtabs = embed_concat_pe_bwd(ids, tables, dout.contiguous(), ctx.scale, ctx.rate, ctx.seed)
        return (None,) * 6 + (None,) * ctx.n + tuple(dtabs)


class AttnBlockFn(torch.autograd.Function):
    """R8 + first half of R10: LN1(x + drop(MHA(x))).  x: [T, d] in the compute dtype."""

    @staticmethod
    def forward(ctx, x, key_pad, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, pk_qkv, pk_o, B, S, H, rate, seed, training, cu=None):
        T_tok, d = x.shape
        dh = d // H
        wt_qkv, _, b_qkv = pk_qkv.get(x.dtype, d, training)
        wt_o, _, b_o = pk_o.get(x.dtype, d, training)
        with _timed('qkv_fwd'):
            qkv = gemm_nt(x, wt_qkv, 3 * d, b_qkv)
        with _timed('attn_fwd'):
            o, lse = attn_fwd(qkv, key_pad, B, S, H, dh, cu)
        if gemm_ln_supported(o, d):
            z, out, stats = gemm_nt_add_ln(o, wt_o, b_o, x, gamma.detach(), beta.detach(), rate if training else 0.0, seed,
                                           save=training)
        else:
            y = gemm_nt(o, wt_o, d, b_o)
            z, out, stats = add_dropout_layernorm_fwd(x, y, gamma.detach(), beta.detach(), rate if training else 0.0, seed,
                                                      save=training)
        if training:
            ctx.save_for_backward(x, key_pad, qkv, o, lse, z, stats, gamma)
            ctx.pk = (pk_qkv, pk_o)
            ctx.dims = (B, S, H, dh, rate, seed)
            ctx.cu = cu
            ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, key_pad, qkv, o, lse, z, stats, gamma = ctx.saved_tensors
        pk_qkv, pk_o = ctx.pk
        B, S, H, dh, rate, seed = ctx.dims
        wq, bq, wk, bk, wv, bv, wo, bo, gam, bet = ctx.params
        d = H * dh
        actx = arena_context(*ctx.params)
        inplace = actx is not None
        _, wc_o, _ = pk_o.get(x.dtype, d, True)
        _, wc_qkv, _ = pk_qkv.get(x.dtype, d, True)
        if inplace and fused_attn_out_bwd and attn_out_bwd_supported(o, z):
            dz, d_o = attn_out_bwd(dout.contiguous(), z, stats, gamma.detach(), rate, seed, o, wc_o, wo.grad, bo.grad, gam.grad, bet.grad)
            _ready(wo, bo)
        else:
            dz, dy, dgamma, dbeta = add_dropout_layernorm_bwd(dout.contiguous(), z, stats, gamma.detach(), rate, seed,
                                                              into=(gam.grad, bet.grad) if inplace else None)
            if inplace and fused_dxdw >= 3 and dxdw_supported(o, dy, 1):
                d_o = gemm_dxdw(o, dy, wc_o, [wo.grad], [bo.grad])
                _ready(wo, bo)
            else:
                if inplace:
                    queue_dw(actx, o, dy, d, d, [wo.grad], [bo.grad], (wo, bo))
                else:
                    dWo, dbo = gemm_tn(o, dy, d, d)
                d_o = gemm_nt(dy, wc_o, d)
        with _timed('attn_bwd'):
            dqkv = attn_bwd(qkv, key_pad, o, d_o, lse, B, S, H, dh, ctx.cu, actx)
        if inplace and fused_dxdw and dxdw_supported(x, dqkv, 3):
            # dX and dW | db of the fused Q | K | V projection in one pass over dqkv (csrc/gemm_dxdw.hip)
            dx = gemm_dxdw(x, dqkv, wc_qkv, [wq.grad, wk.grad, wv.grad], [bq.grad, bk.grad, bv.grad], residual=dz)
            _ready(wq, bq, wk, bk, wv, bv)
        else:
            if inplace:
                queue_dw(actx, x, dqkv, d, 3 * d, [wq.grad, wk.grad, wv.grad], [bq.grad, bk.grad, bv.grad], (wq, bq, wk, bk, wv, bv))
            else:
                dWqkv, dbqkv = gemm_tn(x, dqkv, d, 3 * d)
            dx = gemm_nt(dqkv, wc_qkv, d, residual=dz)
        if inplace:
            _ready(gam, bet)
            flush_pending_dw(actx)      # this layer's four weight gradients (two queued by FFNBlockFn.backward) in one launch
            return (dx,) + (None,) * 20
        (gq, gk, gv), (gbq, gbk, gbv) = pk_qkv.split_grads(dWqkv, dbqkv)
        return (dx, None, gq, gbq, gk, gbk, gv, gbv, dWo, dbo, dgamma, dbeta) + (None,) * 9


def rows_add_(dst, idx, src):
    """dst[idx[r]] += src[r] for idx[r] >= 0 (distinct indices)."""
    if src.shape[0] == 0:
        return dst
    L.check(L.lib().b4c_rows_add(_p(dst), dst.stride(0), _p(idx), _p(src), src.stride(0), src.shape[0], src.shape[1],
                                 dt_code(dst.dtype), dt_code(src.dtype), _st()), 'rows_add')
    return dst


# Cloze path: the last encoder layer runs for the [MASK] rows only (MQAttnBlockFn); B4C_MQ_LAST_LAYER=0 computes the full layer
mq_last_layer = os.environ.get('B4C_MQ_LAST_LAYER', '1') != '0'


class MQAttnBlockFn(torch.autograd.Function):
    """The attention half of the LAST encoder layer for the query rows `midx` only: LN1(x_m + drop(MHA(x)_m)).
    Every other row of that layer's output is never read on the Cloze path (clickstream_transformer.py:281-295 keeps the
    [MASK] rows); keys and values still come from every token.
      x [T, d] (all tokens), midx [R] int32 (token row of each query row, -1 = unused row), moff [B+1] (query rows of
      sequence b), cu [B+1] (its token rows), key_pad [T] or None  ->  out1_m [R, d]."""

    @staticmethod
    def forward(ctx, x, midx, moff, cu, key_pad, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, pk_qkv, pk_o, B, max_len, H, rate, seed,
                training):
        T_tok, d = x.shape
        dh = d // H
        R = midx.shape[0]
        wt_qkv, _, b_qkv = pk_qkv.get(x.dtype, d, training)
        wt_o, _, b_o = pk_o.get(x.dtype, d, training)
        kv = gemm_nt(x, wt_qkv[d:3 * d], 2 * d, b_qkv[d:3 * d])            # K | V of every token
        x_m = gather_rows(x, midx, R)
        q_m = gemm_nt(x_m, wt_qkv[:d], d, b_qkv[:d])
        o_m, lse = attn_mq_fwd(q_m, kv, cu, moff, B, max_len, H, dh, key_pad)
        if gemm_ln_supported(o_m, d):
            z, out, stats = gemm_nt_add_ln(o_m, wt_o, b_o, x_m, gamma.detach(), beta.detach(), rate if training else 0.0, seed,
                                           save=training)
        else:
            y = gemm_nt(o_m, wt_o, d, b_o)
            z, out, stats = add_dropout_layernorm_fwd(x_m, y, gamma.detach(), beta.detach(), rate if training else 0.0, seed,
                                                      save=training)
        if training:
            ctx.save_for_backward(x, x_m, midx, moff, cu, key_pad, q_m, kv, o_m, lse, z, stats, gamma)
            ctx.pk = (pk_qkv, pk_o)
            ctx.dims = (B, max_len, H, dh, rate, seed)
            ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, x_m, midx, moff, cu, key_pad, q_m, kv, o_m, lse, z, stats, gamma = ctx.saved_tensors
        pk_qkv, pk_o = ctx.pk
        B, max_len, H, dh, rate, seed = ctx.dims
        wq, bq, wk, bk, wv, bv, wo, bo, gam, bet = ctx.params
        d = H * dh
        actx = arena_context(*ctx.params)
        inplace = actx is not None
        _, wc_o, _ = pk_o.get(x.dtype, d, True)
        _, wc_qkv, _ = pk_qkv.get(x.dtype, d, True)
        if inplace and fused_attn_out_bwd and attn_out_bwd_supported(o_m, z):
            dz, d_o = attn_out_bwd(dout.contiguous(), z, stats, gamma.detach(), rate, seed, o_m, wc_o, wo.grad, bo.grad, gam.grad, bet.grad)
            _ready(wo, bo)
        else:
            dz, dy, dgamma, dbeta = add_dropout_layernorm_bwd(dout.contiguous(), z, stats, gamma.detach(), rate, seed,
                                                              into=(gam.grad, bet.grad) if inplace else None)
            if inplace:
                queue_dw(actx, o_m, dy, d, d, [wo.grad], [bo.grad], (wo, bo))
            else:
                dWo, dbo = gemm_tn(o_m, dy, d, d)
            d_o = gemm_nt(dy, wc_o, d)
        dq, dkv = attn_mq_bwd(q_m, kv, cu, moff, o_m, d_o, lse, B, max_len, H, dh, key_pad)
        if inplace:
            queue_dw(actx, x_m, dq, d, d, [wq.grad], [bq.grad], (wq, bq))
            flush_pending_dw(actx)                  # (the query-row problems have R rows, the key / value problem T)
        else:
            dWq, dbq = gemm_tn(x_m, dq, d, d)
        dx_m = gemm_nt(dq, wc_qkv[:, :d], d, residual=dz, out_dtype=torch.float32)      # query rows: through Wq + the residual branch (kept in fp32 until it joins dx)
        # every token: through Wk | Wv
        if inplace and fused_dxdw >= 2 and dxdw_supported(x, dkv, 2):
            dx = gemm_dxdw(x, dkv, wc_qkv[:, d:3 * d], [wk.grad, wv.grad], [bk.grad, bv.grad])
            _ready(wk, bk, wv, bv)
        else:
            if inplace:
                queue_dw(actx, x, dkv, d, 2 * d, [wk.grad, wv.grad], [bk.grad, bv.grad], (wk, bk, wv, bv))
            else:
                dWkv, dbkv = gemm_tn(x, dkv, d, 2 * d)
            dx = gemm_nt(dkv, wc_qkv[:, d:3 * d], d)
        rows_add_(dx, midx, dx_m)
        if inplace:
            _ready(gam, bet)
            flush_pending_dw(actx)
            return (dx,) + (None,) * 22
        dWk, dWv = dWkv[:, :d].contiguous(), dWkv[:, d:].contiguous()
        dbk, dbv = dbkv[:d].contiguous(), dbkv[d:].contiguous()
        return (dx, None, None, None, None, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo, dgamma, dbeta) + (None,) * 8


class FFNBlockFn(torch.autograd.Function):
    """R9 + second half of R10: LN2(x + drop(relu(x W1 + b1) W2 + b2))."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, pk1, pk2, rate, seed, training):
        T_tok, d = x.shape
        wt1, _, bb1 = pk1.get(x.dtype, d, training)
        Fp = pk1.Np
        wt2, _, bb2 = pk2.get(x.dtype, Fp, training)
        if fused_ffn_fwd and ffn_fwd_supported(x, Fp):
            h, z, out, stats = ffn_fwd(x, wt1, bb1, wt2, bb2, gamma.detach(), beta.detach(), pk1.N, Fp, rate if training else 0.0, seed,
                                       save=training)
        elif gemm_ln_supported((h := gemm_nt(x, wt1, Fp, bb1, act=L.ACT_RELU)), d):
            z, out, stats = gemm_nt_add_ln(h, wt2, bb2, x, gamma.detach(), beta.detach(), rate if training else 0.0, seed,
                                           save=training)
        else:
            y = gemm_nt(h, wt2, d, bb2)
            z, out, stats = add_dropout_layernorm_fwd(x, y, gamma.detach(), beta.detach(), rate if training else 0.0, seed,
                                                      save=training)
        if training:
            ctx.save_for_backward(x, h, z, stats, gamma)
            ctx.pk = (pk1, pk2)
            ctx.dims = (rate, seed)
            ctx.params = (w1, b1, w2, b2, gamma, beta)
            # A model of the fused kernels' shape runs the head's dW sweep in the foreground (_background_dw_for) -- decided by the
            # SHAPE, whatever the batch's row count, so that it agrees with background_dw_expected(), by which callers order their
            # gradient arenas (a background sweep's gradient completes last and belongs to the reducer's last bucket)
            if fused_ffn_bwd and ffn_bwd_shape_ok(x, h, z):
                actx = arena_context(w1, b1, w2, b2, gamma, beta)
                if actx is not None:
                    actx.fused_blocks = True
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h, z, stats, gamma = ctx.saved_tensors
        pk1, pk2 = ctx.pk
        rate, seed = ctx.dims
        w1, b1, w2, b2, gam, bet = ctx.params
        d, Fp = x.shape[1], h.shape[1]
        actx = arena_context(*ctx.params)
        inplace = actx is not None
        _, wc1, _ = pk1.get(x.dtype, d, True)
        _, wc2, _ = pk2.get(x.dtype, Fp, True)
        if inplace and fused_ffn_bwd and ffn_bwd_supported(x, h, z):
            dx = ffn_bwd(dout.contiguous(), z, stats, gamma.detach(), rate, seed, h, x, wc2, wc1, pk1.N, w1.grad, b1.grad,
                         w2.grad, b2.grad, gam.grad, bet.grad)
            _ready(w1, b1, w2, b2, gam, bet)
            return (dx,) + (None,) * 11
        dz, dy, dgamma, dbeta = add_dropout_layernorm_bwd(dout.contiguous(), z, stats, gamma.detach(), rate, seed,
                                                          into=(gam.grad, bet.grad) if inplace else None)
        if inplace:
            queue_dw(actx, h, dy, pk2.K, d, [w2.grad], [b2.grad], (w2, b2))
        else:
            dW2, db2 = gemm_tn(h, dy, pk2.K, d)
        dh = gemm_nt(dy, wc2, Fp, gate=h)
        if inplace:
            queue_dw(actx, x, dh, d, pk1.N, [w1.grad], [b1.grad], (w1, b1))
        else:
            dW1, db1 = gemm_tn(x, dh, d, pk1.N)
        dx = gemm_nt(dh, wc1, d, residual=dz)
        if inplace:
            _ready(gam, bet)
            return (dx,) + (None,) * 11
        return (dx, dW1, db1, dW2, db2, dgamma, dbeta) + (None,) * 5


class MLPFn(torch.autograd.Function):
    """Chain of dense layers (relu on all but the last): the SoftMaxHead's trunk + vocabulary
    projection (R12, logits).  apply(x, packs, training, out_fp32, *[k0, b0, k1, b1, ...])"""

    @staticmethod
    def forward(ctx, x, packs, training, out_fp32, *params):
        # out_fp32 == 'relu_last': every layer (also the last) is followed by relu -- the head's trunk alone
        relu_last = out_fp32 == 'relu_last'
        acts = [x]
        n = len(packs)
        for i, pk in enumerate(packs):
            a = acts[-1]
            wt, _, bias = pk.get(x.dtype, a.shape[1], training)
            last = i == n - 1
            odt = torch.float32 if (last and out_fp32 is True) else a.dtype
            with _timed('vocab_proj_fwd' if last else 'head_mlp_fwd'):
                acts.append(gemm_nt(a, wt, pk.Np, bias, act=L.ACT_RELU if (relu_last or not last) else L.ACT_NONE,
                                    out_dtype=odt, out=empty_rows(a.shape[0], pk.Np, odt, a.device)))
        if training:
            ctx.save_for_backward(*(acts if relu_last else acts[:-1]))
            ctx.packs = packs
            ctx.params = params
            ctx.relu_last = relu_last
        return acts[-1]

    @staticmethod
    def backward(ctx, dlogits):
        acts = ctx.saved_tensors
        packs = ctx.packs
        g = dlogits
        if g.dtype != acts[0].dtype:
            g = g.to(acts[0].dtype)
        if ctx.relu_last:
            if g.is_cuda and g.numel() % 8 == 0 and g.dtype == acts[-1].dtype:
                g = relu_gate(g, acts[-1])
            else:
                g = g * (acts[-1] > 0).to(g.dtype)
        g = _rows_ok(g, acts[0].dtype)      # a pitched [R, Vp] view (empty_rows) is taken as it is
        grads = [None] * (2 * len(packs))
        dx = None
        actx = arena_context(*ctx.params)
        inplace = actx is not None
        for i in range(len(packs) - 1, -1, -1):
            a = acts[i]
            pk = packs[i]
            _, wc, _ = pk.get(a.dtype, a.shape[1], True)
            last = i == len(packs) - 1
            kern, bias = ctx.params[2 * i], ctx.params[2 * i + 1]
            if inplace:          # queued: the layers' weight gradients go out as one grouped launch below
                queue_dw(actx, a, g, pk.K, pk.N, [kern.grad], [bias.grad], (kern, bias))
                dW = db = None
            else:
                dW, db = gemm_tn(a, g, pk.K, pk.N)
            grads[2 * i], grads[2 * i + 1] = dW, db
            with _timed('vocab_proj_dx' if last else 'head_mlp_dx'):
                g = gemm_nt(g, wc, a.shape[1], gate=a if i > 0 else None)
            dx = g
        flush_pending_dw(actx)
        return (dx, None, None, None) + tuple(grads)


# Vocabulary-head weight gradient BESIDE the encoder backward.  The dW sweep is MFMA / VALU bound and leaves HBM idle;
# the encoder backward is HBM bound and leaves the matrix pipes idle; nothing in backward consumes the sweep's result.
# As a plain second-stream launch the two only time-slice the CUs (the sweep's 512-thread workgroups hold 2 x 212 registers
# per SIMD: nothing else fits beside them; 10.78 -> 10.71 ms).  What shares the CUs is the BACKGROUND form of the sweep
# (b4c_vocab_ce_dw_sweep, background_workgroups > 0): one wave per SIMD, at most one workgroup per CU, cut into pieces
# that start in the windows BETWEEN the resident attention backward kernels -- those take a whole CU's LDS (146 KB) and
# cannot start on a CU that holds a sweep workgroup.  A piece is launched on the side stream right after every attention
# backward launch (ops.attn_bwd -> _background_kick); how many such launches a backward pass has is learned from the
# previous pass.  The main stream joins, and the gradient is announced to a reducer, when backward ends (join_side_work).
# B4C_OVERLAP_DW=0 switches it off (the sweep then runs in the foreground, first thing in backward), =1 forces it on.
# Unset (None): decided per model.  The fused backward kernels of the d_model = 128 encoder (b4c_ffn_bwd, b4c_gemm_dxdw) hold a
# whole CU like the resident attention backward -- 97 - 145 KB of LDS, 2 x 255 registers per SIMD lane -- so a background sweep
# finds no CU to share and the two only block each other (C2, interleaved on one box: fused + foreground 8.92 - 9.00 ms, five
# kernels + background 9.02 - 9.04, fused + background 9.18 - 9.25, five kernels + foreground 9.59).  A model whose feed-forward
# blocks take the fused backward runs the sweep in the foreground; every other model keeps the background form.
overlap_vocab_dw = {'1': True, '0': False}.get(os.environ.get('B4C_OVERLAP_DW', ''), None)


def background_dw_expected(d_model, dff, dtype):
    """Will a model of this shape run the vocabulary head's dW sweep as a background job (see overlap_vocab_dw)?"""
    if overlap_vocab_dw is not None:
        return overlap_vocab_dw
    return not (fused_ffn_bwd and d_model == 128 and dff <= 128 and dtype == torch.bfloat16)


def _background_dw_for(actx):
    if overlap_vocab_dw is not None:
        return overlap_vocab_dw
    return not getattr(actx, 'fused_blocks', False)
background_workgroups = int(os.environ.get('B4C_VCE_DW_BG', '0'))      # 0: one per CU of the device
# deterministic_vocab_dw: the projection's dW / db are summed in a fixed order (no float atomics: one workgroup per
# vocabulary tile walks every token, the label term goes through a stable sort of the rows by label) -- two identical steps
# give a bit-identical projection gradient (LayerNorm dgamma / dbeta and a few embedding rows still meet through float atomics).
# Default off: the split sweep fills the CUs better (B4C_DETERMINISTIC_DW=1 switches it on).
deterministic_vocab_dw = os.environ.get('B4C_DETERMINISTIC_DW', '0') == '1'


def background_wgs(device):
    """workgroups of a background sweep: ops.background_workgroups, or one per CU of `device` (256 on an MI355X)"""
    if background_workgroups > 0:
        return background_workgroups
    return int(torch.cuda.get_device_properties(device).multi_processor_count)


# the windows' relative lengths: the first one (head trunk + the rows-only last layer + half a layer), one per layer, the last
# one (half a layer + the embedding backward)
background_weights = tuple(float(x) for x in os.environ.get('B4C_VCE_DW_WEIGHTS', '2.2,1.0,0.6').split(','))
_side_streams = {}


def _side_stream(device):
    s = _side_streams.get(device)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _side_streams[device] = s
    return s


def _background_plan(n_tiles, kicks):
    """tile boundaries of the kicks + 1 pieces"""
    w0, wm, wl = background_weights
    w = [w0] + [wm] * max(kicks - 1, 0) + ([wl] if kicks > 0 else [])
    tot, acc, cuts = sum(w), 0.0, [0]
    for x in w[:-1]:
        acc += x
        cuts.append(min(n_tiles, int(round(n_tiles * acc / tot))))
    return cuts + [n_tiles]


def _background_slot(c):
    """One launch opportunity: the queue's head goes out on the side stream, behind everything the main stream has been
    given so far.  The closures still queued are shared out evenly over the opportunities left."""
    left = max(c.slots, 1)
    c.slots = max(c.slots - 1, 0)
    if not c.queue:
        return
    n = max(1, min(len(c.queue), -(-len(c.queue) // left)))
    main = torch.cuda.current_stream()
    side = _side_stream(main.device)
    side.wait_stream(main)
    c.side_launched = main.device
    with torch.cuda.stream(side):
        for _ in range(n):
            c.queue.pop(0)()


def _background_kick(c):
    if c is None or not c.counting:
        return
    c.kicks += 1
    _background_slot(c)


def _background_drain(c):
    """everything queued goes out on the side stream now"""
    if not c.queue:
        return
    main = torch.cuda.current_stream()
    side = _side_stream(main.device)
    side.wait_stream(main)
    c.side_launched = main.device
    with torch.cuda.stream(side):
        while c.queue:
            c.queue.pop(0)()


def join_side_work(c):
    """The current stream waits for everything context `c` issued on the side stream; the gradients that work produced are
    then announced (grad-ready callback), each parameter once and only after EVERY piece has been waited for.  Runs at the
    end of every backward pass that used the side stream (autograd engine callback); optimizers and reducers call it too --
    it is a no-op when nothing is pending."""
    if c is None:
        return
    if c.counting:                  # the pass is over: its number of attention launches is the plan of the next one
        c.counting, c.kicks_expected = False, c.kicks
    _background_drain(c)            # fewer attention launches than planned: what is left of the sweeps goes out now
    c.slots = 0
    done, c.pending[:] = list(c.pending), []
    cur = torch.cuda.current_stream() if done else None
    for ev, _, stream in done:
        stream.wait_event(ev)               # the stream backward ran on (this may be the engine's thread, with another current stream)
        cur.wait_event(ev)
    if c.side_launched is not None:
        if not done:                        # (side-stream work without a closing event: wait for the stream itself)
            torch.cuda.current_stream(c.side_launched).wait_stream(_side_stream(c.side_launched))
        c.side_launched = None
    seen, params = set(), []
    for _, ps, _ in done:
        for p in ps:
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
    _ready(*params)


def _dw_pieces(c, h, wt, b, labels_i32, rowscal, V, kernel, bias, cuts):
    """closures for the vocabulary tiles cuts[i] .. cuts[i + 1]; the last one adds the label term and records the event that
    completes the projection's gradient"""
    dW, db = kernel.grad, bias.grad
    side = _side_stream(h.device)
    main = torch.cuda.current_stream(h.device)
    for t in (h, rowscal, labels_i32, wt, b, dW, db):
        if t is not None:
            t.record_stream(side)
    spans = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]

    def piece(lo, hi, last):
        def run():
            vocab_ce_dw_sweep(h, wt, b, rowscal, V, dW, db, lo, hi, background_wgs(h.device))
            if last:
                vocab_ce_dw_labels(h, labels_i32, rowscal, V, dW, db)
                c.pending.append((torch.cuda.current_stream().record_event(), (kernel, bias), main))
        return run
    return [piece(lo, hi, i == len(spans) - 1) for i, (lo, hi) in enumerate(spans)]


def _queue_background_dw(c, h, wt, b, labels_i32, rowscal, V, kernel, bias):
    """(inside a backward pass) the dW sweep of the vocabulary head as kicks + 1 background pieces: one now, one behind
    every attention backward launch of this pass"""
    if c.counting or c.queue:
        join_side_work(c)               # a second head on this arena in the same backward pass: finish the first one's sweep first
    cuts = _background_plan((V + 127) // 128, c.kicks_expected)
    c.queue.extend(_dw_pieces(c, h, wt, b, labels_i32, rowscal, V, kernel, bias, cuts))
    c.slots = len(c.queue)
    c.counting, c.kicks = True, 0
    _background_slot(c)
    torch.autograd.Variable._execution_engine.queue_callback(lambda: join_side_work(c))


class VocabCEFn(torch.autograd.Function):
    """R12 + R13 + R14 for training without the (R x V) logits: vocabulary projection, softmax, masked sparse
    CE (mean over valid rows) and their backward, logits recomputed in MFMA accumulators (csrc/vocab_ce.hip).
    apply(h, pack, labels_i32, V, variant, unit_grad, kernel, bias)"""

    @staticmethod
    def forward(ctx, h, pack, labels_i32, V, variant, unit_grad, kernel, bias, poison=None):
        h = h.contiguous()
        wt, _, b = pack.get(h.dtype, h.shape[1], False)
        scale = label_scale(labels_i32, V)                    # [1 / n_valid, n_valid]
        item, dh, rowscal = vocab_ce_fwd(h, wt, b, labels_i32, scale, V, variant)
        ctx.save_for_backward(h, dh, rowscal, labels_i32)
        ctx.pack, ctx.V, ctx.unit_grad = pack, V, unit_grad
        ctx.params = (kernel, bias)
        return sum_scaled(item, scale, poison)

    @staticmethod
    def backward(ctx, g):
        h, dh, rowscal, labels_i32 = ctx.saved_tensors
        kernel, bias = ctx.params
        if not ctx.unit_grad:       # the kernels ran at scale 1/valid in forward: fold the upstream gradient in now
            gf = g.to(torch.float32).reshape(1).contiguous()
            dh_g, rowscal_g = torch.empty_like(dh), torch.empty_like(rowscal)
            L.check(L.lib().b4c_vocab_ce_apply_grad(_p(dh), dh.stride(0), _p(rowscal), _p(gf), _p(dh_g), dh_g.stride(0),
                                                    _p(rowscal_g), dh.shape[0], dh.shape[1], _st()), 'vocab_ce_apply_grad')
            dh, rowscal = dh_g, rowscal_g
        wt, _, b = ctx.pack.get(h.dtype, h.shape[1], False)
        off = getattr(ctx.pack, 'tied_offset', None)
        if off is not None:     # tied head: `kernel` is the (rows, K) embedding table; dW [K, V] is added transposed
            dWt = torch.zeros(h.shape[1], ctx.V, dtype=torch.float32, device=h.device)
            inplace = _inplace_ok(kernel, bias)
            db = bias.grad if inplace else torch.zeros(bias.shape, dtype=torch.float32, device=h.device)
            vocab_ce_dw(h, wt, b, labels_i32, rowscal, ctx.V, dWt, db)
            dtab = kernel.grad if inplace else torch.zeros(kernel.shape, dtype=torch.float32, device=h.device)
            transpose_add_(dtab[off:off + ctx.V], dWt)
            if inplace:
                _ready(bias)        # the table is announced by the embedding backward, which runs last
                return dh, None, None, None, None, None, None, None, None
            return dh, None, None, None, None, None, dtab, db, None
        actx = arena_context(kernel, bias)
        if actx is not None:
            if h.is_cuda and _background_dw_for(actx):
                _queue_background_dw(actx, h, wt, b, labels_i32, rowscal, ctx.V, kernel, bias)
            else:
                vocab_ce_dw(h, wt, b, labels_i32, rowscal, ctx.V, kernel.grad, bias.grad)
                _ready(kernel, bias)
            dW = db = None
        else:
            dW = torch.zeros(kernel.shape, dtype=torch.float32, device=h.device)
            db = torch.zeros(bias.shape, dtype=torch.float32, device=h.device)
            vocab_ce_dw(h, wt, b, labels_i32, rowscal, ctx.V, dW, db)
        return dh, None, None, None, None, None, dW, db, None


class GatherRowsFn(torch.autograd.Function):
    """R11: rows of the encoder output at the given flat indices (idx < 0 -> zero row)."""

    @staticmethod
    def forward(ctx, src2d, idx, n_out):
        ctx.save_for_backward(idx)
        ctx.n_src = src2d.shape[0]
        return gather_rows(src2d, idx, n_out)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return scatter_rows(dout.contiguous(), idx, ctx.n_src), None, None


class FusedSoftmaxCEFn(torch.autograd.Function):
    """R12 tail + R13 + R14 fused for training: softmax over V, masked sparse CE (mean over valid
    rows), and d loss / d logits written in place of the logits (which are consumed)."""

    @staticmethod
    def forward(ctx, logits, labels_i32, V, variant, unit_grad, poison=None):
        scale = label_scale(labels_i32, V)                    # [1 / n_valid, n_valid]
        with _timed('softmax_ce'):
            item = softmax_ce_fwd_bwd_(logits, labels_i32, scale, V, variant)
        ctx.save_for_backward(logits)
        ctx.unit_grad = unit_grad
        return sum_scaled(item, scale, poison)

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        if not ctx.unit_grad:
            dlogits = dlogits * g.to(dlogits.dtype)
        return dlogits, None, None, None, None, None


# --------------------------------------------------------------------------------------
# round 2: the reference's own composition loss(y, model(x)) is differentiable; other heads; MHA in general
# --------------------------------------------------------------------------------------
def dropout(x, rate, seed):
    """keep(seed, e) ? x / (1 - rate) : 0 over the flat element index (its own backward)."""
    _cuda(x)
    x = x.contiguous()
    if x.numel() % 8:
        raise B4CError('dropout: element count %d must be a multiple of 8' % x.numel())
    y = torch.empty_like(x)
    L.check(L.lib().b4c_dropout(_p(x), _p(y), x.numel(), rate, seed, dt_code(x.dtype), _st()), 'dropout')
    return y


class DropoutFn(torch.autograd.Function):
    """Encoder.call's input dropout (transformer.py:263) for an Encoder used on its own."""

    @staticmethod
    def forward(ctx, x, rate, seed):
        ctx.rate, ctx.seed = rate, seed
        return dropout(x, rate, seed)

    @staticmethod
    def backward(ctx, g):
        return dropout(g, ctx.rate, ctx.seed), None, None


def softmax_rows_bwd(probs, g, V):
    R, ld = probs.shape[0], probs.stride(0)
    dx = torch.empty(R, ld, dtype=probs.dtype, device=probs.device)
    if ld != probs.shape[1]:
        dx = dx[:, :probs.shape[1]]
    if R == 0:
        return dx
    L.check(L.lib().b4c_softmax_rows_bwd(_p(probs), ld, _p(g), g.stride(0), _p(dx), ld, R, V, dt_code(probs.dtype), _st()),
            'softmax_rows_bwd')
    return dx


class SoftmaxRowsFn(torch.autograd.Function):
    """Dense(V, softmax)'s activation on materialised logits [R, ld] (head.py:36) with its Jacobian."""

    @staticmethod
    def forward(ctx, logits, V):
        probs = softmax_rows(logits, V)
        ctx.save_for_backward(probs)
        ctx.V = V
        return probs

    @staticmethod
    def backward(ctx, g):
        (probs,) = ctx.saved_tensors
        g = g.to(probs.dtype)
        if g.stride(1) != 1 or g.stride(0) % 8:
            g = g.contiguous()
        return softmax_rows_bwd(probs, g, ctx.V), None


class MaskedSparseCEFn(torch.autograd.Function):
    """MaskedLoss(sparse_categorical_crossentropy)(y_true, y_pred) on PROBABILITIES (losses.py:31-98, main.py:89):
    mean over non-pad rows of the TF-backend sparse CE; differentiable w.r.t. the probabilities.
    apply(probs [R, >=V] with 8-aligned row pitch, labels fp32 [R] (-1 = pad), V, variant)"""

    @staticmethod
    def forward(ctx, probs, labels_f32, V, variant):
        item, nval = sparse_ce_from_probs(probs, labels_f32, V, variant)
        ctx.save_for_backward(probs, labels_f32, nval)
        ctx.V, ctx.variant = V, variant
        return item.sum() / nval[0]

    @staticmethod
    def backward(ctx, g):
        probs, labels_f32, nval = ctx.saved_tensors
        R, ld = probs.shape[0], probs.stride(0)
        gscale = (g.to(torch.float32) / nval[0]).reshape(1).contiguous()
        dp = torch.empty(R, ld, dtype=probs.dtype, device=probs.device)
        if R:
            L.check(L.lib().b4c_sparse_ce_from_probs_bwd(_p(probs), ld, _p(labels_f32), _p(gscale), _p(dp), ld, R, ctx.V,
                                                         ctx.variant, dt_code(probs.dtype), _st()), 'sparse_ce_from_probs_bwd')
        return dp[:, :probs.shape[1]], None, None, None


def sigmoid(x):
    _cuda(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(L.lib().b4c_sigmoid_fwd(_p(x), _p(y), x.numel(), dt_code(x.dtype), _st()), 'sigmoid_fwd')
    return y


class SigmoidFn(torch.autograd.Function):
    """Dense(activation='sigmoid') of BinaryClassificationHead / MultiLabel_MultiClass_classification (head.py:12,59)."""

    @staticmethod
    def forward(ctx, x):
        y = sigmoid(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.to(y.dtype).contiguous()
        dx = torch.empty_like(y)
        L.check(L.lib().b4c_sigmoid_bwd(_p(y), _p(g), _p(dx), y.numel(), dt_code(y.dtype), _st()), 'sigmoid_bwd')
        return dx


def masked_bce(probs, labels_f32, pos_weight=None, want_grad=False):
    """-> (item_loss fp32 [n], sums fp32 [2] = (sum of weighted losses, non-pad count), dprobs fp32 [n] or None)"""
    _cuda(probs)
    n = probs.numel()
    item = torch.empty(n, dtype=torch.float32, device=probs.device)
    sums = torch.zeros(2, dtype=torch.float32, device=probs.device)
    dp = torch.empty(n, dtype=torch.float32, device=probs.device) if want_grad else None
    L.check(L.lib().b4c_masked_bce(_p(probs), _p(labels_f32), float(pos_weight) if pos_weight is not None else 0.0, _p(item),
                                   _p(sums), _p(dp), n, dt_code(probs.dtype), _st()), 'masked_bce')
    return item, sums, dp


class MaskedBCEFn(torch.autograd.Function):
    """MaskedLoss(binary_crossentropy, pos_weight)(y_true, y_pred) (losses.py:31-98): weighted mean over non-pad items,
    divided by (pos_weight + 1) / 2 when a pos_weight is given.  apply(probs (any shape), labels fp32 same count, pos_weight)"""

    @staticmethod
    def forward(ctx, probs, labels_f32, pos_weight):
        p = probs.contiguous()
        _, sums, dp = masked_bce(p, labels_f32, pos_weight, want_grad=True)
        norm = (float(pos_weight) + 1.0) / 2.0 if pos_weight is not None else 1.0
        ctx.save_for_backward(dp, sums)
        ctx.norm, ctx.shape, ctx.dtype = norm, probs.shape, probs.dtype
        return sums[0] / sums[1] / norm

    @staticmethod
    def backward(ctx, g):
        dp, sums = ctx.saved_tensors
        return (dp * (g.to(torch.float32) / sums[1] / ctx.norm)).to(ctx.dtype).view(ctx.shape), None, None


def binary_counts(y_true_f32, y_pred):
    _cuda(y_pred)
    out = torch.zeros(6, dtype=torch.float32, device=y_pred.device)
    yp = y_pred.contiguous()
    L.check(L.lib().b4c_binary_counts(_p(y_true_f32), _p(yp), _p(out), yp.numel(), dt_code(yp.dtype), _st()), 'binary_counts')
    return out


def compact_labels(labels_padded, counts, offsets, cap, flat_idx=None):
    """(B, M) fp32 labels padded with -1 -> int32 [cap] in mask order; entries >= R are -1 (and flat_idx[>= R] = -1)."""
    _cuda(labels_padded)
    B, M = labels_padded.shape
    lab = labels_padded.to(torch.float32).contiguous()
    out = torch.empty(cap, dtype=torch.int32, device=lab.device)
    if M == 0:          # a batch whose label rows are all empty (padded_batch of zero-length arrays, input_pipeline.py:198-214)
        out.fill_(-1)
        if flat_idx is not None:
            flat_idx.fill_(-1)
        return out
    L.check(L.lib().b4c_compact_labels(_p(lab), B, M, _p(counts), _p(offsets), _p(out), _p(flat_idx), cap, _st()),
            'compact_labels')
    return out


def attn_weights(qkv, key_pad, lse, B, S, H, dh):
    w = torch.empty(B, H, S, S, dtype=torch.float32, device=qkv.device)
    L.check(L.lib().b4c_attn_weights(_p(qkv), qkv.stride(0), _p(key_pad), _p(lse), _p(w), B, S, H, dh, dt_code(qkv.dtype),
                                     _st()), 'attn_weights')
    return w


def transpose_add_(dst, src):
    """dst [N, K] += src [K, N]^T (fp32)."""
    K, N = src.shape
    L.check(L.lib().b4c_transpose_add(_p(src), src.stride(0), _p(dst), dst.stride(0), K, N, _st()), 'transpose_add')


def rows_gather_f32(src, idx):
    out = torch.empty(idx.shape[0], src.shape[1], dtype=torch.float32, device=src.device)
    L.check(L.lib().b4c_rows_gather_f32(_p(src), src.stride(0), _p(idx), _p(out), out.stride(0), idx.shape[0], src.shape[1],
                                        _st()), 'rows_gather_f32')
    return out


def rows_scatter_add_f32_(dst, idx, src):
    L.check(L.lib().b4c_rows_scatter_add_f32(_p(src), src.stride(0), _p(idx), _p(dst), dst.stride(0), idx.shape[0],
                                             src.shape[1], _st()), 'rows_scatter_add_f32')


class MHAFn(torch.autograd.Function):
    """MultiHeadAttention.call(v, k, q, mask) in general (transformer.py:137-160): distinct value / key / query inputs
    of one padded length S, key-side padding mask, attention weights on request.  Differentiable in the three inputs
    and all eight parameters.  apply(xq, xk, xv, key_pad, wq, bq, wk, bk, wv, bv, wo, bo, pk_qkv, pk_o, B, S, H, same, want_w)"""

    @staticmethod
    def forward(ctx, xq, xk, xv, key_pad, wq, bq, wk, bk, wv, bv, wo, bo, pk_qkv, pk_o, B, S, H, same, want_w):
        T_tok, d = xq.shape
        dh = d // H
        wt_qkv, _, b_qkv = pk_qkv.get(xq.dtype, d, True)
        wt_o, _, b_o = pk_o.get(xq.dtype, d, True)
        if same:
            qkv = gemm_nt(xq, wt_qkv, 3 * d, b_qkv)
        else:
            qkv = torch.empty(T_tok, 3 * d, dtype=xq.dtype, device=xq.device)
            for i, x in enumerate((xq, xk, xv)):
                gemm_nt(x, wt_qkv[i * d:(i + 1) * d], d, b_qkv[i * d:(i + 1) * d], out=qkv[:, i * d:(i + 1) * d])
        o, lse = attn_fwd(qkv, key_pad, B, S, H, dh)
        out = gemm_nt(o, wt_o, d, b_o)
        w = attn_weights(qkv, key_pad, lse, B, S, H, dh) if want_w else None
        ctx.save_for_backward(xq, xk, xv, key_pad, qkv, o, lse)
        ctx.pk, ctx.dims, ctx.same = (pk_qkv, pk_o), (B, S, H, dh), same
        if w is None:
            w = torch.empty(0, device=xq.device)
        ctx.mark_non_differentiable(w)
        return out, w

    @staticmethod
    def backward(ctx, dout, _):
        xq, xk, xv, key_pad, qkv, o, lse = ctx.saved_tensors
        pk_qkv, pk_o = ctx.pk
        B, S, H, dh = ctx.dims
        d = H * dh
        _, wc_o, _ = pk_o.get(xq.dtype, d, True)
        _, wc_qkv, _ = pk_qkv.get(xq.dtype, d, True)
        dy = dout.contiguous()
        dWo, dbo = gemm_tn(o, dy, d, d)
        d_o = gemm_nt(dy, wc_o, d)
        dqkv = attn_bwd(qkv, key_pad, o, d_o, lse, B, S, H, dh)
        if ctx.same:
            dW, db = gemm_tn(xq, dqkv, d, 3 * d)
            (gq, gk, gv), (gbq, gbk, gbv) = pk_qkv.split_grads(dW, db)
            dx = gemm_nt(dqkv, wc_qkv, d)
            return (dx, None, None, None, gq, gbq, gk, gbk, gv, gbv, dWo, dbo) + (None,) * 7
        gw, gb, gx = [], [], []
        for i, x in enumerate((xq, xk, xv)):
            g = dqkv[:, i * d:(i + 1) * d]
            dW, db = gemm_tn(x, g, d, d)
            gw.append(dW)
            gb.append(db)
            gx.append(gemm_nt(g, wc_qkv[:, i * d:(i + 1) * d], d))
        return (gx[0], gx[1], gx[2], None, gw[0], gb[0], gw[1], gb[1], gw[2], gb[2], dWo, dbo) + (None,) * 7


class TiedPackedLinear(PackedLinear):
    """Compute copies of a projection whose weights are rows off .. off+V of an embedding table (rows, K) fp32,
    vocabulary-major: ``wt`` [Vp][Kp] is a straight copy of those rows, ``wc`` [Kp][Vp] their transpose, ``bias``
    the head's own fp32 [V].  Re-packed with the dense layers (same batched launch: the table slice is handed to
    b4c_pack_weights_batched as a "[V][K] kernel" with the two outputs swapped)."""

    def __init__(self, table, off, V, bias):
        import weakref
        self.table, self.tied_offset, self.bias_p = table, int(off), bias
        self.kernels, self.biases = [table], [bias]
        self.K = int(table.shape[1])
        self.Ns = [int(V)]
        self.N = int(V)
        self.Np = rup8(self.N)
        self._cache = {}
        _pack_registry.append(weakref.ref(self))

    def _descs(self, ent, Kp):
        t = self.table.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise B4CError('embedding tables must be contiguous float32')
        lz = getattr(self.table, '_b4c_lazy', None)
        if lz is not None:          # row-lazy optimizer: the whole table is about to be read
            lz.sync()
        rows = t[self.tied_offset:self.tied_offset + self.N]
        ent['bias'][:self.N].copy_(self.bias_p.detach())
        # as a Keras kernel [K' = V][N' = K]:  wt'[n'][k'] = src[k'][n'] is our wc [K][V],  wc'[k'][n'] is our wt [V][K]
        return [L.PackDesc(rows.data_ptr(), None, ent['wc'].data_ptr() if ent['wc'] is not None else None,
                           ent['wt'].data_ptr(), None, self.N, self.K, self.Np, Kp, 0, 0)]

    def split_grads(self, dW, db):
        return [dW], [db]


class TiedLogitsFn(torch.autograd.Function):
    """logits [R, Vp] = h . E[off : off + V]^T + bias for the tied-weight head (materialised route)."""

    @staticmethod
    def forward(ctx, h, table, bias, pack, out_fp32):
        h = h.contiguous()
        wt, _, b = pack.get(h.dtype, h.shape[1], True)
        ctx.save_for_backward(h)
        ctx.pack, ctx.params = pack, (table, bias)
        odt = torch.float32 if out_fp32 else h.dtype
        return gemm_nt(h, wt, pack.Np, b, out_dtype=odt, out=empty_rows(h.shape[0], pack.Np, odt, h.device))

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        pack = ctx.pack
        table, bias = ctx.params
        g = _rows_ok(g, h.dtype)
        _, wc, _ = pack.get(h.dtype, h.shape[1], True)
        dWt, db = gemm_tn(h, g, pack.K, pack.N)                 # [K, V] fp32, [V]
        dh = gemm_nt(g, wc, h.shape[1])
        off = pack.tied_offset
        if _inplace_ok(table, bias):
            transpose_add_(table.grad[off:off + pack.N], dWt)
            bias.grad += db
            _ready(bias)
            return dh, None, None, None, None
        dtab = torch.zeros(table.shape, dtype=torch.float32, device=h.device)
        transpose_add_(dtab[off:off + pack.N], dWt)
        return dh, dtab, db, None, None


# --------------------------------------------------------------------------------------
# sampled-softmax head (config 5; no reference counterpart)
# --------------------------------------------------------------------------------------
def log_uniform_sample(seed, n, range_max, device):
    """n ids (int64) from the log-uniform sampler over [0, range_max) with replacement, and log Q(id) (fp32)."""
    ids = torch.empty(n, dtype=torch.int64, device=device)
    logq = torch.empty(n, dtype=torch.float32, device=device)
    L.check(L.lib().b4c_log_uniform_sample(int(seed) & 0xFFFFFFFFFFFFFFFF, n, range_max, _p(ids), _p(logq), _st()),
            'log_uniform_sample')
    return ids, logq


def row_dot(a, b):
    out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    L.check(L.lib().b4c_row_dot(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), a.shape[0], a.shape[1], dt_code(a.dtype),
                                _st()), 'row_dot')
    return out


def scatter_add_1d_(dst, idx, src):
    L.check(L.lib().b4c_scatter_add_1d(_p(src), _p(idx), _p(dst), idx.shape[0], _st()), 'scatter_add_1d')


def row_scale_f32(src, scale):
    out = torch.empty(src.shape[0], src.shape[1], dtype=torch.float32, device=src.device)
    L.check(L.lib().b4c_row_scale_f32(_p(src), src.stride(0), _p(scale), _p(out), out.stride(0), src.shape[0], src.shape[1],
                                      dt_code(src.dtype), _st()), 'row_scale_f32')
    return out


class SampledCEFn(torch.autograd.Function):
    """Sampled-softmax masked-item loss (tf.nn.sampled_softmax_loss semantics: shared log-uniform negatives, logQ
    correction, accidental hits removed), mean over valid rows.  The vocabulary-major projection table (V, K) is only
    touched at the sampled and the label rows: its gradient is row-sparse.
    apply(h [R, K], table (V, K) fp32, bias (V,) fp32, labels_i32 [R], samples int64 [Ns], logq fp32 [Ns], unit_grad)"""

    @staticmethod
    def forward(ctx, h, table, bias, labels_i32, samples, logq, unit_grad):
        h = h.contiguous()
        V, Kd = table.shape
        Ns = samples.shape[0]
        valid_m = (labels_i32 >= 0) & (labels_i32 < V)
        valid = valid_m.sum().to(torch.float32)
        scale = torch.where(valid > 0, 1.0 / valid.clamp(min=1.0), torch.zeros_like(valid)).reshape(1)
        tab = table.detach()
        idx_y = torch.where(valid_m, labels_i32, torch.full_like(labels_i32, -1)).to(torch.int64)
        lz = getattr(table, '_b4c_lazy', None)
        if lz is not None:          # row-lazy optimizer: the sampled and the label rows are brought up to date before they are read
            lz.catch_up(samples, note=ctx.needs_input_grad[1])
            lz.catch_up(idx_y, note=ctx.needs_input_grad[1])
        Ws = rows_gather_f32(tab, samples).to(h.dtype)                               # [Ns, K]
        bs = (bias.detach()[samples] - logq).contiguous()
        Z = gemm_nt(h, Ws, Ns, bs)                                                   # negatives' logits, bias - logQ folded in
        Wy = rows_gather_f32(tab, idx_y).to(h.dtype)                                 # [R, K], zero rows where ignored
        ztrue = row_dot(h, Wy) + bias.detach()[idx_y.clamp(min=0)]
        item = torch.empty(h.shape[0], dtype=torch.float32, device=h.device)
        dtrue = torch.empty(h.shape[0], dtype=torch.float32, device=h.device)
        if h.shape[0]:
            L.check(L.lib().b4c_sampled_ce_fwd_bwd(_p(Z), Z.stride(0), _p(ztrue), _p(samples), _p(labels_i32), V, _p(item),
                                                   _p(dtrue), _p(scale), h.shape[0], Ns, dt_code(Z.dtype), _st()),
                    'sampled_ce_fwd_bwd')
        ctx.save_for_backward(h, Z, dtrue, Ws, Wy, samples, idx_y)
        ctx.params, ctx.unit_grad = (table, bias), unit_grad
        return item.sum() * scale[0]

    @staticmethod
    def backward(ctx, g):
        h, dZ, dtrue, Ws, Wy, samples, idx_y = ctx.saved_tensors
        table, bias = ctx.params
        Kd = h.shape[1]
        if not ctx.unit_grad:
            dZ = dZ * g.to(dZ.dtype)
            dtrue = dtrue * g.to(torch.float32)
        dh = gemm_nt(dZ, Ws.t().contiguous(), Kd)
        dh = torch.addcmul(dh, dtrue.to(dh.dtype)[:, None], Wy)
        dWT, dbs = gemm_tn(h, dZ, Kd, dZ.shape[1])                  # [K, Ns] = h^T dZ, [Ns] = column sums of dZ
        dWs = torch.zeros(dZ.shape[1], Kd, dtype=torch.float32, device=h.device)
        transpose_add_(dWs, dWT)
        dWy = row_scale_f32(h, dtrue)
        inplace = _inplace_ok(table, bias)
        tg = table.grad if inplace else torch.zeros(table.shape, dtype=torch.float32, device=h.device)
        bg = bias.grad if inplace else torch.zeros(bias.shape, dtype=torch.float32, device=h.device)
        rows_scatter_add_f32_(tg, samples, dWs)
        rows_scatter_add_f32_(tg, idx_y, dWy)
        scatter_add_1d_(bg, samples, dbs.contiguous())
        scatter_add_1d_(bg, idx_y, dtrue.contiguous())
        if inplace:
            _ready(table, bias)
            return dh, None, None, None, None, None, None
        return dh, tg, bg, None, None, None, None
