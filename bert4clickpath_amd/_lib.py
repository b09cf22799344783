"""ctypes binding of libb4c_hip.so (include/b4c.h).  There is no CPU fallback: if the
library is missing, or a call fails, this raises -- the product path is the HIP path."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('B4C_LIB_PATH') or os.path.join(_HERE, 'libb4c_hip.so')     # override: A/B of two builds (scratch)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'b4c.h')

ABI_VERSION = 12      # include/b4c.h; b4c_abi_version() of the library must agree
F32, BF16 = 0, 1
ACT_NONE, ACT_RELU = 0, 1
CE_TF, CE_PLAIN = 0, 1
MAX_FEATURES, MAX_TOPK = 4, 16


class B4CError(RuntimeError):
    pass


class PackDesc(ctypes.Structure):
    """b4c_pack_desc of include/b4c.h."""
    _fields_ = [('src', ctypes.c_void_p), ('bias_src', ctypes.c_void_p), ('wt', ctypes.c_void_p), ('wc', ctypes.c_void_p),
                ('bias_dst', ctypes.c_void_p), ('K', ctypes.c_int32), ('N', ctypes.c_int32), ('ld_t', ctypes.c_int32),
                ('ld_c', ctypes.c_int32), ('col_off', ctypes.c_int32), ('_pad', ctypes.c_int32)]


class TNDesc(ctypes.Structure):
    """b4c_tn_desc of include/b4c.h."""
    _fields_ = [('A', ctypes.c_void_p), ('G', ctypes.c_void_p), ('dW', ctypes.c_void_p * 4), ('db', ctypes.c_void_p * 4),
                ('lda', ctypes.c_int32), ('ldg', ctypes.c_int32), ('K', ctypes.c_int32), ('n_seg', ctypes.c_int32),
                ('seg_width', ctypes.c_int32), ('ldw', ctypes.c_int32)]


def declared_symbols(header_path=HEADER_PATH):
    """Names of every function include/b4c.h declares (used by the symbol-export test)."""
    with open(header_path) as f:
        src = f.read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(b4c_[a-z0-9_]+)\s*\(', src)))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise B4CError(
                'libb4c_hip.so not found at %s: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C bert4clickpath_amd/csrc`).  There is no CPU fallback.' % LIB_PATH)
        # PyTorch's HIP runtime must have seen the device before this library (and the system HIP runtime it links)
        # comes into the process: loaded the other way round -- e.g. build() and then smoke() in ONE process -- this
        # library's launches fail with "no ROCm-capable device is detected".  (No GPU: nothing to initialise.)
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        c = ctypes
        vp, i32, i64, f32, u64 = c.c_void_p, c.c_int, c.c_int64, c.c_float, c.c_uint64
        pp = c.POINTER(c.c_void_p)
        sig = {
            'b4c_abi_version': (i32, []),
            'b4c_last_error': (c.c_char_p, []),
            'b4c_keep': (i32, [u64, u64, f32]),
            'b4c_embed_concat_pe_fwd': (i32, [i32, pp, pp, c.POINTER(i32), c.POINTER(i64), vp, f32, vp, i32, vp, i32, i32, i32, f32, u64, i32, vp]),
            'b4c_embed_concat_pe_bwd': (i32, [i32, pp, pp, c.POINTER(i32), c.POINTER(i64), f32, vp, i32, i32, i32, i32, f32, u64, i32, vp]),
            'b4c_embed_concat_pe_bwd_sorted': (i32, [i32, pp, pp, pp, c.POINTER(i32), c.POINTER(i64), f32, vp, i32, i32, i32, i32, f32, u64, i32, vp]),
            'b4c_pack_weight': (i32, [vp, i32, i32, vp, i32, i32, i32, vp]),
            'b4c_gemm_nt': (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp]),
            'b4c_gemm_nt_add_ln': (i32, [vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, u64, i32, vp]),
            'b4c_gemm_dxdw_workspace_bytes': (i64, [i64, i32]),
            'b4c_gemm_dxdw': (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, pp, pp, i32, i64, vp, i64, vp]),
            'b4c_ffn_bwd_workspace_bytes': (i64, [i64]),
            'b4c_ffn_bwd': (i32, [vp, vp, vp, vp, f32, u64, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, i64, vp, i64, vp]),
            'b4c_ffn_fwd': (i32, [vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, i32, i32, vp, i32, vp, vp, vp, i64, f32, f32, u64, vp]),
            'b4c_attn_out_bwd_workspace_bytes': (i64, [i64]),
            'b4c_attn_out_bwd': (i32, [vp, vp, vp, vp, f32, u64, vp, i32, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, i64, vp, i64, vp]),
            'b4c_gemm_tn_group_workspace_bytes': (i64, [vp, i32, i32]),
            'b4c_gemm_tn_group': (i32, [vp, i32, i32, i32, vp, i64, vp]),
            'b4c_gemm_tn_workspace_bytes': (i64, [i32, i32, i32, i32]),
            'b4c_gemm_tn': (i32, [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i64, vp]),
            'b4c_gemm_tn_seg': (i32, [vp, i32, vp, i32, i32, pp, pp, i32, i32, i32, i32, vp, i64, vp]),
            'b4c_pack_weights_batched': (i32, [vp, i32, i32, i32, vp]),
            'b4c_attn_fwd': (i32, [vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
            'b4c_attn_bwd': (i32, [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
            'b4c_attn_bwd_workspace_bytes': (i64, [i32, i32, i32, i32, i32]),
            'b4c_attn_bwd_ws': (i32, [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, i32, vp]),
            'b4c_add_dropout_layernorm_fwd': (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, u64, i32, vp]),
            'b4c_add_dropout_layernorm_bwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, u64, i32, vp]),
            'b4c_add_dropout_layernorm_bwd_workspace_bytes': (i64, [i64, i32]),
            'b4c_add_dropout_layernorm_bwd_ws': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, u64, vp, i64, i32, vp]),
            'b4c_embed_concat_pe_bwd_sorted_workspace_bytes': (i64, [i32, c.POINTER(i32), i32, i32]),
            'b4c_embed_concat_pe_bwd_sorted_ws': (i32, [i32, pp, pp, pp, c.POINTER(i32), c.POINTER(i64), f32, vp, i32, i32, i32, i32, f32, u64, vp, i64, i32, vp]),
            'b4c_vocab_rank_workspace_bytes': (i64, [i64, i32, i32]),
            'b4c_vocab_rank': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i64, i64, i32, i32, vp]),
            'b4c_rank_metrics': (i32, [vp, i64, i32, vp, vp, vp]),
            'b4c_vocab_topk': (i32, [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp]),
            'b4c_poison_rows': (i32, [vp, i32, i64, i32, vp, i32, vp]),
            'b4c_mask_positions': (i32, [vp, i32, i32, i64, vp, vp, vp, i32, vp, vp, vp]),
            'b4c_padded_index': (i32, [vp, vp, vp, i32, i32, vp, vp]),
            'b4c_gather_rows': (i32, [vp, i32, vp, vp, i32, i64, i32, i32, vp]),
            'b4c_scatter_rows': (i32, [vp, i32, vp, vp, i32, i64, i64, i32, i32, vp]),
            'b4c_softmax_rows': (i32, [vp, i32, vp, i32, i64, i32, i32, vp]),
            'b4c_sparse_ce_from_probs': (i32, [vp, i32, vp, vp, vp, i64, i32, i32, i32, vp]),
            'b4c_softmax_ce_fwd_bwd': (i32, [vp, i32, vp, vp, vp, i64, i32, i32, i32, vp]),
            'b4c_vocab_ce_workspace_bytes': (i64, [i64, i32, i32]),
            'b4c_vocab_ce_fwd': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i64, i64, i32, i32, i32, vp]),
            'b4c_vocab_ce_dw': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, i64, i64, i32, i32, i32, vp]),
            'b4c_vocab_ce_dw_sweep': (i32, [vp, i32, vp, i32, vp, vp, vp, i32, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
            'b4c_vocab_ce_dw_labels': (i32, [vp, i32, vp, vp, vp, i32, vp, vp, i64, i64, i32, i32, i32, vp]),
            'b4c_sort_ids_workspace_bytes': (i64, [i64, i32]),
            'b4c_sort_ids': (i32, [vp, i64, i32, vp, vp, i64, vp]),
            'b4c_gather_i64': (i32, [vp, vp, vp, i64, vp]),
            'b4c_zero': (i32, [vp, i64, vp]),
            'b4c_chain_ids': (i32, [vp, vp, vp, i32, i32, i64, i64, vp, i32, vp]),
            'b4c_rows_add': (i32, [vp, i32, vp, vp, i32, i64, i32, i32, i32, vp]),
            'b4c_attn_mq_fwd': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
            'b4c_attn_mq_bwd': (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
            'b4c_label_scale': (i32, [vp, i64, i32, vp, vp]),
            'b4c_sum_scaled': (i32, [vp, i64, vp, vp, vp, vp]),
            'b4c_vocab_ce_apply_grad': (i32, [vp, i32, vp, vp, vp, i32, vp, i64, i32, vp]),
            'b4c_relu_gate': (i32, [vp, vp, vp, i64, i32, vp]),
            'b4c_vocab_lse': (i32, [vp, i32, vp, i32, vp, vp, vp, i64, i64, i32, i32, vp]),
            'b4c_gemm_nt_softmax': (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp]),
            'b4c_topk_rows': (i32, [vp, i32, i64, i32, i32, vp, vp, vp, vp, i32, vp]),
            'b4c_topk_rows_ws': (i32, [vp, i32, i64, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
            'b4c_adam_step': (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp]),
            'b4c_adam_rows': (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp, i32, f32, f32, f32, f32, i32, vp]),
            'b4c_dropout': (i32, [vp, vp, i64, f32, u64, i32, vp]),
            'b4c_softmax_rows_bwd': (i32, [vp, i32, vp, i32, vp, i32, i64, i32, i32, vp]),
            'b4c_sparse_ce_from_probs_bwd': (i32, [vp, i32, vp, vp, vp, i32, i64, i32, i32, i32, vp]),
            'b4c_sigmoid_fwd': (i32, [vp, vp, i64, i32, vp]),
            'b4c_sigmoid_bwd': (i32, [vp, vp, vp, i64, i32, vp]),
            'b4c_masked_bce': (i32, [vp, vp, f32, vp, vp, vp, i64, i32, vp]),
            'b4c_binary_counts': (i32, [vp, vp, vp, i64, i32, vp]),
            'b4c_compact_labels': (i32, [vp, i32, i32, vp, vp, vp, vp, i32, vp]),
            'b4c_attn_weights': (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
            'b4c_transpose_add': (i32, [vp, i32, vp, i32, i32, i32, vp]),
            'b4c_log_uniform_sample': (i32, [u64, i32, i64, vp, vp, vp]),
            'b4c_row_dot': (i32, [vp, i32, vp, i32, vp, i64, i32, i32, vp]),
            'b4c_sampled_ce_fwd_bwd': (i32, [vp, i32, vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, vp]),
            'b4c_scatter_add_1d': (i32, [vp, vp, vp, i64, vp]),
            'b4c_row_scale_f32': (i32, [vp, i32, vp, vp, i32, i64, i32, i32, vp]),
            'b4c_nonpad_positions': (i32, [vp, i32, i32, i64, vp, vp, vp, i32, vp, vp, vp]),
            'b4c_remap_index': (i32, [vp, vp, vp, i64, vp]),
            'b4c_embed_concat_pe_fwd_packed': (i32, [i32, pp, pp, c.POINTER(i32), c.POINTER(i64), vp, f32, vp, i32, vp, i32, i32, i32, f32, u64, vp, i64, i32, vp]),
            'b4c_attn_fwd_varlen': (i32, [vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
            'b4c_attn_bwd_varlen': (i32, [vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, i32, vp]),
            'b4c_rows_gather_f32': (i32, [vp, i32, vp, vp, i32, i64, i32, vp]),
            'b4c_rows_scatter_add_f32': (i32, [vp, i32, vp, vp, i32, i64, i32, vp]),
        }
        # The argument lists below belong to ONE ABI version: a library that exports the same names with older lists would
        # take a device pointer for a stream and fault on the GPU.  Compare before anything is bound.
        try:
            L.b4c_abi_version.restype = i32
            have = int(L.b4c_abi_version())
        except AttributeError:
            raise B4CError('%s does not export b4c_abi_version: not a libb4c_hip.so of this tree; rebuild it '
                           '(`make -C bert4clickpath_amd/csrc`)' % LIB_PATH) from None
        if have != ABI_VERSION:
            raise B4CError('%s is ABI %d, this binding expects ABI %d; rebuild it (`make -C bert4clickpath_amd/csrc`)'
                           % (LIB_PATH, have, ABI_VERSION))
        for name, (res, args) in sig.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                raise B4CError('%s (ABI %d) does not export %s; rebuild it (`make -C bert4clickpath_amd/csrc`)'
                               % (LIB_PATH, have, name)) from None
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=''):
    if rc != 0:
        raise B4CError('%s failed (rc=%d): %s' % (what or 'b4c call', rc, lib().b4c_last_error().decode()))
