"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/b4c.h declares, host
logic (vocabulary lookup, token chaining, Cloze masking, synthetic batches) agrees with the oracle, the
keep-mask hash agrees between host numpy and the library, and the product path refuses to run on CPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import numpy_ref as nr


def test_library_exports_every_declared_symbol():
    from bert4clickpath_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 21 and 'b4c_gemm_nt' in names and 'b4c_attn_bwd' in names
    L = ctypes.CDLL(_lib.LIB_PATH)          # loads without a GPU: no device call at load time
    for n in names:
        assert hasattr(L, n), 'libb4c_hip.so does not export %s' % n
    assert _lib.lib().b4c_abi_version() == _lib.ABI_VERSION == 12


def test_keep_mask_hash_host_vs_library():
    from bert4clickpath_amd import _lib, ops
    L = _lib.lib()
    for seed, rate in ((1, 0.1), (0xDEADBEEFCAFE, 0.5), (2 ** 63 + 5, 0.25)):
        host = ops.keep_mask(seed, 257, rate)
        libv = np.asarray([L.b4c_keep(seed, e, rate) for e in range(257)], dtype=bool)
        assert np.array_equal(host, libv)
        assert abs(host.mean() - (1 - rate)) < 0.12


def test_invalid_arguments_fail_loudly_without_a_gpu():
    from bert4clickpath_amd import _lib
    L = _lib.lib()
    rc = L.b4c_gemm_nt(None, 8, None, 8, None, 8, 4, 4, 8, None, 0, None, 0, None, 0, 1, 1, None)
    assert rc == -1 and b'gemm_nt' in L.b4c_last_error()
    rc = L.b4c_add_dropout_layernorm_fwd(1, 1, 1, 1, 1, 1, 1, 10, 12, 1e-6, 0.0, 0, 1, None)   # d % 8 != 0
    assert rc == -1 and b'multiple of 8' in L.b4c_last_error()
    rc = L.b4c_topk_rows(1, 8, 1, 8, 17, 1, None, None, None, 0, None)
    assert rc == -1


def test_no_cpu_fallback():
    from bert4clickpath_amd import ops
    from bert4clickpath_amd._lib import B4CError
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    with pytest.raises(B4CError):
        ops.topk_rows(torch.zeros(2, 8), 8, 1)
    model = ClickstreamTransformer({'items': ['asin']}, {'items': ['a', 'b']}, {'items': 16}, SoftMaxHead([8], 2),
                                   value_to_head='[MASK]')
    with pytest.raises(B4CError):
        model({'asin': [['a', '[MASK]']]}, training=False)      # model on CPU -> refuses


def test_vocab_lookup_chain_and_constants_match_oracle():
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead, constants
    from bert4clickpath_amd.clickstream_transformer.clickstream_transformer import TransformerInputPrep
    assert constants.RESERVED_TOKENS == nr.RESERVED_TOKENS and constants.LABEL_PAD == nr.LABEL_PAD
    assert (constants.INPUT_PAD, constants.MASK_ID, constants.CLS, constants.SEP) == (nr.INPUT_PAD, nr.MASK_ID, nr.CLS, nr.SEP)
    vocab = ['B0%02d' % i for i in range(27)]
    model = ClickstreamTransformer({'items': ['asin']}, {'items': vocab}, {'items': 16}, SoftMaxHead([8], 27),
                                   value_to_head='[MASK]')
    rows = [['B003', '[MASK]', 'B011', 'ZZZ', 'B026'], ['B001', 'B002', '[PAD]', '[PAD]', '[PAD]'],
            ['[MASK]', 'B005', 'B005', '[MASK]', '[PAD]']]
    feats, starts, ends = TransformerInputPrep({'items': ['asin']})({'asin': rows, 'other': 1})
    assert 'asin' not in feats and feats['other'] == 1
    chained = feats['items']
    assert chained.tolist() == nr.chain_sequences([rows])
    table, oov, size = nr.build_lookup(vocab)
    assert np.array_equal(model.lookup('items', chained), nr.lookup(table, oov, chained.tolist()))
    assert model.embedding_sizes['items'] == size == 38
    assert (starts, ends) == nr.segment_bounds(chained[0].tolist())
    # integer inputs chain with ids 3 / 4
    ids = torch.tensor([[11, 12, 0], [13, 0, 0]])
    f2, _, _ = TransformerInputPrep({'items': ['asin']})({'asin': ids})
    assert f2['items'].tolist() == [[3, 4, 11, 12, 0, 4], [3, 4, 13, 0, 0, 4]]
    # two chained sequences: [CLS] [SEP] s1 [SEP] s2 [SEP]
    f3, st3, en3 = TransformerInputPrep({'items': ['a', 'b']})({'a': ids, 'b': ids[:, :1]})
    assert f3['items'].shape == (2, 8) and en3 == [1, 5, 7] and st3 == [0, 2, 6]
    with pytest.raises(AssertionError):
        ClickstreamTransformer({'items': ['asin']}, {'items': vocab}, {'items': 16}, SoftMaxHead([8], 27),
                               value_to_head='[MASK]', segment_to_head=0)
    with pytest.raises(KeyError):
        ClickstreamTransformer({'items': ['asin']}, {'items': vocab}, {'items': 16, 'events': 8}, SoftMaxHead([8], 27),
                               value_to_head='[MASK]')
    with pytest.raises(IsADirectoryError):
        ClickstreamTransformer({'items': ['asin']}, {'items': os.path.dirname(__file__)}, {'items': 16},
                               SoftMaxHead([8], 27), value_to_head='[MASK]')


def test_cloze_masking_rules_match_oracle():
    from bert4clickpath_amd import input_pipeline as ip
    assert [ip.n_masked(n) for n in range(200)] == [nr.n_masked(n) for n in range(200)]
    vocab = ['i%d' % i for i in range(30)]
    table = {t: i for i, t in enumerate(vocab)}
    items = ['i%d' % i for i in (4, 9, 1, 1, 22, 17, 3, 8, 8, 2, 5)]
    a = ip.cloze_data_prep(items, ip.TRAIN, table, np.random.default_rng(5))
    b = nr.cloze_data_prep(items, 'train', vocab, np.random.default_rng(5))
    assert a[0] == b[0] and a[1].tolist() == b[1].tolist()
    a = ip.cloze_data_prep(items, ip.EVAL, table)
    b = nr.cloze_data_prep(items, 'eval', vocab)
    assert a[0] == b[0] and a[1].tolist() == b[1].tolist() == [5.0]
    its, labs = ip.padded_batch([['a', 'b'], ['c']], [np.asarray([1.0]), np.asarray([], np.float32)])
    assert its.tolist() == [['a', 'b'], ['c', '[PAD]']] and labs.tolist() == [[1.0], [-1.0]]


def test_synthetic_batch_invariants():
    from bert4clickpath_amd import input_pipeline as ip
    B, S, V = 64, 50, 300
    b = ip.synthetic_cloze_batch(B, S, V, seed=9, min_len=3, n_extra_features=1, extra_vocab=20)
    ids = b['ids']
    assert ids.shape == (B, S) and (ids[:, 0] == 3).all() and (ids[:, 1] == 4).all() and (ids[:, -1] == 4).all()
    idx, counts = nr.mask_positions(ids, nr.MASK_ID)
    assert np.array_equal(b['flat_idx'], (idx[:, 0] * S + idx[:, 1]).astype(np.int32))     # row-major, sorted
    assert np.array_equal(counts, np.minimum((2 * b['lens']) // 5, 10))
    assert ((b['labels'] >= 0) & (b['labels'] < V)).all()
    lp = b['labels_padded']
    assert np.array_equal(lp[lp != -1].astype(np.int32), b['labels'])
    e = b['extra'][0]
    assert ((e == 0) == (ids == 0)).all() and (e[:, 0] == 3).all()
    again = ip.synthetic_cloze_batch(B, S, V, seed=9, min_len=3)
    assert np.array_equal(again['ids'], ids)                       # seeded -> reproducible
    full = ip.synthetic_cloze_batch(4, 200, 50000, seed=1, full_length=True)
    assert (full['ids'] != 0).all() and len(full['labels']) == 40


def test_positional_encoding_and_padding_mask_match_oracle():
    from bert4clickpath_amd.clickstream_transformer.transformer import create_padding_mask, positional_encoding, create_segment_markers
    for d in (64, 128, 256):
        pe = positional_encoding(512, d)
        assert pe.shape == (1, 512, d) and pe.dtype == torch.float32
        assert np.array_equal(pe.numpy(), nr.positional_encoding(512, d))
    seq = torch.tensor([[3, 4, 7, 0, 0, 4], [3, 4, 7, 8, 9, 4]])
    assert np.array_equal(create_padding_mask(seq).numpy(), nr.create_padding_mask(seq.numpy()))
    assert np.array_equal(create_segment_markers(seq).numpy(), nr.create_segment_markers(seq.numpy()))


def test_state_dict_names_mirror_keras_tree():
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    rng = np.random.default_rng(0)
    P = nr.init_params(rng, {'items': 48}, {'items': 64}, 2, 100, [16, 8], 37)
    model = ClickstreamTransformer({'items': ['asin']}, {'items': ['x%d' % i for i in range(37)]}, {'items': 64},
                                   SoftMaxHead([16, 8], 37), value_to_head='[MASK]', num_encoder_layers=2,
                                   num_attention_heads=2)
    sd = model.state_dict()
    assert set(sd.keys()) == set(P.keys())
    assert all(tuple(sd[k].shape) == P[k].shape for k in P)
    assert model.transformer.encoder_ff_dim == 100       # hard-coded by the reference's wrapper


def test_reference_import_paths_resolve():
    from clickstream_transformer.clickstream_transformer import ClickstreamTransformer   # noqa: F401
    from clickstream_transformer.transformer import Transformer, Encoder, EncoderLayer, MultiHeadAttention   # noqa: F401
    from clickstream_transformer.transformer import point_wise_feed_forward_network, scaled_dot_product_attention   # noqa: F401
    from clickstream_transformer.head import SoftMaxHead   # noqa: F401
    from clickstream_transformer.losses import MaskedLoss   # noqa: F401
    from clickstream_transformer.constants import RESERVED_TOKENS, LABEL_PAD   # noqa: F401
    from clickstream_transformer.training_utils import load_vocabulary   # noqa: F401


def test_row_pitch_of_vocabulary_wide_tensors():
    """ops.row_pitch / empty_rows: vocabulary-wide rows start on 256-byte boundaries (128 elements), narrow ones keep their
    width; the view handed out has the requested shape, and a (B, M, V) view of it reshapes to rows without a copy."""
    import torch
    from bert4clickpath_amd import ops
    assert ops.row_pitch(104) == 104 and ops.row_pitch(2047) == 2047
    assert ops.row_pitch(2048) == 2048 and ops.row_pitch(50000) == 50048 and ops.row_pitch(100000) == 100096
    t = ops.empty_rows(6, 50000, torch.bfloat16, 'cpu')
    assert t.shape == (6, 50000) and t.stride() == (50048, 1) and (t.stride(0) * t.element_size()) % 256 == 0
    assert ops.empty_rows(6, 104, torch.float32, 'cpu').is_contiguous()
    v = t.view(2, 3, 50000)
    r = v.reshape(-1, 50000)
    assert r.data_ptr() == t.data_ptr() and r.stride() == (50048, 1)
    g = ops._rows_ok(t, torch.bfloat16)
    assert g.data_ptr() == t.data_ptr()                      # pitched rows are taken as they are
    assert ops._rows_ok(t.t()[:8].t(), torch.bfloat16).stride(1) == 1


def test_background_plan_partitions_the_vocabulary_tiles():
    """ops._background_plan: kicks + 1 pieces (one at the head's backward, one behind every attention backward launch of
    the previous pass), contiguous, covering every 128-id tile exactly once, first piece the largest."""
    from bert4clickpath_amd import ops
    for n_tiles in (1, 2, 7, 391, 782, 15625):
        for kicks in (0, 1, 3, 5, 11):
            cuts = ops._background_plan(n_tiles, kicks)
            assert cuts[0] == 0 and cuts[-1] == n_tiles and len(cuts) == kicks + 2
            assert all(a <= b for a, b in zip(cuts[:-1], cuts[1:]))
            sizes = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
            assert sum(sizes) == n_tiles
            if n_tiles >= 100 and kicks:
                assert sizes[0] == max(sizes) and min(sizes) > 0


def test_feature_sum_restatement_and_config():
    """feature_combine='sum' (an extension: the reference only concatenates, transformer.py:384-388): the restatement adds
    the gathered rows in feature order before the sqrt(d) scale; the host class takes d_model from the one shared width."""
    from oracle import torch_ref as tr
    from bert4clickpath_amd.clickstream_transformer.transformer import Transformer
    rng = np.random.default_rng(2)
    d, S = 16, 7
    ids = {'a': rng.integers(0, 9, (2, S)), 'b': rng.integers(0, 5, (2, S))}
    tabs = {'a': rng.standard_normal((9, d)), 'b': rng.standard_normal((5, d))}
    got = nr.embed_concat_pe(ids, tabs, d, np.float64, combine='sum')
    want = (tabs['a'][ids['a']] + tabs['b'][ids['b']]) * np.float64(np.sqrt(np.float32(d))) + nr.positional_encoding(nr.MAX_POSITION, d)[:, :S]
    assert np.array_equal(got, want)
    cat = nr.embed_concat_pe(ids, tabs, 2 * d, np.float64)
    assert cat.shape == (2, S, 2 * d)
    # numpy and torch restatements agree through a whole encoder
    t = Transformer(1, 2, {'a': 9, 'b': 5}, {'a': d, 'b': d}, 100, 0.0, feature_combine='sum')
    assert t.d_model == d and t.get_config()['feature_combine'] == 'sum'
    tP = {k: v.detach().double() for k, v in t.state_dict().items() if 'pos_encoding' not in k}
    e_t = tr.transformer_forward({k: torch.from_numpy(v) for k, v in ids.items()}, tP, 1, 2, combine='sum')
    e_n = nr.transformer_forward(ids, {k: v.numpy() for k, v in tP.items()}, 1, 2, np.float64, combine='sum')
    assert float(np.abs(e_t.numpy() - e_n).max()) < 1e-12
    assert 'feature_combine' not in Transformer(1, 2, {'a': 9}, {'a': d}, 100, 0.0).get_config()      # the reference's config keys
    import pytest
    with pytest.raises(ValueError):
        Transformer(1, 2, {'a': 9, 'b': 5}, {'a': d, 'b': 8}, 100, 0.0, feature_combine='sum')
    with pytest.raises(ValueError):
        Transformer(1, 2, {'a': 9}, {'a': d}, 100, 0.0, feature_combine='sum')
    with pytest.raises(ValueError):
        Transformer(1, 2, {'a': 9}, {'a': d}, 100, 0.0, feature_combine='mean')


def test_only_the_checkers_touch_the_oracle():
    """oracle/ is test infrastructure: the package, the examples and bench.py's timed path never import it -- only tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline leg do -- and the package has no CPU fallback to route through."""
    import ast
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent

    def oracle_imports(path):
        tree = ast.parse(path.read_text())
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and (node.module or '').split('.')[0] == 'oracle':
                hits.append(node.lineno)
            if isinstance(node, ast.Import) and any(a.name.split('.')[0] == 'oracle' for a in node.names):
                hits.append(node.lineno)
        return hits
    for path in list((root / 'bert4clickpath_amd').rglob('*.py')) + list((root / 'clickstream_transformer').rglob('*.py')) + \
            list((root / 'examples').rglob('*.py')):
        assert not oracle_imports(path), '%s imports the oracle' % path
    # bench.py: inside cpu_baseline() only;  __graft_entry__.py: inside smoke() only
    for name, fn in (('bench.py', 'cpu_baseline'), ('__graft_entry__.py', 'smoke')):
        tree = ast.parse((root / name).read_text())
        allowed = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and node.name == fn:
                allowed = set(range(node.lineno, node.end_lineno + 1))
        lines = oracle_imports(root / name)
        assert lines and all(l in allowed for l in lines), (name, lines)


def test_a_missing_library_is_an_error_not_a_fallback():
    """the product path without libb4c_hip.so: a B4CError that says how to build it -- nothing is computed another way"""
    import subprocess
    import sys
    code = ("import torch\n"
            "from bert4clickpath_amd import ops\n"
            "from bert4clickpath_amd._lib import B4CError\n"
            "try:\n"
            "    ops.keep_mask(1, 8, 0.5) if False else ops.L.lib()\n"
            "except B4CError as e:\n"
            "    print('B4CError:', e)\n"
            "else:\n"
            "    print('loaded')\n")
    env = dict(os.environ, B4C_LIB_PATH='/nonexistent/libb4c_hip.so', PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    assert 'B4CError' in out.stdout and 'no CPU fallback' in out.stdout and 'loaded' not in out.stdout


def test_a_library_of_another_abi_is_refused_at_load(monkeypatch):
    """The argument lists bound in _lib.lib() belong to one ABI version; a library that answers another one must not be bound
    (same names, older lists: a device pointer would be taken for a stream).  B4CError, not AttributeError."""
    from bert4clickpath_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'ABI_VERSION', _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.B4CError, match='ABI'):
        _lib.lib()
    monkeypatch.setattr(_lib, 'ABI_VERSION', _lib.ABI_VERSION - 1)
    assert _lib.lib().b4c_abi_version() == _lib.ABI_VERSION
