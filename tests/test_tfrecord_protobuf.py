"""tf.train.Example payloads (the reference's data_prep writes them, data_utils.py:7-50; the pipeline reads them,
input_pipeline.py) against an INDEPENDENT implementation of the wire format: Google's protobuf runtime with the published
schema of tensorflow/core/example/{example,feature}.proto built at run time (TensorFlow itself is not installable here, protobuf
is).  Randomly drawn feature maps (hypothesis): what this repo encodes, protobuf parses to the same values; what protobuf
serialises -- packed or not -- this repo parses to the same values."""
import struct

import pytest
from hypothesis import given, settings, strategies as st

from bert4clickpath_amd import tfrecord as tfr

pb = pytest.importorskip('google.protobuf')
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402


def _example_class():
    f = descriptor_pb2.FileDescriptorProto(name='b4c_example.proto', package='b4c', syntax='proto3')
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = f.message_type.add()
        m.name = name
        return m

    def field(m, name, number, ftype, label=T.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
        x = m.field.add()
        x.name, x.number, x.type, x.label = name, number, ftype, label
        if type_name:
            x.type_name = type_name
        if packed is not None:
            x.options.packed = packed
        if oneof is not None:
            x.oneof_index = oneof
        return x
    field(msg('BytesList'), 'value', 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    field(msg('FloatList'), 'value', 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    field(msg('Int64List'), 'value', 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    field(msg('FloatListU'), 'value', 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=False)         # the unpacked encodings a proto2
    field(msg('Int64ListU'), 'value', 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=False)         # writer may produce
    for suffix in ('', 'U'):
        m = msg('Feature' + suffix)
        m.oneof_decl.add().name = 'kind'
        field(m, 'bytes_list', 1, T.TYPE_MESSAGE, type_name='.b4c.BytesList', oneof=0)
        field(m, 'float_list', 2, T.TYPE_MESSAGE, type_name='.b4c.FloatList' + suffix, oneof=0)
        field(m, 'int64_list', 3, T.TYPE_MESSAGE, type_name='.b4c.Int64List' + suffix, oneof=0)
        fs = msg('Features' + suffix)
        e = fs.nested_type.add()
        e.name = 'FeatureEntry'
        e.options.map_entry = True
        field(e, 'key', 1, T.TYPE_STRING)
        field(e, 'value', 2, T.TYPE_MESSAGE, type_name='.b4c.Feature' + suffix)
        field(fs, 'feature', 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name='.b4c.Features%s.FeatureEntry' % suffix)
        field(msg('Example' + suffix), 'features', 1, T.TYPE_MESSAGE, type_name='.b4c.Features' + suffix)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    get = getattr(message_factory, 'GetMessageClass', None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        return fac.GetPrototype(pool.FindMessageTypeByName('b4c.Example')), fac.GetPrototype(pool.FindMessageTypeByName('b4c.ExampleU'))
    return get(pool.FindMessageTypeByName('b4c.Example')), get(pool.FindMessageTypeByName('b4c.ExampleU'))


Example, ExampleU = _example_class()
f32 = st.floats(width=32, allow_nan=False)
names = st.text(alphabet=st.characters(min_codepoint=33, max_codepoint=0x24F), min_size=1, max_size=12)
feature_values = st.one_of(st.lists(st.binary(max_size=40), max_size=6),
                           st.lists(f32, min_size=1, max_size=8),
                           st.lists(st.integers(-2 ** 63, 2 ** 63 - 1), min_size=1, max_size=8))
feature_maps = st.dictionaries(names, feature_values, max_size=6)


def _to_pb(cls, features):
    ex = cls()
    for k, v in features.items():
        ft = ex.features.feature[k]
        if not v:
            ft.bytes_list.SetInParent()
        elif isinstance(v[0], bytes):
            ft.bytes_list.value.extend(v)
        elif isinstance(v[0], float):
            ft.float_list.value.extend(v)
        else:
            ft.int64_list.value.extend(v)
    return ex


def _from_pb(ex):
    out = {}
    for k, ft in ex.features.feature.items():
        kind = ft.WhichOneof('kind')
        out[k] = [] if kind is None else list(getattr(ft, kind).value)
    return out


def _f32(v):
    return [struct.unpack('<f', struct.pack('<f', x))[0] if isinstance(x, float) else x for x in v]


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(features=feature_maps)
def test_what_this_repo_encodes_protobuf_reads(features):
    blob = tfr.encode_example(features)
    ex = Example()
    ex.ParseFromString(blob)
    got = _from_pb(ex)
    assert set(got) == set(features)
    for k, v in features.items():
        assert got[k] == _f32(v), k


@settings(max_examples=200, deadline=None, derandomize=True, database=None)
@given(features=feature_maps, packed=st.booleans())
def test_what_protobuf_writes_this_repo_reads(features, packed):
    blob = _to_pb(Example if packed else ExampleU, features).SerializeToString()
    got = tfr.parse_example(blob)
    assert set(got) == set(features)
    for k, v in features.items():
        assert got[k] == _f32(v), k


@settings(max_examples=50, deadline=None, derandomize=True, database=None)
@given(rows=st.lists(feature_maps, max_size=5))
def test_record_framing_round_trip(tmp_path_factory, rows):
    """length | masked crc32c(length) | payload | masked crc32c(payload): written here, read back with the CRCs verified"""
    p = tmp_path_factory.mktemp('rec') / 'x.tfrecord'
    tfr.write_records(str(p), [tfr.encode_example(r) for r in rows])
    back = [tfr.parse_example(b) for b in tfr.read_records(str(p), verify_crc=True)]
    assert len(back) == len(rows)
    for a, b in zip(back, rows):
        assert {k: v for k, v in a.items()} == {k: _f32(v) for k, v in b.items()}
