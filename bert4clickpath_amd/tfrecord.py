"""TFRecord files of `tf.train.Example` without TensorFlow: the on-disk format of the reference's training data
(examples/BERT4Rec/data_prep/main.py:88-110 writes, source/input_pipeline.py:147-160 reads
`{'reviewerID': FixedLenFeature([], string), 'asin': VarLenFeature(string)}`; the writer helpers are
clickstream_transformer/data_utils.py:7-50, 412-480).

Record framing (tensorflow/core/lib/io/record_writer.cc):
    uint64 length | uint32 masked_crc32c(length) | byte data[length] | uint32 masked_crc32c(data)
    masked_crc = ((crc >> 15) | (crc << 17)) + 0xa282ead8   (mod 2^32), crc = CRC-32C (Castagnoli)
Payload (tensorflow/core/example/example.proto, feature.proto):
    Example { Features features = 1 }   Features { map<string, Feature> feature = 1 }
    Feature { oneof kind { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 } }
    BytesList { repeated bytes value = 1 }  FloatList { repeated float value = 1 [packed] }
    Int64List { repeated int64 value = 1 [packed] }

Host-side I/O only (the hot path starts at the id tensors); plain Python, no dependency."""
import glob
import struct

_MASK_DELTA = 0xA282EAD8


def _make_crc_table():
    tab = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TABLE = _make_crc_table()


def crc32c(data):
    c = 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


# ---- protobuf wire format (the subset example.proto uses) -------------------------------------------
def _read_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _write_varint(v):
    if v < 0:
        v += 1 << 64          # int64 two's complement, ten bytes on the wire
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf):
    """(field_number, wire_type, value) triples of one message; value is int (varint / fixed) or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = buf[pos:pos + ln]
            if len(v) != ln:
                raise ValueError('truncated length-delimited field')
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError('unsupported wire type %d' % wt)
        yield fno, wt, v


def _parse_feature(buf):
    for fno, wt, v in _fields(buf):
        if wt != 2:
            continue
        if fno == 1:            # BytesList
            return [bytes(x) for f, w, x in _fields(v) if f == 1 and w == 2]
        if fno == 2:            # FloatList: packed or not
            out = []
            for f, w, x in _fields(v):
                if f != 1:
                    continue
                if w == 2:
                    out.extend(struct.unpack('<%df' % (len(x) // 4), x))
                elif w == 5:
                    out.append(struct.unpack('<f', x)[0])
            return out
        if fno == 3:            # Int64List: packed or not
            out = []
            for f, w, x in _fields(v):
                if f != 1:
                    continue
                if w == 2:
                    p = 0
                    while p < len(x):
                        val, p = _read_varint(x, p)
                        out.append(val - (1 << 64) if val >= (1 << 63) else val)
                elif w == 0:
                    out.append(x - (1 << 64) if x >= (1 << 63) else x)
            return out
    return []                   # a Feature with no kind set


def parse_example(serialized):
    """bytes of one tf.train.Example -> {feature name: list of bytes | float | int} (every feature is a list,
    as tf.io.VarLenFeature yields; a FixedLenFeature([]) is the single element)."""
    out = {}
    for fno, wt, features in _fields(memoryview(serialized)):
        if fno != 1 or wt != 2:
            continue
        for f2, w2, entry in _fields(features):       # map<string, Feature> entries
            if f2 != 1 or w2 != 2:
                continue
            key, val = None, []
            for f3, w3, x in _fields(entry):
                if f3 == 1 and w3 == 2:
                    key = bytes(x).decode('utf-8')
                elif f3 == 2 and w3 == 2:
                    val = _parse_feature(x)
            if key is not None:
                out[key] = val
    return out


def _ld(fno, payload):
    return _write_varint((fno << 3) | 2) + _write_varint(len(payload)) + payload


def encode_example(features):
    """{name: list of bytes/str | float | int (or a scalar)} -> serialized tf.train.Example (the role of
    data_utils.py:7-50 `to_feature` / `encode_tf_example`: bytes -> BytesList, float -> FloatList, int -> Int64List)."""
    entries = b''
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, str, int, float)):
            v = [v]
        v = list(v)
        if v and isinstance(v[0], (bytes, str)):
            lst = b''.join(_ld(1, x.encode('utf-8') if isinstance(x, str) else x) for x in v)
            feat = _ld(1, lst)
        elif v and isinstance(v[0], float):
            feat = _ld(2, _ld(1, struct.pack('<%df' % len(v), *v)))
        elif v and isinstance(v[0], int):
            feat = _ld(3, _ld(1, b''.join(_write_varint(x) for x in v)))
        elif not v:
            feat = b''
        else:
            raise TypeError('feature %r: unsupported element type %s' % (name, type(v[0])))
        entries += _ld(1, _ld(1, name.encode('utf-8')) + _ld(2, feat))
    return _ld(1, entries)


# ---- record framing ---------------------------------------------------------------------------------
def read_records(path, verify_crc=False):
    """Yield the payload of every record of one TFRecord file."""
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise ValueError('%s: truncated record header' % path)
            length, len_crc = struct.unpack('<QI', head)
            if verify_crc and masked_crc32c(head[:8]) != len_crc:
                raise ValueError('%s: corrupted record length' % path)
            data = f.read(length)
            tail = f.read(4)
            if len(data) != length or len(tail) != 4:
                raise ValueError('%s: truncated record' % path)
            if verify_crc and masked_crc32c(data) != struct.unpack('<I', tail)[0]:
                raise ValueError('%s: corrupted record data' % path)
            yield data


def write_records(path, payloads):
    """Write serialized records (bytes) as one TFRecord file; returns the number written."""
    n = 0
    with open(path, 'wb') as f:
        for data in payloads:
            head = struct.pack('<Q', len(data))
            f.write(head)
            f.write(struct.pack('<I', masked_crc32c(head)))
            f.write(data)
            f.write(struct.pack('<I', masked_crc32c(data)))
            n += 1
    return n


def read_examples(pattern, verify_crc=False):
    """Parsed examples of every file matching `pattern` (e.g. data/*.tfrecord, input_pipeline.py:147-149), sorted by
    file name."""
    for path in sorted(glob.glob(pattern)):
        for rec in read_records(path, verify_crc):
            yield parse_example(rec)


def read_item_sequences(pattern, item_feature='asin', id_feature='reviewerID', verify_crc=False):
    """-> (ids, sequences): the reference's BERT4Rec records as Python lists of item strings, ready for
    input_pipeline.cloze_data_prep / ClickstreamTransformer.lookup."""
    ids, seqs = [], []
    for ex in read_examples(pattern, verify_crc):
        seqs.append([x.decode('utf-8') for x in ex.get(item_feature, [])])
        rid = ex.get(id_feature, [b''])
        ids.append(rid[0].decode('utf-8') if rid else '')
    return ids, seqs
