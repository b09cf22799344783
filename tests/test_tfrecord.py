"""TFRecord / tf.train.Example reader-writer (bert4clickpath_amd/tfrecord.py) against hand-assembled wire bytes and
the published CRC-32C check value: the format of the reference's data files (data_prep/main.py:88-110,
source/input_pipeline.py:147-160)."""
import struct

import pytest

from bert4clickpath_amd import tfrecord as tfr


def test_crc32c_check_value_and_mask():
    assert tfr.crc32c(b'123456789') == 0xE3069283                     # CRC-32C (Castagnoli) check value
    assert tfr.crc32c(b'') == 0
    c = tfr.crc32c(b'abc')
    assert tfr.masked_crc32c(b'abc') == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_parse_hand_assembled_example():
    # Example{features{ feature{"id": int64_list[7, -1]}, feature{"asin": bytes_list["B01", "B02"]}, feature{"w": float_list[0.5]} }}
    def ld(fno, payload):
        return bytes([(fno << 3) | 2, len(payload)]) + payload
    int_list = ld(1, bytes([7]) + b'\xff' * 9 + b'\x01')               # packed varints: 7, -1 (ten bytes)
    f_id = ld(3, int_list)
    f_asin = ld(1, ld(1, b'B01') + ld(1, b'B02'))
    f_w = ld(2, ld(1, struct.pack('<f', 0.5)))
    entries = b''.join(ld(1, ld(1, k) + ld(2, v)) for k, v in ((b'id', f_id), (b'asin', f_asin), (b'w', f_w)))
    ex = ld(1, entries)
    got = tfr.parse_example(ex)
    assert got == {'id': [7, -1], 'asin': [b'B01', b'B02'], 'w': [0.5]}
    # unpacked encodings of the repeated scalars parse to the same lists
    f_id2 = ld(3, bytes([0x08, 7]) + bytes([0x08]) + b'\xff' * 9 + b'\x01')
    f_w2 = ld(2, bytes([0x0D]) + struct.pack('<f', 0.5))
    ex2 = ld(1, ld(1, ld(1, b'id') + ld(2, f_id2)) + ld(1, ld(1, b'w') + ld(2, f_w2)))
    assert tfr.parse_example(ex2) == {'id': [7, -1], 'w': [0.5]}


def test_round_trip_and_corruption(tmp_path):
    rows = [{'reviewerID': 'A1', 'asin': ['B0001', 'B0002', 'B0003']},
            {'reviewerID': 'A2', 'asin': []},
            {'reviewerID': 'A3', 'asin': ['x' * 300], 'unixReviewTime': [1400000000, -5], 'score': [1.5, 2.25]}]
    p = tmp_path / 'part-00000.tfrecord'
    assert tfr.write_records(str(p), (tfr.encode_example(r) for r in rows)) == 3
    back = list(tfr.read_examples(str(tmp_path / '*.tfrecord'), verify_crc=True))
    assert back[0] == {'reviewerID': [b'A1'], 'asin': [b'B0001', b'B0002', b'B0003']}
    assert back[1] == {'reviewerID': [b'A2'], 'asin': []}
    assert back[2]['unixReviewTime'] == [1400000000, -5] and back[2]['score'] == [1.5, 2.25] and len(back[2]['asin'][0]) == 300
    ids, seqs = tfr.read_item_sequences(str(p))
    assert ids == ['A1', 'A2', 'A3'] and seqs[0] == ['B0001', 'B0002', 'B0003'] and seqs[1] == []
    raw = bytearray(p.read_bytes())
    raw[20] ^= 0x01                                                   # flip a payload bit of the first record
    p.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        list(tfr.read_records(str(p), verify_crc=True))
    p.write_bytes(bytes(raw[:-3]))                                    # truncated file
    with pytest.raises(ValueError):
        list(tfr.read_records(str(p)))


def test_sequences_feed_the_cloze_pipeline(tmp_path):
    import numpy as np
    from bert4clickpath_amd import input_pipeline
    p = tmp_path / 'd.tfrecord'
    tfr.write_records(str(p), [tfr.encode_example({'reviewerID': 'u', 'asin': ['i%d' % k for k in range(12)]})])
    _, seqs = tfr.read_item_sequences(str(p))
    rng = np.random.default_rng(0)
    table = {'i%d' % k: k for k in range(12)}
    items, labels = input_pipeline.cloze_data_prep(seqs[0], input_pipeline.TRAIN, table, rng)
    # TRAIN drops the last item, then masks n_masked positions (input_pipeline.py:92-128 of the reference)
    assert len(items) == 11 and sum(t == '[MASK]' for t in items) == input_pipeline.n_masked(11) == len(labels)
