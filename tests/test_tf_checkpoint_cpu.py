"""TF V2 checkpoint reader/writer - CPU tests.

Parity unpinned: neither TensorFlow nor a checkpoint written by it exists here or in the reference, so
the format restatement is checked against (i) published CRC-32C known answers (RFC 3720 B.4), (ii) a
table assembled by hand in this file, byte by byte from the LevelDB table layout and the
BundleEntryProto field numbers, independent of the module's writer, and (iii) round trips through
the module's own writer."""
import os
import struct

import numpy as np
import pytest

from nmt_gan_amd import tf_checkpoint as ck


def test_crc32c_known_answers():
    assert ck.crc32c(b"123456789") == 0xE3069283
    assert ck.crc32c(bytes(32)) == 0x8A9136AA                      # RFC 3720 B.4
    assert ck.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert ck.crc32c(bytes(range(32))) == 0x46DD794E
    assert ck.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert ck.crc32c(b"") == 0
    assert ck.unmask_crc(ck.mask_crc(0xDEADBEEF)) == 0xDEADBEEF
    # leveldb crc32c_test: Mask(Value("foo")) differs from the raw value and unmasks back
    v = ck.crc32c(b"foo")
    assert ck.mask_crc(v) != v and ck.unmask_crc(ck.mask_crc(ck.mask_crc(v))) == ck.mask_crc(v)


@pytest.mark.parametrize("n", [1 << 14, (1 << 14) + 1, 70001, (1 << 24) + 5])
def test_crc32c_chunked_path_equals_bytewise(n):
    data = np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8)
    assert ck.crc32c(data) == ck._crc_bytes(data.tobytes(), 0xFFFFFFFF) ^ 0xFFFFFFFF


def _hand_table(compress_literal=False):
    """A one-block index written out by hand: header entry + tensor "ab" float32 [2,3] + tensor "ac"
    int32 [] (scalar) at offset 24."""
    data = np.arange(6, dtype="<f4").tobytes() + struct.pack("<i", -7)
    crc_ab = ck.mask_crc(ck.crc32c(data[:24]))
    crc_ac = ck.mask_crc(ck.crc32c(data[24:]))
    header = bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])
    shape_ab = bytes([0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03])
    e_ab = bytes([0x08, 0x01, 0x12, len(shape_ab)]) + shape_ab + bytes([0x28, 24, 0x35]) + struct.pack("<I", crc_ab)
    e_ac = bytes([0x08, 0x03, 0x12, 0x00, 0x20, 24, 0x28, 4, 0x35]) + struct.pack("<I", crc_ac)
    block = b""
    block += bytes([0, 0, len(header)]) + header                    # key ""
    block += bytes([0, 2, len(e_ab)]) + b"ab" + e_ab                # shared 0, non-shared 2
    block += bytes([1, 1, len(e_ac)]) + b"c" + e_ac                 # shares "a" with the previous key
    block += struct.pack("<II", 0, 1)                               # one restart at 0
    out = bytearray()

    def emit(contents, snappy=False):
        off = len(out)
        if snappy:                                                  # snappy: varint length + one literal element
            n = len(contents)
            assert n < 256
            tag = bytes([(n - 1) << 2]) if n <= 60 else bytes([60 << 2, n - 1])
            payload = ck._put_varint(n) + tag + contents
            out.extend(payload + b"\x01" + struct.pack("<I", ck.mask_crc(ck.crc32c(payload + b"\x01"))))
            return ck._put_varint(off) + ck._put_varint(len(payload))
        out.extend(contents + b"\x00" + struct.pack("<I", ck.mask_crc(ck.crc32c(contents + b"\x00"))))
        return ck._put_varint(off) + ck._put_varint(len(contents))

    h_data = emit(block, snappy=compress_literal)
    h_meta = emit(struct.pack("<II", 0, 1))
    index = bytes([0, 2, len(h_data)]) + b"ad" + h_data + struct.pack("<II", 0, 1)   # separator key >= "ac"
    h_index = emit(index)
    footer = h_meta + h_index
    out.extend(footer + bytes(40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57))
    return bytes(out), data


@pytest.mark.parametrize("snappy", [False, True])
def test_reads_hand_assembled_bundle(tmp_path, snappy):
    index, data = _hand_table(compress_literal=snappy)
    prefix = str(tmp_path / "model")
    open(prefix + ".index", "wb").write(index)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    assert ck.list_variables(prefix) == [("ab", (2, 3)), ("ac", ())]
    got = ck.load_checkpoint(prefix, verify_data=True)
    assert got["ab"].dtype == np.float32 and (got["ab"] == np.arange(6, dtype=np.float32).reshape(2, 3)).all()
    assert got["ac"].dtype == np.int32 and got["ac"].shape == () and int(got["ac"]) == -7


def test_corruption_is_detected(tmp_path):
    index, data = _hand_table()
    prefix = str(tmp_path / "model")
    bad = bytearray(index)
    bad[10] ^= 0x40
    open(prefix + ".index", "wb").write(bytes(bad))
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    with pytest.raises(ck.CheckpointError):
        ck.read_index(prefix)
    open(prefix + ".index", "wb").write(index)
    open(prefix + ".data-00000-of-00001", "wb").write(data[:8] + b"\x01" + data[9:])
    with pytest.raises(ck.CheckpointError):
        ck.load_checkpoint(prefix, verify_data=True)
    open(prefix + ".data-00000-of-00001", "wb").write(data[:20])
    with pytest.raises(ck.CheckpointError):
        ck.load_checkpoint(prefix)
    open(prefix + ".index", "wb").write(index[:-1] + b"\x00")
    with pytest.raises(ck.CheckpointError):
        ck.read_index(prefix)


def test_snappy_copies():
    # "abcdabcdabcdX": literal "abcd", copy(offset 4, len 8) overlapping its own output, literal "X"
    raw = ck._put_varint(13) + bytes([3 << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([0]) + b"X"
    assert ck._snappy_decompress(raw) == b"abcdabcdabcdX"
    raw2 = ck._put_varint(10) + bytes([1 << 2]) + b"xy" + bytes([((8 - 1) << 2) | 2, 2, 0])
    assert ck._snappy_decompress(raw2) == b"xy" * 5
    with pytest.raises(ck.CheckpointError):
        ck._snappy_decompress(ck._put_varint(4) + bytes([((4 - 1) << 2) | 2, 9, 0]))


def test_round_trip_many_blocks_and_state_file(tmp_path):
    rng = np.random.default_rng(3)
    W = {"decoder/block_%d/decoder_self_attention/qkv_transform_single/kernel" % i:
         rng.standard_normal((1, 1, 8, 24)).astype(np.float32) for i in range(40)}
    W.update({"discnn/conv_maxpool-%d/conv_batch_norm/bn/moving_variance" % w: rng.random(100 + 10 * w).astype(np.float32)
              for w in range(1, 50, 4)})
    W["global_step"] = np.int64(1234)
    W["empty"] = np.zeros((0, 4), np.float32)
    W["half"] = rng.standard_normal(9).astype(np.float16)
    prefix = str(tmp_path / "logdir" / "model_epoch_1_step_10")
    ck.save_checkpoint(prefix, W, block_size=512)
    got = ck.load_checkpoint(prefix, verify_data=True)
    assert sorted(got) == sorted(W)
    for k, v in W.items():
        assert got[k].dtype == np.asarray(v).dtype and got[k].shape == np.asarray(v).shape, k
        assert (got[k] == v).all(), k
    assert ck.latest_checkpoint(str(tmp_path / "logdir")) == prefix
    assert ck.latest_checkpoint(str(tmp_path)) is None
    lazy = ck.LazyCheckpoint(prefix)
    assert "global_step" in lazy and "nope" not in lazy and int(lazy["global_step"]) == 1234
    with pytest.raises(KeyError):
        lazy["nope"]
    only = ck.load_checkpoint(prefix, names=["half"])
    assert list(only) == ["half"]


def test_facade_weight_dicts_survive_a_checkpoint(tmp_path):
    """The synthetic generator + discriminator weight dicts written as bundles and
    read back are identical, so Model.init_and_restore(modelFile=prefix) / DisCNN(reload=True, saveto=prefix)
    see the arrays load_weights() was given."""
    from nmt_gan_amd import synthetic
    Wg = synthetic.generator_weights(64, 2, 96, 80, seed=0)
    Wd = synthetic.discriminator_weights(12, 16, 80, 96, seed=1)
    ck.save_checkpoint(str(tmp_path / "g" / "model"), Wg)
    ck.save_checkpoint(str(tmp_path / "d" / "disc"), Wd, update_state=False)
    assert not os.path.exists(str(tmp_path / "d" / "checkpoint"))
    for W, prefix in ((Wg, str(tmp_path / "g" / "model")), (Wd, str(tmp_path / "d" / "disc"))):
        lazy = ck.LazyCheckpoint(prefix, verify_data=True)
        assert set(lazy.keys()) == set(W)
        for k in W:
            assert (lazy[k] == np.asarray(W[k])).all() and lazy[k].shape == np.asarray(W[k]).shape
