"""TensorFlow V2 checkpoint ("tensor bundle") reader/writer in pure Python + numpy.

The reference keeps its weights in tf.train.Saver bundles: the generator restores its trainable
variables from `modelFile` or the latest checkpoint of `config.train.logdir` (model.py:425-435) and
saves after every generator iteration (gan_train.py:118); the discriminator restores every variable of
its scope from `saveto` (cnn_discriminator.py:202-217).  This module reads such a bundle into the
{variable name: ndarray} dict the facades' load_weights() take (the reference's TF variable names), and writes
one, with nothing executed from the file and no TensorFlow.

Format, restated from TensorFlow 1.x (tensorflow/core/util/tensor_bundle, core/lib/io/table*; TF is
not under /root/reference and not installed, and the reference ships no checkpoint file, so this
restatement is **parity unpinned**: the tests round-trip through this module's own writer and check a
hand-assembled byte vector of the table layout, not a file TF wrote):

  <prefix>.index                 an SSTable (LevelDB table format): data blocks of prefix-compressed
                                 (key, value) entries + restart array, a metaindex block, an index block
                                 and a 48-byte footer ending in the magic 0xdb4775248b80fb57; every block is
                                 followed by 1 compression byte (0 raw, 1 snappy) and a masked CRC32C.
                                 key ""   -> BundleHeaderProto {num_shards=1, endianness=2, version=3}
                                 key name -> BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5,
                                             crc32c=6 (fixed32, masked), slices=7}
  <prefix>.data-SSSSS-of-NNNNN   raw little-endian tensor bytes at [offset, offset+size) of shard S
  checkpoint                     text file: model_checkpoint_path: "<name>"  (tf.train.latest_checkpoint)

Partitioned variables (entries with slices) are rejected: the reference creates none.
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_TRAILER_LEN = 5
RESTART_INTERVAL = 16
MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
           6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
           22: np.dtype("<u4"), 23: np.dtype("<u8")}
DT_BFLOAT16 = 14                         # returned widened to float32 (numpy has no bfloat16)
_DTYPE_CODES = {v: k for k, v in _DTYPES.items()}


class CheckpointError(ValueError):
    pass


# ----------------------------------------------------------------------------------------- CRC32C
def _make_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        t[i] = c
    return t


_CRC_T = _make_table()
_CRC_L = [int(v) for v in _CRC_T]
_ZOPS: Dict[int, List[List[int]]] = {}


def _crc_bytes(data: bytes, reg: int) -> int:
    t = _CRC_L
    for b in data:
        reg = t[(reg ^ b) & 0xFF] ^ (reg >> 8)
    return reg


def _zero_operator(L: int) -> List[List[int]]:
    """4 x 256 tables of the linear map 'register after L zero bytes' (the register update is linear
    over GF(2) when the data bytes are zero)."""
    if L not in _ZOPS:
        basis = (np.uint32(1) << np.arange(32, dtype=np.uint32)).astype(np.uint32)
        for _ in range(L):
            basis = _CRC_T[basis & 0xFF] ^ (basis >> 8)
        tabs = []
        for byte in range(4):
            tab = [0] * 256
            for v in range(1, 256):
                low = v & -v
                tab[v] = tab[v ^ low] ^ int(basis[8 * byte + low.bit_length() - 1])
            tabs.append(tab)
        _ZOPS[L] = tabs
    return _ZOPS[L]


def crc32c(data) -> int:
    """CRC-32C (Castagnoli, reflected, init/xorout 0xFFFFFFFF).  Large buffers are cut into equal chunks
    whose registers advance together in numpy (one byte column per iteration) and are then folded with
    the zero-advance operator, so a 100 MB tensor takes a fraction of a second instead of a minute."""
    buf = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8) if not isinstance(data, np.ndarray) \
        else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    n = buf.size
    if n < 1 << 14:
        return _crc_bytes(buf.tobytes(), 0xFFFFFFFF) ^ 0xFFFFFFFF
    L = 1024 if n < 1 << 24 else 4096
    n_chunks = (n + L - 1) // L
    padded = np.zeros(n_chunks * L, np.uint8)          # zeros in FRONT leave a zero register untouched
    padded[n_chunks * L - n:] = buf
    head = n_chunks * L - n
    padded[head:head + 4] ^= 0xFF                      # init 0xFFFFFFFF == first four message bytes inverted
    cols = np.ascontiguousarray(padded.reshape(n_chunks, L).T)
    reg = np.zeros(n_chunks, np.uint32)
    for j in range(L):
        reg = _CRC_T[(reg ^ cols[j]) & 0xFF] ^ (reg >> 8)
    z0, z1, z2, z3 = _zero_operator(L)
    acc = 0
    for r in reg.tolist():
        acc = z0[acc & 0xFF] ^ z1[(acc >> 8) & 0xFF] ^ z2[(acc >> 16) & 0xFF] ^ z3[acc >> 24] ^ r
    return acc ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m: int) -> int:
    rot = (m - MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ----------------------------------------------------------------------------------------- varints / protobuf
def _get_varint(b: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        if pos >= len(b):
            raise CheckpointError("truncated varint")
        c = b[pos]
        pos += 1
        out |= (c & 0x7F) << shift
        if not c & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise CheckpointError("varint too long")


def _put_varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _pb_fields(b: bytes) -> Iterable[Tuple[int, int, object]]:
    pos = 0
    while pos < len(b):
        tag, pos = _get_varint(b, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _get_varint(b, pos)
        elif wire == 1:
            v, pos = b[pos:pos + 8], pos + 8
        elif wire == 2:
            ln, pos = _get_varint(b, pos)
            v, pos = b[pos:pos + ln], pos + ln
            if len(v) != ln:
                raise CheckpointError("truncated protobuf field")
        elif wire == 5:
            v, pos = b[pos:pos + 4], pos + 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wire)
        yield field, wire, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_shape(b: bytes) -> Tuple[int, ...]:
    dims = []
    for field, _, v in _pb_fields(b):
        if field == 2:
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = _signed64(v2)
            dims.append(size)
        elif field == 3 and v:
            raise CheckpointError("tensor of unknown rank")
    return tuple(dims)


def _parse_entry(b: bytes) -> dict:
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, slices=0)
    for field, _, v in _pb_fields(b):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            e["shape"] = _parse_shape(v)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = _signed64(v)
        elif field == 5:
            e["size"] = _signed64(v)
        elif field == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif field == 7:
            e["slices"] += 1
    return e


def _parse_header(b: bytes) -> dict:
    h = dict(num_shards=0, endianness=0, producer=0)
    for field, _, v in _pb_fields(b):
        if field == 1:
            h["num_shards"] = v
        elif field == 2:
            h["endianness"] = v
        elif field == 3:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    h["producer"] = v2
    return h


def _encode_entry(dtype_code: int, shape, offset: int, size: int, crc_masked: int) -> bytes:
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype_code) + b"\x12" + _put_varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _put_varint(offset)           # shard_id 0 and zero offsets are proto3 defaults: omitted
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc_masked)
    return out


# ----------------------------------------------------------------------------------------- snappy (raw format)
def _snappy_decompress(b: bytes) -> bytes:
    n, pos = _get_varint(b, 0)
    out = bytearray()
    while pos < len(b):
        tag = b[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(b[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += b[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | b[pos]
            pos += 1
        elif kind == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(b[pos:pos + 2], "little")
            pos += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(b[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy block")
        for _ in range(ln):                            # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy length mismatch")
    return bytes(out)


# ----------------------------------------------------------------------------------------- table reader
def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    end = offset + size + BLOCK_TRAILER_LEN
    if offset < 0 or end > len(buf):
        raise CheckpointError("block handle outside the index file")
    raw, ctype = buf[offset:offset + size], buf[offset + size]
    if verify:
        want = unmask_crc(struct.unpack("<I", buf[offset + size + 1:end])[0])
        if crc32c(buf[offset:offset + size + 1]) != want:
            raise CheckpointError("index block checksum mismatch at offset %d" % offset)
    if ctype == 0:
        return raw
    if ctype == 1:
        return _snappy_decompress(raw)
    raise CheckpointError("unknown block compression %d" % ctype)


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise CheckpointError("block too small")
    n_restarts = struct.unpack("<I", block[-4:])[0]
    limit = len(block) - 4 - 4 * n_restarts
    if limit < 0:
        raise CheckpointError("bad restart array")
    pos, key, out = 0, b"", []
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise CheckpointError("corrupt block entry")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(prefix: str, verify: bool = True) -> Tuple[dict, Dict[str, dict]]:
    """(header, {tensor name: entry}) of <prefix>.index."""
    with open(prefix + ".index", "rb") as f:
        buf = f.read()
    if len(buf) < FOOTER_LEN:
        raise CheckpointError("index file shorter than a table footer")
    footer = buf[-FOOTER_LEN:]
    if struct.unpack("<Q", footer[40:])[0] != TABLE_MAGIC:
        raise CheckpointError("%s.index is not a TensorFlow V2 checkpoint index (bad table magic)" % prefix)
    pos = 0
    _, pos = _get_varint(footer, pos)                  # metaindex handle (unused)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    header, entries = None, {}
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for key, value in _block_entries(_read_block(buf, boff, bsize, verify)):
            if key == b"":
                header = _parse_header(value)
            else:
                entries[key.decode("utf-8")] = _parse_entry(value)
    if header is None:
        raise CheckpointError("bundle header entry missing")
    if header["endianness"] != 0:
        raise CheckpointError("big-endian bundle")
    return header, entries


def list_variables(prefix: str) -> List[Tuple[str, Tuple[int, ...]]]:
    return [(k, e["shape"]) for k, e in sorted(read_index(prefix)[1].items())]


def load_checkpoint(prefix: str, names: Optional[Iterable[str]] = None, verify_data: bool = False
                    ) -> Dict[str, np.ndarray]:
    """{variable name: ndarray} of a bundle (tf.train.Saver.restore without TensorFlow).  `names`
    restricts the tensors read; verify_data also checks every tensor's CRC32C."""
    header, entries = read_index(prefix)
    wanted = set(entries) if names is None else set(names)
    missing = wanted - set(entries)
    if missing:
        raise KeyError("not in checkpoint %s: %s" % (prefix, ", ".join(sorted(missing)[:8])))
    shards = {}
    out = {}
    for name in sorted(wanted):
        e = entries[name]
        if e["slices"]:
            raise CheckpointError("%s is a partitioned variable (tensor slices): not supported" % name)
        if e["dtype"] == DT_BFLOAT16:
            dt = np.dtype("<u2")
        elif e["dtype"] in _DTYPES:
            dt = _DTYPES[e["dtype"]]
        else:
            raise CheckpointError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if count * dt.itemsize != e["size"]:
            raise CheckpointError("%s: size %d does not match shape %s" % (name, e["size"], e["shape"]))
        sid = e["shard_id"]
        if sid not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"])
            shards[sid] = np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, np.uint8)
        data = shards[sid]
        if e["offset"] < 0 or e["offset"] + e["size"] > data.size:
            raise CheckpointError("%s: data range outside shard %d" % (name, sid))
        raw = np.array(data[e["offset"]:e["offset"] + e["size"]])
        if verify_data and e["crc32c"] is not None and crc32c(raw) != unmask_crc(e["crc32c"]):
            raise CheckpointError("%s: tensor checksum mismatch" % name)
        a = raw.view(dt).reshape(e["shape"])
        if e["dtype"] == DT_BFLOAT16:
            a = (a.astype(np.uint32) << 16).view(np.float32)
        out[name] = a
    return out


class LazyCheckpoint(object):
    """Mapping view of a bundle that reads a tensor when it is asked for, so load_weights() pulls only
    the variables the path uses out of a checkpoint that also holds optimizer slots and counters."""

    def __init__(self, prefix: str, verify_data: bool = False):
        self.prefix, self.verify_data = prefix, verify_data
        self.header, self.entries = read_index(prefix)

    def __contains__(self, name):
        return name in self.entries

    def keys(self):
        return self.entries.keys()

    def __getitem__(self, name):
        if name not in self.entries:
            raise KeyError("variable %r not in checkpoint %s" % (name, self.prefix))
        return load_checkpoint(self.prefix, [name], self.verify_data)[name]


def latest_checkpoint(logdir: str) -> Optional[str]:
    """tf.train.latest_checkpoint: the model_checkpoint_path of <logdir>/checkpoint (relative paths are
    relative to logdir), or None."""
    state = os.path.join(logdir, "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state, "r") as f:
        m = re.search(r'^model_checkpoint_path:\s*"(.*)"\s*$', f.read(), re.M)
    if not m:
        return None
    path = m.group(1)
    path = path if os.path.isabs(path) else os.path.join(logdir, path)
    return path if os.path.exists(path + ".index") else None


# ----------------------------------------------------------------------------------------- writer
class _BlockBuilder:
    def __init__(self, interval):
        self.interval, self.buf, self.restarts, self.count, self.last = interval, bytearray(), [0], 0, b""

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _emit_block(out: bytearray, contents: bytes) -> bytes:
    """Appends block + trailer (no compression), returns its BlockHandle encoding."""
    handle = _put_varint(len(out)) + _put_varint(len(contents))
    out += contents + b"\x00" + struct.pack("<I", mask_crc(crc32c(contents + b"\x00")))
    return handle


def save_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], block_size: int = 4096, update_state: bool = True):
    """Writes {name: ndarray} as a one-shard V2 bundle (tf.train.Saver.save) and, like the Saver, the
    `checkpoint` state file beside it."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = []
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            a = np.asarray(tensors[name])
            dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
            if np.dtype(dt) not in _DTYPE_CODES:
                raise CheckpointError("%s: dtype %s has no TensorFlow enum here" % (name, a.dtype))
            raw = np.ascontiguousarray(a, dtype=dt)
            f.write(raw.tobytes())
            items.append((name.encode("utf-8"), _encode_entry(_DTYPE_CODES[np.dtype(dt)], a.shape, offset, raw.nbytes,
                                                              mask_crc(crc32c(raw)))))
            offset += raw.nbytes
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"         # num_shards=1, LITTLE (default), version{producer=1}
    out = bytearray()
    index = _BlockBuilder(1)
    blk = _BlockBuilder(RESTART_INTERVAL)
    for key, value in [(b"", header)] + items:
        blk.add(key, value)
        if len(blk.buf) >= block_size:
            index.add(key, _emit_block(out, blk.finish()))   # index key >= last key of the block
            blk = _BlockBuilder(RESTART_INTERVAL)
    if blk.buf:
        index.add(blk.last, _emit_block(out, blk.finish()))
    meta_handle = _emit_block(out, _BlockBuilder(RESTART_INTERVAL).finish())
    index_handle = _emit_block(out, index.finish())
    footer = meta_handle + index_handle
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    if update_state:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
