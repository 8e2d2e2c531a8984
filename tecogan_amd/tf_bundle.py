"""Reader / writer of TensorFlow checkpoint "tensor bundles" (`<prefix>.index` + `<prefix>.data-0000k-of-0000N`).

SURVEY.md section 8(f) item 1: the reference saves and restores its models with `tf.train.Saver`
(reference main.py:224,245,307,324,342,365,420; lib/ops.py:370-391 reads shapes through
`tf.train.NewCheckpointReader`), i.e. in this format.  Pure Python + numpy, no TensorFlow.

Format restated from the TensorFlow sources (tensorflow/core/util/tensor_bundle/, tensorflow/core/lib/io/table*,
protobuf tensor_bundle.proto) -- **unpinned against a file written by real TensorFlow** (none is available offline):
the tests pin the CRC, varint, block and proto encodings with hand-built bytes and round trips.

  .index   a LevelDB-style sorted string table:
             data blocks | metaindex block | index block | 48-byte footer
           block   = entries, restart offsets (uint32 LE each), restart count (uint32 LE); followed by a 5-byte trailer:
                     compression type (0 none, 1 snappy) + masked CRC32C of (block bytes + type byte)
           entry   = varint shared-key-bytes, varint unshared-key-bytes, varint value-bytes, key suffix, value
           footer  = metaindex handle, index handle (each: varint offset, varint size), zero padding to 40 bytes,
                     magic 0xdb4775248b80fb57 (uint64 LE)
           keys    "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}
                   variable name -> BundleEntryProto {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size,
                                                      6: masked crc32c (fixed32), 7: slices (partitioned variables)}
  .data-*  raw little-endian tensor bytes at the recorded offsets.
"""
import os
import struct
from collections import OrderedDict

import numpy as np

MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DT_TO_NP = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
            17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
NP_TO_DT = {np.dtype(v): k for k, v in DT_TO_NP.items()}
DT_BFLOAT16 = 14


# ---- CRC32C (Castagnoli), table driven ----------------------------------------------------------------------------
def _make_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_TABLE = _make_table()
_TABLE_NP = np.array(_TABLE, dtype=np.uint32)


def _crc_raw_small(data, reg):
    t = _TABLE
    for b in data:
        reg = t[(reg ^ b) & 0xFF] ^ (reg >> 8)
    return reg


# CRC registers are linear over GF(2): "append n zero bytes" is a 32x32 bit matrix (32 uint32 columns), and
# R(A || B, r) = Z_len(B)(R(A, r)) ^ R(B, 0).  That lets numpy run many chunks of a large tensor in lock step.
def _op_apply(op, x):
    r, i = 0, 0
    while x:
        if x & 1:
            r ^= op[i]
        x >>= 1
        i += 1
    return r


def _op_mul(a, b):
    return [_op_apply(a, col) for col in b]


def _zero_bytes_op(n):
    """Operator of n zero bytes on the raw register."""
    result = [1 << i for i in range(32)]
    base = [_TABLE[(1 << i) & 0xFF] ^ ((1 << i) >> 8) for i in range(32)]       # one zero byte
    while n:
        if n & 1:
            result = _op_mul(base, result)
        base = _op_mul(base, base)
        n >>= 1
    return result


def crc32c(data, crc=0):
    """CRC32C (Castagnoli) of bytes-like `data`, continuing from `crc` (check value: crc32c(b"123456789") == 0xE3069283).
    Large inputs (checkpoint tensors) are processed as 2048 interleaved chunks with numpy and recombined."""
    data = bytes(data)
    n = len(data)
    reg0 = crc ^ 0xFFFFFFFF
    if n < (1 << 16):
        return _crc_raw_small(data, reg0) ^ 0xFFFFFFFF
    lanes = 2048
    L = (n + lanes - 1) // lanes
    buf = np.zeros(lanes * L, dtype=np.uint8)
    buf[lanes * L - n:] = np.frombuffer(data, dtype=np.uint8)       # zero padding IN FRONT: R(zeros, 0) == 0
    cols = buf.reshape(lanes, L)
    reg = np.zeros(lanes, dtype=np.uint32)
    tab = _TABLE_NP
    for i in range(L):
        reg = tab[(reg ^ cols[:, i]) & 0xFF] ^ (reg >> 8)
    zl = _zero_bytes_op(L)
    ztab = [[_op_apply(zl, b << (8 * k)) for b in range(256)] for k in range(4)]  # byte-wise tables of Z_L
    r = 0
    for v in reg.tolist():
        r = ztab[0][r & 0xFF] ^ ztab[1][(r >> 8) & 0xFF] ^ ztab[2][(r >> 16) & 0xFF] ^ ztab[3][r >> 24] ^ v
    r ^= _op_apply(_zero_bytes_op(n), reg0)                           # contribution of the initial register
    return r ^ 0xFFFFFFFF


def mask_crc(crc):
    """leveldb/TF crc masking: rotate right by 15 and add a constant (a CRC of data that embeds CRCs stays well-behaved)."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---- varints / minimal protobuf -------------------------------------------------------------------------------------
def _put_varint(v):
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _proto_fields(buf):
    """Yield (field_number, wire_type, value) of one serialized message; length-delimited values are zero-copy
    memoryview slices (a VGG fc6 entry is 400 MB: nothing is copied until a wanted tensor is materialised)."""
    buf = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _field(field, wt, payload):
    return _put_varint((field << 3) | wt) + payload


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _encode_shape(shape):
    return b"".join(_field(2, 2, _put_varint(len(d)) + d) for d in (_field(1, 0, _put_varint(int(s))) for s in shape))


def _decode_shape(buf):
    dims = []
    for f, _, v in _proto_fields(buf):
        if f == 2:
            size = 0
            for f2, _, v2 in _proto_fields(v):
                if f2 == 1:
                    size = _signed64(v2)
            dims.append(size)
        elif f == 3 and v:
            raise ValueError("tensor of unknown rank in checkpoint")
    return tuple(dims)


# ---- snappy (raw format) decompression: index blocks may be compressed ----------------------------------------------
def _snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                             # byte-wise: copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# ---- sorted string table ----------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
    contents = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack_from("<I", data, offset + size + 1)[0]
        if unmask_crc(stored) != crc32c(data[offset:offset + size + 1]):
            raise ValueError("index block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        contents = _snappy_decompress(contents)
    elif ctype != 0:
        raise ValueError("unknown block compression type %d" % ctype)
    return contents


def _block_entries(block):
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 * (nrestart + 1)
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _build_block(items, restart_interval):
    out, restarts, prev, count = bytearray(), [], b"", 0
    for key, value in items:
        shared = 0
        if count % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(prev), len(key))
            while shared < m and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        prev, count = key, count + 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _handle(offset, size):
    return _put_varint(offset) + _put_varint(size)


class BundleReader:
    """`BundleReader(prefix)`: names/shapes/dtypes from the index; `get(name)` reads one tensor from the data shard."""

    def __init__(self, prefix, verify_index=True):
        self.prefix = prefix
        with open(prefix + ".index", "rb") as fh:
            data = fh.read()
        if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != MAGIC:
            raise ValueError("%s.index is not a TensorFlow checkpoint index (bad magic)" % prefix)
        foot = data[len(data) - 48:]
        _, pos = _get_varint(foot, 0)                     # metaindex handle (unused: no filter / properties)
        _, pos = _get_varint(foot, pos)
        ioff, pos = _get_varint(foot, pos)
        isize, pos = _get_varint(foot, pos)
        self.entries, self.header = OrderedDict(), None
        for _, hv in _block_entries(_read_block(data, ioff, isize, verify_index)):
            boff, p = _get_varint(hv, 0)
            bsize, _ = _get_varint(hv, p)
            for key, value in _block_entries(_read_block(data, boff, bsize, verify_index)):
                if key == b"":
                    self.header = self._decode_header(value)
                else:
                    self.entries[key.decode("utf-8")] = self._decode_entry(value)
        if self.header is None:
            raise ValueError("%s.index has no bundle header" % prefix)
        if self.header["endianness"] != 0:
            raise ValueError("big-endian bundles are not supported")

    @staticmethod
    def _decode_header(buf):
        h = {"num_shards": 0, "endianness": 0, "producer": 0}
        for f, _, v in _proto_fields(buf):
            if f == 1:
                h["num_shards"] = v
            elif f == 2:
                h["endianness"] = v
            elif f == 3:
                for f2, _, v2 in _proto_fields(v):
                    if f2 == 1:
                        h["producer"] = v2
        return h

    @staticmethod
    def _decode_entry(buf):
        e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
        for f, _, v in _proto_fields(buf):
            if f == 1:
                e["dtype"] = v
            elif f == 2:
                e["shape"] = _decode_shape(v)
            elif f == 3:
                e["shard_id"] = v
            elif f == 4:
                e["offset"] = v
            elif f == 5:
                e["size"] = v
            elif f == 6:
                e["crc32c"] = v
            elif f == 7:
                e["sliced"] = True
        return e

    def keys(self):
        return list(self.entries)

    def shape(self, name):
        return self.entries[name]["shape"]

    def _shard_path(self, shard):
        return "%s.data-%05d-of-%05d" % (self.prefix, shard, self.header["num_shards"])

    def get(self, name, verify=False):
        e = self.entries[name]
        if e["sliced"]:
            raise NotImplementedError("partitioned variable %s (tensor slices) is not supported" % name)
        with open(self._shard_path(e["shard_id"]), "rb") as fh:
            fh.seek(e["offset"])
            raw = fh.read(e["size"])
        if len(raw) != e["size"]:
            raise ValueError("data shard truncated while reading %s" % name)
        if verify and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("checksum mismatch for tensor %s" % name)
        if e["dtype"] == DT_BFLOAT16:                      # widen to float32 (exact)
            a = (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
        elif e["dtype"] in DT_TO_NP:
            a = np.frombuffer(raw, dtype=np.dtype(DT_TO_NP[e["dtype"]]).newbyteorder("<"))
        else:
            raise NotImplementedError("tensor %s has unsupported dtype enum %d" % (name, e["dtype"]))
        n = 1
        for d in e["shape"]:
            n *= d
        if a.size != n:
            raise ValueError("tensor %s: %d elements in the data file, shape %s" % (name, a.size, e["shape"]))
        return a.reshape(e["shape"]).copy()


def is_bundle(prefix):
    return os.path.exists(prefix + ".index")


def read_bundle(prefix, names=None, verify=False):
    """name -> numpy array for every (or the listed) tensor of the checkpoint `prefix`."""
    r = BundleReader(prefix)
    return OrderedDict((n, r.get(n, verify)) for n in (r.keys() if names is None else names))


def write_bundle(prefix, tensors, block_size=4096, with_crc=True):
    """Write `tensors` (mapping name -> array-like) as a single-shard bundle `prefix`.index / .data-00000-of-00001."""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    items, offset = [], 0
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open("%s.data-00000-of-00001" % prefix, "wb") as fh:
        for name in names:
            a = np.asarray(tensors[name])
            if not a.flags.c_contiguous:                  # (np.ascontiguousarray would turn 0-d scalars into shape (1,))
                a = a.copy(order="C")
            if a.dtype not in NP_TO_DT:
                raise TypeError("tensor %s: dtype %s has no TensorFlow enum here" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            fh.write(raw)
            entry = _field(1, 0, _put_varint(NP_TO_DT[a.dtype]))
            shp = _encode_shape(a.shape)
            entry += _field(2, 2, _put_varint(len(shp)) + shp)
            if offset:
                entry += _field(4, 0, _put_varint(offset))
            entry += _field(5, 0, _put_varint(len(raw)))
            if with_crc:
                entry += _field(6, 5, struct.pack("<I", mask_crc(crc32c(raw))))
            items.append((name.encode("utf-8"), entry))
            offset += len(raw)
    version = _field(1, 0, _put_varint(1))                                     # VersionDef.producer = 1
    header = _field(1, 0, _put_varint(1)) + _field(3, 2, _put_varint(len(version)) + version)   # num_shards = 1, LITTLE
    items.insert(0, (b"", header))
    out, index_items = bytearray(), []

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                                          # no compression
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return off, len(block)

    cur, cur_bytes = [], 0
    for kv in items:
        cur.append(kv)
        cur_bytes += len(kv[0]) + len(kv[1]) + 6
        if cur_bytes >= block_size:
            off, size = emit(_build_block(cur, 16))
            index_items.append((cur[-1][0], _handle(off, size)))
            cur, cur_bytes = [], 0
    if cur:
        off, size = emit(_build_block(cur, 16))
        index_items.append((cur[-1][0], _handle(off, size)))
    moff, msize = emit(_build_block([], 1))                                    # empty metaindex
    ioff, isize = emit(_build_block(index_items, 1))
    foot = _handle(moff, msize) + _handle(ioff, isize)
    out.extend(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC))
    with open(prefix + ".index", "wb") as fh:
        fh.write(bytes(out))
    return prefix


# ---- TensorFlow V1 ("tensor slice") checkpoints: slim's vgg_19.ckpt ---------------------------------------------------
# One sorted string table (blocks usually snappy-compressed, tensorflow/core/util/tensor_slice_writer.cc):
#   key ""  -> SavedTensorSlices{1: meta = SavedTensorSliceMeta{1: repeated SavedSliceMeta{1: name, 2: shape, 3: type,
#              4: repeated TensorSliceProto}}}
#   other   -> SavedTensorSlices{2: data = SavedSlice{1: name, 2: TensorSliceProto, 3: TensorProto}}
#              TensorProto{1: dtype, 2: shape, 4: tensor_content, 5: packed float_val, 6: double_val, 7: int_val,
#                          10: int64_val}; TensorSliceProto{1: repeated Extent{1: start, 2: length}} (empty = whole dim)
# The reader walks the data entries and uses the name stored INSIDE each value, so it does not depend on the
# ordered-code key encoding.  Restated from the TensorFlow sources; unpinned against a TF-written file.
def is_v1_checkpoint(path):
    if not os.path.isfile(path) or os.path.getsize(path) < 48:
        return False
    with open(path, "rb") as fh:
        fh.seek(-8, os.SEEK_END)
        return struct.unpack("<Q", fh.read(8))[0] == MAGIC


def _tensor_from_proto(buf, shape, dtype_enum):
    content, floats, doubles, ints, int64s = None, [], [], [], []
    for f, wt, v in _proto_fields(buf):
        if f == 4:
            content = v
        elif f == 5:
            floats.append(np.frombuffer(v, "<f4") if wt == 2 else np.array([struct.unpack("<f", struct.pack("<I", v))[0]], "<f4"))
        elif f == 6:
            doubles.append(np.frombuffer(v, "<f8") if wt == 2 else np.array([struct.unpack("<d", struct.pack("<Q", v))[0]], "<f8"))
        elif f == 7:
            if wt == 2:
                pos, vals = 0, []
                while pos < len(v):
                    x, pos = _get_varint(v, pos)
                    vals.append(_signed64(x))
                ints.append(np.array(vals, np.int32))
            else:
                ints.append(np.array([_signed64(v)], np.int32))
        elif f == 10:
            if wt == 2:
                pos, vals = 0, []
                while pos < len(v):
                    x, pos = _get_varint(v, pos)
                    vals.append(_signed64(x))
                int64s.append(np.array(vals, np.int64))
            else:
                int64s.append(np.array([_signed64(v)], np.int64))
    if content is not None and len(content):
        a = np.frombuffer(content, np.dtype(DT_TO_NP[dtype_enum]).newbyteorder("<"))
    else:
        parts = floats or doubles or ints or int64s
        a = np.concatenate(parts) if parts else np.zeros(0, DT_TO_NP.get(dtype_enum, np.float32))
    n = 1
    for d in shape:
        n *= d
    if a.size == 1 and n > 1:                              # TensorProto may store a constant tensor as one value
        a = np.full(n, a[0], a.dtype)
    if a.size != n:
        raise ValueError("V1 checkpoint: tensor with %d values for shape %s" % (a.size, shape))
    return a.reshape(shape).copy()


def read_v1_checkpoint(path, names=None, verify=True):
    """name -> numpy array for every (or the listed) whole-tensor entry of a V1 checkpoint file."""
    with open(path, "rb") as fh:
        data = fh.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != MAGIC:
        raise ValueError("%s is not a TensorFlow V1 checkpoint (bad magic)" % path)
    foot = data[len(data) - 48:]
    _, pos = _get_varint(foot, 0)
    _, pos = _get_varint(foot, pos)
    ioff, pos = _get_varint(foot, pos)
    isize, pos = _get_varint(foot, pos)
    want = None if names is None else set(names)
    meta, out = {}, OrderedDict()
    for _, hv in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _get_varint(hv, 0)
        bsize, _ = _get_varint(hv, p)
        for key, value in _block_entries(_read_block(data, boff, bsize, verify)):
            for f, _, v in _proto_fields(value):
                if f == 1 and key == b"":                  # meta
                    for f2, _, v2 in _proto_fields(v):
                        if f2 == 1:
                            name, shape, dt = None, (), 1
                            for f3, _, v3 in _proto_fields(v2):
                                if f3 == 1:
                                    name = bytes(v3).decode("utf-8")
                                elif f3 == 2:
                                    shape = _decode_shape(v3)
                                elif f3 == 3:
                                    dt = v3
                            meta[name] = (shape, dt)
                elif f == 2:                               # data: SavedSlice
                    name, tproto, partial = None, None, False
                    for f2, _, v2 in _proto_fields(v):
                        if f2 == 1:
                            name = bytes(v2).decode("utf-8")
                        elif f2 == 2:
                            for f3, _, v3 in _proto_fields(v2):          # any Extent with a start/length = a partial slice
                                if f3 == 1 and len(v3):
                                    partial = True
                        elif f2 == 3:
                            tproto = v2
                    if name is None or tproto is None or (want is not None and name not in want):
                        continue
                    if partial:
                        raise NotImplementedError("V1 checkpoint: %s is stored in partial slices" % name)
                    shape, dt = meta.get(name, (None, 1))
                    if shape is None:
                        raise ValueError("V1 checkpoint: data for %s precedes its metadata" % name)
                    out[name] = _tensor_from_proto(tproto, shape, dt)
    return out
