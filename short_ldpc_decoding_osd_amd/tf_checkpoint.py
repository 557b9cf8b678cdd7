"""TensorFlow checkpoint bundle (``<prefix>.index`` + ``<prefix>.data-00000-of-00001``) without TensorFlow
(SURVEY.md 8(f) N3 ii): how the reference's testing stage receives the learned NMS weight
(``ldpc_128_testing.py:57-68``: ``tf.train.Checkpoint(myAwesomeModel=model).restore(latest)``; written by
``Ldpc_128_training/ms_decoder_dense.py`` through a ``CheckpointManager``).

Format, as published in TensorFlow's sources (tensor_bundle.proto, core/util/tensor_bundle, core/lib/io/table*):
  * ``.index`` is a LevelDB-style sorted table: data blocks of prefix-compressed (key, value) entries with a
    restart array, each block followed by a 1-byte compression tag and a masked CRC-32C; an index block of
    (separator key -> block handle); a 48-byte footer (metaindex handle, index handle, padding, magic
    0xdb4775248b80fb57).  Bundles are written uncompressed.
  * key ``""`` -> BundleHeaderProto {num_shards = 1, endianness, version}; every other key is a tensor name
    -> BundleEntryProto {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6 (masked)}.
  * tensor bytes sit at ``offset`` of shard ``shard_id`` (``.data-<id>-of-<n>``), little endian.
  * object-based checkpoints name a variable ``<object path>/.ATTRIBUTES/VARIABLE_VALUE``; for the NMS-1
    weight that is ``myAwesomeModel/layer/shared_check_weight/.ATTRIBUTES/VARIABLE_VALUE`` (f32[1]).
  * ``checkpoint`` next to the files is a text proto whose ``model_checkpoint_path`` names the latest prefix.

PARITY UNPINNED: the reference ships no checkpoint and TensorFlow is not installed in the build image, so
this module is tested against files written by its own ``write_checkpoint`` and against tables a test builds
byte by byte the way TensorFlow's table builder lays them out (prefix-compressed keys with restart interval 16,
several data blocks, two data shards, a compressed block that must be refused) -- never against a
TensorFlow-written file.  ``weights.load_values_txt`` -- the text mirror
the training stage writes next to every checkpoint -- is the tested route.
"""
from __future__ import annotations

import os
import re
import struct

import numpy as np

from .tfrecord import _len_field, _parse_fields, _read_varint, _signed64, _varint, masked_crc

TABLE_MAGIC = 0xDB4775248B80FB57
VARIABLE_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
NMS_WEIGHT_KEYS = (          # add_weight names of Decoder_Layer.build (ms_test.py:82-91) as object-graph paths
    ("shared_check_weight", "shared_check_weight"), ("shared_bit_weight1", "shared_bit_weight1"),
    ("shared_bit_weight2", "shared_bit_weight2"), ("shared_bit_weight", "shared_bit_weight"))

# tensorflow DataType enum -> numpy (the numeric types a Keras variable can have)
_DTYPES = {1: "<f4", 2: "<f8", 3: "<i4", 4: "u1", 5: "<i2", 6: "i1", 9: "<i8", 10: "?", 17: "<u2", 19: "<f2",
           22: "<u4", 23: "<u8"}
_DTYPE_IDS = {np.dtype(v).str: k for k, v in _DTYPES.items()}


# ---------------------------------------------------------------------------- table reading
def _block(buf, offset, size):
    data, tag = buf[offset:offset + size], buf[offset + size]
    (crc,) = struct.unpack_from("<I", buf, offset + size + 1)
    if masked_crc(bytes(buf[offset:offset + size + 1])) != crc:
        raise ValueError("checkpoint index: block checksum mismatch")
    if tag != 0:
        raise ValueError("checkpoint index: compressed block (bundles are written uncompressed)")
    return data


def _entries(block):
    (nrestart,) = struct.unpack_from("<I", block, len(block) - 4)
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _handle(buf, pos=0):
    off, pos = _read_varint(buf, pos)
    size, pos = _read_varint(buf, pos)
    return off, size, pos


def read_index(path):
    """``<prefix>.index`` -> (header dict, {tensor name: entry dict})."""
    buf = memoryview(open(path, "rb").read())
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorFlow checkpoint index (bad table magic)")
    footer = buf[len(buf) - 48:]
    _, _, pos = _handle(footer)                        # metaindex handle (unused by bundles)
    ioff, isize, _ = _handle(footer, pos)
    header, entries = {}, {}
    for _, handle in _entries(_block(buf, ioff, isize)):
        boff, bsize, _ = _handle(handle)
        for key, value in _entries(_block(buf, boff, bsize)):
            fields = {}
            for fno, wt, val in _parse_fields(value):
                fields.setdefault(fno, []).append(val)
            if key == b"":
                header = dict(num_shards=fields.get(1, [1])[0], endianness=fields.get(2, [0])[0])
                continue
            shape = []
            for shp in fields.get(2, []):
                for fno, _, dim in _parse_fields(shp):
                    if fno == 2:
                        shape.append(_signed64(dict((f, v) for f, _, v in _parse_fields(dim)).get(1, 0)))
            crc = struct.unpack("<I", bytes(fields[6][0]))[0] if 6 in fields else None
            entries[key.decode()] = dict(dtype=fields.get(1, [0])[0], shape=tuple(shape), shard_id=fields.get(3, [0])[0],
                                         offset=fields.get(4, [0])[0], size=fields.get(5, [0])[0], crc32c=crc,
                                         sliced=7 in fields)
    return header, entries


def read_checkpoint(prefix, verify=True):
    """All numeric tensors of the bundle ``prefix`` -> {name: ndarray}.  String tensors (the object graph)
    and sliced entries are skipped."""
    header, entries = read_index(prefix + ".index")
    if header.get("endianness", 0) != 0:
        raise ValueError(f"{prefix}: big-endian bundle")
    nshards = int(header.get("num_shards", 1))
    shards, out = {}, {}
    for name, e in entries.items():
        if e["dtype"] not in _DTYPES or e["sliced"]:
            continue
        sid = int(e["shard_id"])
        if sid not in shards:
            shards[sid] = np.memmap(f"{prefix}.data-{sid:05d}-of-{nshards:05d}", dtype=np.uint8, mode="r")
        raw = bytes(shards[sid][e["offset"]: e["offset"] + e["size"]])
        if len(raw) != e["size"]:
            raise ValueError(f"{prefix}: tensor '{name}' runs past the end of its data shard")
        if verify and e["crc32c"] is not None and masked_crc(raw) != e["crc32c"]:
            raise ValueError(f"{prefix}: checksum mismatch in tensor '{name}'")
        out[name] = np.frombuffer(raw, dtype=_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    return out


def latest_checkpoint(directory):
    """``tf.train.latest_checkpoint``: the prefix named by ``model_checkpoint_path`` in ``<dir>/checkpoint``."""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    m = re.search(r'^model_checkpoint_path:\s*"([^"]+)"', open(state, "rt").read(), re.M)
    if not m:
        return None
    p = m.group(1)
    return p if os.path.isabs(p) else os.path.join(directory, p)


def load_checkpoint(model, path):
    """Restore ``Decoding_model`` weights from a checkpoint directory (its latest checkpoint) or a prefix;
    returns the names restored.  What ldpc_128_testing.py:57-68 does through tf.train.Checkpoint."""
    prefix = latest_checkpoint(path) if os.path.isdir(path) else path
    if prefix is None:
        raise FileNotFoundError(f"{path}: no 'checkpoint' state file with a model_checkpoint_path")
    tensors = read_checkpoint(prefix)
    done = []
    for name, value in tensors.items():
        if not name.endswith(VARIABLE_SUFFIX):
            continue
        leaf = name[: -len(VARIABLE_SUFFIX)].split("/")[-1]
        for key, attr in NMS_WEIGHT_KEYS:
            if leaf == key:
                setattr(model.layer, attr, np.asarray(value, dtype=np.float32).reshape(-1)[:1].copy())
                done.append(attr)
                break
    if "shared_check_weight" not in done:
        raise KeyError(f"{prefix}: no '.../shared_check_weight{VARIABLE_SUFFIX}' variable in the bundle")
    return done


# ---------------------------------------------------------------------------- writing (tests, export)
def _emit_block(entries):
    """One table block, every entry a restart point (no prefix sharing) + trailer."""
    body, restarts = bytearray(), []
    for key, value in entries:
        restarts.append(len(body))
        body += _varint(0) + _varint(len(key)) + _varint(len(value)) + key + value
    for r in restarts or [0]:
        body += struct.pack("<I", r)
    body += struct.pack("<I", max(len(restarts), 1))
    tagged = bytes(body) + b"\x00"
    return tagged + struct.pack("<I", masked_crc(tagged)), len(body)


def write_checkpoint(prefix, tensors, step_state=True):
    """Write {name: ndarray} as a one-shard bundle (+ the ``checkpoint`` state file)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    data, rows = bytearray(), []
    header = _varint((1 << 3) | 0) + _varint(1) + _len_field(3, _varint((1 << 3) | 0) + _varint(1))   # num_shards = 1, version.producer = 1
    rows.append((b"", header))
    for name in sorted(tensors):
        a = np.asarray(tensors[name], order="C")   # (ascontiguousarray would turn a scalar into shape (1,))
        a = a.astype(a.dtype.newbyteorder("<")) if a.dtype.byteorder == ">" else a
        raw = a.tobytes()
        shape = b"".join(_len_field(2, _varint((1 << 3) | 0) + _varint(int(d))) for d in a.shape)
        entry = (_varint((1 << 3) | 0) + _varint(_DTYPE_IDS[np.dtype(a.dtype).str]) + _len_field(2, shape)
                 + _varint((4 << 3) | 0) + _varint(len(data)) + _varint((5 << 3) | 0) + _varint(len(raw))
                 + _varint((6 << 3) | 5) + struct.pack("<I", masked_crc(raw)))
        rows.append((name.encode(), entry))
        data += raw
    out = bytearray()
    block, size = _emit_block(rows)
    out += block
    meta_off = len(out)
    mblock, msize = _emit_block([])
    out += mblock
    index_off = len(out)
    iblock, isize = _emit_block([(rows[-1][0] + b"\x00", _varint(0) + _varint(size))])
    out += iblock
    footer = _varint(meta_off) + _varint(msize) + _varint(index_off) + _varint(isize)
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    open(prefix + ".index", "wb").write(bytes(out))
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    if step_state:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "wt") as f:
            f.write(f'model_checkpoint_path: "{base}"\nall_model_checkpoint_paths: "{base}"\n')
    return prefix
