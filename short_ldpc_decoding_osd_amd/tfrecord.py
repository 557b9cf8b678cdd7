"""TFRecord + tf.train.Example codec without TensorFlow: the on-disk format either side of the
decoding path.

Files written by the reference's untouched TF stages (``Testing_data_gen_128/Main_test.py:68-106``,
``Ldpc_128_testing/data_generating.py:8-26``) are read record by record; files written here are
readable by ``tf.data.TFRecordDataset`` + ``read_TFdata.parse_exmp`` (read_TFdata.py:10-16).

Record framing (TFRecord):  u64 length | u32 masked_crc32c(length) | bytes | u32 masked_crc32c(bytes)
Example schema used by the reference: features {'feature': FloatList[n], 'label': Int64List[n],
'shape': Int64List[1]} (data_generating.py:8-14).
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import _lib

_MASK_DELTA = 0xA282EAD8


def masked_crc(buf: bytes) -> int:
    crc = _lib.load().ldpc_crc32c(buf, len(buf))
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---------------------------------------------------------------------------- protobuf wire helpers
def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _len_field(field_no: int, payload: bytes) -> bytes:
    return _varint((field_no << 3) | 2) + _varint(len(payload)) + payload


def encode_example(feature, label) -> bytes:
    """Serialised tf.train.Example {'feature': FloatList, 'label': Int64List, 'shape': Int64List}
    (get_tfrecords_example, data_generating.py:8-14).  Map entries are written in key order."""
    feature = np.ascontiguousarray(feature, dtype="<f4")
    label = np.asarray(label, dtype=np.int64)
    float_list = _len_field(1, feature.tobytes())                                   # FloatList.value, packed
    int_list = _len_field(1, b"".join(_varint(int(v)) for v in label))              # Int64List.value, packed
    shape_list = _len_field(1, b"".join(_varint(int(v)) for v in feature.shape))
    entries = [("feature", _len_field(2, float_list)), ("label", _len_field(3, int_list)),
               ("shape", _len_field(3, shape_list))]
    features = b"".join(_len_field(1, _len_field(1, k.encode()) + _len_field(2, v)) for k, v in entries)
    return _len_field(1, features)


def _parse_fields(buf):
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _read_varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 2:
            ln, pos = _read_varint(buf, pos)
            yield fno, wt, buf[pos:pos + ln]
            pos += ln
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            yield fno, wt, v
        elif wt == 5:
            yield fno, wt, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield fno, wt, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def decode_example(buf: bytes) -> dict:
    """-> {name: np.ndarray}: float32 for FloatList, int64 for Int64List (packed or not)."""
    out = {}
    for fno, _, features in _parse_fields(buf):
        if fno != 1:
            continue
        for eno, _, entry in _parse_fields(features):
            if eno != 1:
                continue
            key, feat = None, None
            for kno, _, val in _parse_fields(entry):
                if kno == 1:
                    key = bytes(val).decode()
                elif kno == 2:
                    feat = val
            if key is None or feat is None:
                continue
            for kind, _, lst in _parse_fields(feat):
                if kind == 2:      # FloatList
                    vals = []
                    for vno, wt, v in _parse_fields(lst):
                        if vno == 1 and wt == 2:
                            vals.append(np.frombuffer(bytes(v), dtype="<f4"))
                        elif vno == 1 and wt == 5:
                            vals.append(np.frombuffer(bytes(v), dtype="<f4"))
                    out[key] = np.concatenate(vals) if vals else np.zeros(0, np.float32)
                elif kind == 3:    # Int64List
                    vals = []
                    for vno, wt, v in _parse_fields(lst):
                        if vno == 1 and wt == 2:
                            p, b = 0, bytes(v)
                            while p < len(b):
                                x, p = _read_varint(b, p)
                                vals.append(_signed64(x))
                        elif vno == 1 and wt == 0:
                            vals.append(_signed64(v))
                    out[key] = np.array(vals, dtype=np.int64)
    return out


# ---------------------------------------------------------------------------- record framing
class TFRecordWriter:
    def __init__(self, path):
        self._fh = open(path, "wb")

    def write(self, payload: bytes):
        head = struct.pack("<Q", len(payload))
        self._fh.write(head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload)))

    def close(self):
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_records(path, verify=True):
    """Yield the payload of every record; checksum mismatches raise (as TF's reader does)."""
    with open(path, "rb") as fh:
        while True:
            head = fh.read(8)
            if not head:
                return
            if len(head) < 8:
                raise IOError(f"{path}: truncated record header")
            (ln,) = struct.unpack("<Q", head)
            (hcrc,) = struct.unpack("<I", fh.read(4))
            if verify and hcrc != masked_crc(head):
                raise IOError(f"{path}: corrupted record length")
            payload = fh.read(ln)
            crc_raw = fh.read(4)
            if len(payload) < ln or len(crc_raw) < 4:
                raise IOError(f"{path}: truncated record")
            if verify and struct.unpack("<I", crc_raw)[0] != masked_crc(payload):
                raise IOError(f"{path}: corrupted record data")
            yield payload


class RecordDataset:
    """The slice of ``tf.data`` the reference uses: ``batch`` (read_TFdata.py:27), ``take`` and
    ``as_numpy_iterator`` yielding ``(feature [b,n] f32, label [b,n] i64, shape [b] i32)``."""

    def __init__(self, path, code_length, batch_size=1, limit=None):
        self.path, self.code_length, self.batch_size, self.limit = path, code_length, batch_size, limit

    def batch(self, batch_size, drop_remainder=False):
        return RecordDataset(self.path, self.code_length, batch_size, self.limit)

    def take(self, count):
        return RecordDataset(self.path, self.code_length, self.batch_size, count)

    def prefetch(self, buffer_size=None):
        return self

    def cache(self):
        return self

    def as_numpy_iterator(self):
        feats, labs, shapes, batches = [], [], [], 0
        for payload in read_records(self.path):
            ex = decode_example(payload)
            f, lab = ex["feature"], ex["label"]
            if f.shape[0] != self.code_length or lab.shape[0] != self.code_length:
                raise ValueError(f"{self.path}: record with {f.shape[0]}/{lab.shape[0]} values, expected {self.code_length}")
            feats.append(f)
            labs.append(lab)
            shapes.append(np.int32(ex["shape"][0]) if "shape" in ex and len(ex["shape"]) else np.int32(self.code_length))
            if len(feats) == self.batch_size:
                yield np.stack(feats), np.stack(labs), np.array(shapes, dtype=np.int32)
                feats, labs, shapes = [], [], []
                batches += 1
                if self.limit is not None and batches >= self.limit:
                    return
        if feats:
            yield np.stack(feats), np.stack(labs), np.array(shapes, dtype=np.int32)

    def __iter__(self):
        return self.as_numpy_iterator()
