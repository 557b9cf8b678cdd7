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


_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.arange(256, dtype=np.uint32)
        for _ in range(8):
            t = np.where(t & 1, (t >> 1) ^ np.uint32(0x82F63B78), t >> 1).astype(np.uint32)
        _CRC_TABLE = t
    return _CRC_TABLE


def masked_crc_rows(rows: np.ndarray) -> np.ndarray:
    """Masked CRC-32C of every row of a [N, L] uint8 array: the byte-wise table recurrence run over all rows at once (L steps
    of N-wide NumPy operations instead of N calls) -- the bulk reader / writer below checks and makes whole files with it."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    t = _crc_table()
    crc = np.full(rows.shape[0], 0xFFFFFFFF, dtype=np.uint32)
    cols = np.ascontiguousarray(rows.T)
    for j in range(cols.shape[0]):
        crc = t[(crc ^ cols[j]) & np.uint32(0xFF)] ^ (crc >> np.uint32(8))
    crc = ~crc
    return (((crc >> np.uint32(15)) | (crc << np.uint32(17))) + np.uint32(_MASK_DELTA)).astype(np.uint32)


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


# ---------------------------------------------------------------------------- whole files at once
# The reference's files hold ONE kind of record: n float32 values, n labels that are 0 or 1, the shape.  Such a file is a
# [N, 16 + L] byte matrix whose rows differ only in the 4 n bytes of the values, the n bytes of the labels (one-byte varints)
# and the trailing checksum -- it is written and read as that matrix (the per-record Python codec above took 76 s for the stage
# drivers on 60 000 frames, the decoders a few milliseconds).  Anything else -- labels beyond 127, records of different length,
# other keys -- goes through the record-by-record path; both give the same bytes and the same arrays (tests/test_tfrecord.py).
def _payload_layout(example: bytes, feature: np.ndarray, label: np.ndarray):
    """Offsets of the packed float values and of the (one-byte) label varints inside `example`, or None."""
    fb, lb = feature.astype("<f4").tobytes(), bytes(int(v) & 0x7F for v in label)
    if any((int(v) < 0 or int(v) > 127) for v in label):
        return None
    of, ol = example.find(fb), example.find(b"\x0a\x05label")
    if of < 0 or ol < 0 or example.find(fb, of + 1) >= 0:
        return None
    ol = example.find(lb, ol)
    if ol < 0:
        return None
    return of, len(fb), ol, len(lb)


def encode_examples_bulk(features, labels):
    """-> (records [N, 16 + L] uint8 ready to be written, or None when the rows do not share one layout)."""
    features = np.ascontiguousarray(features, dtype="<f4")
    labels = np.asarray(labels)
    N = features.shape[0]
    if N == 0 or features.ndim != 2 or labels.shape != features.shape or labels.min() < 0 or labels.max() > 127:
        return None
    # a template whose value bytes cannot be mistaken for structure: distinct floats, then the real first row is written over it
    probe_f = (np.arange(features.shape[1], dtype=np.float32) + 1000.5).astype("<f4")
    probe_l = np.zeros(features.shape[1], dtype=np.int64)
    probe_l[::2] = 1
    ex = encode_example(probe_f, probe_l)
    lay = _payload_layout(ex, probe_f, probe_l)
    if lay is None:
        return None
    of, nf, ol, nl = lay
    L = len(ex)
    rec = np.empty((N, 16 + L), dtype=np.uint8)
    head = struct.pack("<Q", L)
    rec[:, :8] = np.frombuffer(head, dtype=np.uint8)
    rec[:, 8:12] = np.frombuffer(struct.pack("<I", masked_crc(head)), dtype=np.uint8)
    rec[:, 12:12 + L] = np.frombuffer(ex, dtype=np.uint8)
    rec[:, 12 + of:12 + of + nf] = features.view(np.uint8).reshape(N, nf)
    rec[:, 12 + ol:12 + ol + nl] = labels.astype(np.uint8)
    rec[:, 12 + L:] = masked_crc_rows(rec[:, 12:12 + L]).astype("<u4").view(np.uint8).reshape(N, 4)
    return rec


def write_examples(path, features, labels):
    """The file `TFRecordWriter` + `encode_example` would write row by row, made in one piece when the rows share a layout."""
    rec = encode_examples_bulk(features, labels)
    if rec is not None:
        with open(path, "wb") as fh:
            fh.write(rec.tobytes())
        return
    with TFRecordWriter(path) as wrt:
        for inx in range(len(labels)):
            wrt.write(encode_example(features[inx], labels[inx]))


def read_examples_bulk(path, code_length, verify=True):
    """-> (features [N, n] f32, labels [N, n] i64, shapes [N] i32) of a file whose records all have the layout of its first
    one with one-byte labels, or None (the caller then reads record by record).  Checksums are verified for every record."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < 16:
        return None
    L = int(raw[:8].view("<u8")[0])
    stride = 16 + L
    if L <= 0 or raw.size % stride:
        return None
    N = raw.size // stride
    rec = raw.reshape(N, stride)
    if (rec[:, :12] != rec[0, :12]).any():
        return None
    first = bytes(rec[0, 12:12 + L])
    try:
        ex0 = decode_example(first)
    except Exception:
        return None
    if set(ex0) - {"feature", "label", "shape"} or "feature" not in ex0 or "label" not in ex0:
        return None
    f0, l0 = ex0["feature"], ex0["label"]
    if f0.shape[0] != code_length or l0.shape[0] != code_length:
        return None
    # the layout of the first record, found by re-encoding it: the canonical writer's bytes must be what the file holds
    if encode_example(f0, l0) != first:
        return None
    probe = _payload_layout(first, f0, l0)
    if probe is None:      # (values that repeat inside the record: locate through a probe record of the same shape instead)
        pf = (np.arange(code_length, dtype=np.float32) + 1000.5).astype("<f4")
        pl = np.zeros(code_length, dtype=np.int64); pl[::2] = 1
        probe = _payload_layout(encode_example(pf, pl), pf, pl)
        if probe is None:
            return None
    of, nf, ol, nl = probe
    body = rec[:, 12:12 + L]
    mask = np.ones(L, dtype=bool)
    mask[of:of + nf] = False
    mask[ol:ol + nl] = False
    if (body[:, mask] != body[0, mask]).any():
        return None                                   # some record is laid out differently
    lab_bytes = body[:, ol:ol + nl]
    if (lab_bytes & 0x80).any():
        return None
    if verify:
        head = bytes(rec[0, :8])
        if struct.unpack("<I", bytes(rec[0, 8:12]))[0] != masked_crc(head):
            raise IOError(f"{path}: corrupted record length")
        want = np.ascontiguousarray(rec[:, 12 + L:]).view("<u4").reshape(N)
        if (masked_crc_rows(body) != want).any():
            raise IOError(f"{path}: corrupted record data")
    feats = np.ascontiguousarray(body[:, of:of + nf]).view("<f4").reshape(N, code_length).astype(np.float32, copy=False)
    labs = lab_bytes.astype(np.int64)
    shapes = np.full(N, int(ex0["shape"][0]) if "shape" in ex0 and len(ex0["shape"]) else code_length, dtype=np.int32)
    return feats, labs, shapes


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
        bulk = read_examples_bulk(self.path, self.code_length)
        if bulk is not None:
            F, Lb, Sh = bulk
            nb = -(-F.shape[0] // self.batch_size)
            if self.limit is not None:
                nb = min(nb, self.limit)
            for b in range(nb):
                sl = slice(b * self.batch_size, (b + 1) * self.batch_size)
                yield F[sl], Lb[sl], Sh[sl]
            return
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
