"""Shared helpers of the GPU parity tests."""
import numpy as np
import torch


def pack_np(bits):
    """[B, n] 0/1 -> [B, ceil(n/64)] uint64, bit v at word v//64, position v%64."""
    bits = np.asarray(bits, dtype=np.uint8)
    B, n = bits.shape
    pad = (-n) % 64
    if pad:
        bits = np.concatenate([bits, np.zeros((B, pad), np.uint8)], axis=1)
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64)


def words_np(t):
    return t.cpu().numpy().view(np.uint64)


def to_dev(a, dec, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dec.device)
