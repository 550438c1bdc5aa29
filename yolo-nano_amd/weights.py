"""Deterministic synthetic weights for YOLO-Nano.

There are no trained checkpoints in this environment
(``/root/reference/.MISSING_LARGE_BLOBS``), so every test, fixture and bench
uses weights drawn from ``numpy.random.RandomState`` streams keyed by the
state-dict key (crc32) — *not* torch's RNG, whose stream is version specific.
BatchNorm statistics are randomised on purpose: the reference initialiser
(``backbone/shufflenetv2.py:141-145``) makes BN folding a near no-op, which
would hide folding bugs.
"""
import zlib

import numpy as np

from . import arch


def _rs(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)


def make_state_dict(backbone="1.0x", num_classes=20, num_anchors=3, seed=0):
    """-> {key: np.ndarray} with the reference's key set and shapes."""
    sd = {}
    for sp in arch.conv_specs(backbone, num_classes, num_anchors):
        gain = 2.0 if sp.act != arch.ACT_NONE else 1.0
        std = np.sqrt(gain / sp.fan_in)
        k = sp.conv + ".weight"
        sd[k] = (_rs(k, seed).standard_normal(sp.weight_shape) * std).astype(np.float32)
        if sp.has_bias:
            k = sp.conv + ".bias"
            sd[k] = (_rs(k, seed).standard_normal((sp.cout,)) * 0.1).astype(np.float32)
        if sp.bn is not None:
            c = (sp.cout,)
            k = sp.bn + ".weight"
            sd[k] = _rs(k, seed).uniform(0.7, 1.3, c).astype(np.float32)
            k = sp.bn + ".bias"
            sd[k] = _rs(k, seed).uniform(-0.2, 0.2, c).astype(np.float32)
            k = sp.bn + ".running_mean"
            sd[k] = _rs(k, seed).uniform(-0.2, 0.2, c).astype(np.float32)
            k = sp.bn + ".running_var"
            sd[k] = _rs(k, seed).uniform(0.6, 1.6, c).astype(np.float32)
            sd[sp.bn + ".num_batches_tracked"] = np.zeros((), dtype=np.int64)
    return sd


def make_input(batch, input_size, seed=0):
    """x ~ N(0,1) float32 [B,3,S,S] (post-normalisation statistics)."""
    rs = np.random.RandomState(1000003 * seed + 17)
    return rs.standard_normal((batch, 3, input_size, input_size)).astype(np.float32)
