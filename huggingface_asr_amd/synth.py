"""Deterministic, counter-based synthetic weights and inputs (no torch RNG).

There are no trained checkpoints or real utterances available offline, so
every fixture, parity test and bench run uses weights/inputs produced here.  The generator is
a pure function of (seed, tensor name, element index): a splitmix64 hash mapped to a uniform
value, so the same numbers are reproduced in this container (where the reference is imported
to make golden vectors) and on the GPU box (where only the seed travels).
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(seed: int, name: str, shape, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """float32 uniform [lo, hi) tensor, a pure function of (seed, name, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64((zlib.crc32(name.encode()) << 32) ^ (seed & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(_splitmix64(idx ^ key) + key)
    u = (h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))  # 24-bit mantissa, [0,1)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed: int, name: str, shape, std: float = 1.0) -> np.ndarray:
    """float32 N(0, std) via Box-Muller on two counter-based uniforms."""
    u1 = uniform(seed, name + "/u1", shape, 0.0, 1.0).astype(np.float64)
    u2 = uniform(seed, name + "/u2", shape, 0.0, 1.0).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0**-25)))
    return (std * r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def init_param(seed: int, name: str, shape) -> np.ndarray:
    """Seeded init for one state-dict entry of the E-Branchformer CTC encoder.

    Scales keep activations O(1) through the stack so that parity tests are sensitive:
    matrices ~U(+-sqrt(3/fan_in)) * 0.8, biases +-0.1, LayerNorm gamma 1+-0.1 / beta +-0.1,
    depthwise conv taps +-1/sqrt(k), positional biases +-0.1.
    """
    leaf = name.rsplit(".", 1)[-1]
    is_norm = ("layer_norm" in name) or (".norm." in name) or (".ln_" in name) or \
        name.endswith((".ff1.0.weight", ".ff1.0.bias", ".ff2.0.weight", ".ff2.0.bias"))
    if is_norm:
        if leaf == "weight":
            return 1.0 + uniform(seed, name, shape, -0.1, 0.1)
        return uniform(seed, name, shape, -0.1, 0.1)
    if leaf in ("pos_bias_u", "pos_bias_v", "masked_spec_embed"):
        return uniform(seed, name, shape, -0.1, 0.1)
    if leaf == "bias":
        return uniform(seed, name, shape, -0.1, 0.1)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        a = 0.8 * np.sqrt(3.0 / fan_in)
        return uniform(seed, name, shape, -a, a)
    return uniform(seed, name, shape, -0.1, 0.1)


def state_dict_numpy(shapes: dict, seed: int) -> dict:
    """name -> float32 array for every (name, shape) in `shapes`."""
    return {k: init_param(seed, k, tuple(s)) for k, s in shapes.items()}


def waveforms(seed: int, batch: int, num_samples: int, tone_hz: float = 220.0) -> np.ndarray:
    """(batch, num_samples) float32: 0.1*N(0,1) noise + a quiet tone (SURVEY.md §8d inputs)."""
    x = normal(seed, "wave", (batch, num_samples), 0.1)
    t = np.arange(num_samples, dtype=np.float64) / 16000.0
    x += (0.05 * np.sin(2 * np.pi * tone_hz * t)).astype(np.float32)[None, :]
    return x


def labels(seed: int, batch: int, length: int, vocab: int, lo: int = 5) -> np.ndarray:
    u = uniform(seed, "labels", (batch, length), 0.0, 1.0)
    return (lo + np.floor(u * (vocab - lo))).astype(np.int64).clip(lo, vocab - 1)


def dropout_keep(seed: int, stream_id: int, n: int, p: float) -> np.ndarray:
    """bool keep-mask of csrc/dropout.hip for logical element indices 0..n-1 (the CPU oracle runs with the kernels' exact masks)."""
    key = np.uint64(((stream_id & 0xFFFFFFFF) << 32) ^ (seed & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64((idx >> np.uint64(2)) ^ key)                # one hash per QUAD of elements: four 16-bit draws, from the top (csrc/common.hpp mask_hash / mask_u01)
    u16 = (h >> (np.uint64(48) - np.uint64(16) * (idx & np.uint64(3)))) & np.uint64(0xFFFF)
    u = u16.astype(np.float32) * np.float32(1.0 / (1 << 16))
    return u >= np.float32(p)


def mask_noise(seed: int, stream_id: int, shape, std: float) -> np.ndarray:
    """float32 noise of csrc/bestrq.hip's mask_noise_kernel for logical indices 0..prod(shape)-1 (Box-Muller on two hashed uniforms)."""
    n = int(np.prod(shape))
    key = np.uint64(((stream_id & 0xFFFFFFFF) << 32) ^ (seed & 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h1 = _splitmix64(_splitmix64((np.uint64(2) * idx) ^ key) + key)
        h2 = _splitmix64(_splitmix64((np.uint64(2) * idx + np.uint64(1)) ^ key) + key)
    u1 = np.maximum((h1 >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24)), np.float32(2.98023224e-08))
    u2 = (h2 >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return (np.float32(std) * np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)).astype(np.float32).reshape(shape)
