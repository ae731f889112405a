"""Counter-based synthetic data generator shared by the oracle, tests and bench.

TEST INFRASTRUCTURE / data plumbing -- not part of the product path.

Every value is a pure function of (seed, stream name, element index), so this
container, the GPU box and every rank regenerate bit-identical inputs and
weights without shipping them (SURVEY.md section 8d "Synthetic inputs").
Uses splitmix64 hashing + Box-Muller; numpy only.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream_key(seed, stream):
    if isinstance(stream, str):
        stream = zlib.crc32(stream.encode())
    with np.errstate(over="ignore"):
        k = _splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF))
        k = _splitmix64(k ^ np.uint64(stream))
    return k


def uniform01(seed, stream, n, offset=0):
    """n float64 values in [0,1): element i depends only on (seed, stream, offset+i)."""
    key = _stream_key(seed, stream)
    with np.errstate(over="ignore"):
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        bits = _splitmix64(key ^ (idx * np.uint64(0xD1342543DE82EF95) & _M64))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform(seed, stream, shape, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    u = uniform01(seed, stream, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, stream, shape):
    """Standard normal float32 via Box-Muller on a pair of counter streams."""
    n = int(np.prod(shape))
    u1 = uniform01(seed, stream, n, offset=0)
    u2 = uniform01(seed, stream, n, offset=1 << 40)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    z = r * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)
