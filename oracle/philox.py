"""NumPy restatement of the engine's control-noise sampler (stage S1).  TEST INFRASTRUCTURE.

The reference draws ``np.random.multivariate_normal(0, Sigma, (K, T))``
(/root/reference/controllers/mppi_differential_drive.py:273-283).  The engine replaces
that draw with a counter-based generator so a sample depends only on
``(seed, iteration, global k, t)`` -- shard-count invariant:

* Philox4x32-10, key = (seed_lo, seed_hi), counter = (k_global, t >> 1, iteration, stream)
* words (r0, r1) serve even t, (r2, r3) odd t
* u = (2*(r >> 9) + 1) * 2^-24  in (0, 1)  (exact in f32)
* Box-Muller: z0 = sqrt(-2 ln u_a) cos(2 pi u_b), z1 = ... sin(2 pi u_b)
* eps = chol(Sigma) @ (z0, z1)

The device evaluates the transcendental part with fp32 hardware ops, this file in f64 and
rounds to f32, so the two agree to a few f32 ulps (tests state the tolerance).  Large-K
golden fixtures store only the seed and regenerate eps through this file.
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32 with 10 rounds.  All inputs broadcastable uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            if r < 9:
                k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
                k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def uniform_open(r):
    """(2*(r>>9)+1) * 2^-24, exactly representable in f32."""
    return (2.0 * (np.asarray(r, np.uint32) >> np.uint32(9)).astype(np.float64) + 1.0) * 2.0 ** -24


def standard_normal_pairs(seed: int, iteration: int, K: int, T: int, k_offset: int = 0, stream: int = 0):
    """z[K, T, 2] in f64 (before the Cholesky factor)."""
    k = (np.arange(K, dtype=np.uint64) + np.uint64(k_offset)).astype(np.uint32)[:, None]
    t = np.arange(T, dtype=np.uint32)[None, :]
    r0, r1, r2, r3 = philox4x32_10(k, t >> np.uint32(1), np.uint32(iteration & 0xFFFFFFFF), np.uint32(stream),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    odd = (t & np.uint32(1)).astype(bool)
    ra = np.where(odd, r2, r0)
    rb = np.where(odd, r3, r1)
    ua, ub = uniform_open(ra), uniform_open(rb)
    rad = np.sqrt(-2.0 * np.log(ua))
    ang = 2.0 * np.pi * ub
    return np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1)


def cholesky2(sigma):
    """Lower Cholesky factor of a 2x2 SPD matrix (f64)."""
    s = np.asarray(sigma, dtype=np.float64)
    l00 = np.sqrt(s[0, 0])
    l10 = s[1, 0] / l00
    l11 = np.sqrt(s[1, 1] - l10 * l10)
    return np.array([[l00, 0.0], [l10, l11]])


def sample_epsilon(sigma, seed: int, iteration: int, K: int, T: int, k_offset: int = 0, stream: int = 0) -> np.ndarray:
    """eps[K, T, 2] as float32, distributed N(0, sigma).  ``stream``: the counter's fourth word (mppi_config.noise_stream
    + the agent of a batched handle)."""
    z = standard_normal_pairs(seed, iteration, K, T, k_offset, stream)
    L = cholesky2(sigma).astype(np.float32).astype(np.float64)
    e0 = L[0, 0] * z[..., 0]
    e1 = L[1, 0] * z[..., 0] + L[1, 1] * z[..., 1]
    return np.stack([e0, e1], axis=-1).astype(np.float32)
