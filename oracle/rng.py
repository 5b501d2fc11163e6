"""Counter-based white noise -- oracle of the device generator (TEST INFRASTRUCTURE ONLY).

The reference draws `randn!(rng, Map)` from whatever `AbstractRNG` the caller passes (src/base_fields.jl:169-170,
src/specialops.jl:6,93; CURAND on its GPU path, ext/CMBLensingCUDAExt.jl:67-73); its streams are not reproducible outside
Julia, and its tests are seed-agnostic.  The engine therefore defines its own generator and this file restates it:
Philox4x32-10 (Salmon, Moraes, Dror & Shaw, SC'11; Random123 v1.14), pinned by that paper's known-answer vectors
(tests/test_oracle_rng.py), followed by Box-Muller in float64.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32 words, key: (2,) ints -> (..., 4) uint32 words after 10 rounds."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]                      # 32x32 -> 64 bit products, no overflow in uint64
        c = [(p1 >> _S32) ^ c[1] ^ np.uint64(k0), p1 & _MASK, (p0 >> _S32) ^ c[3] ^ np.uint64(k1), p0 & _MASK]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def philox_words(seed, stream, ncounters):
    """words (ncounters, 4) of key = seed (lo, hi), counters (c lo, c hi, stream lo, stream hi), c = 0..ncounters-1"""
    c = np.arange(ncounters, dtype=np.uint64)
    ctr = np.stack([c & _MASK, c >> _S32, np.full_like(c, stream & 0xFFFFFFFF), np.full_like(c, (stream >> 32) & 0xFFFFFFFF)], axis=-1)
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def randn(seed, stream, n, dtype=np.float64):
    """n standard normals: counter c -> elements 4c..4c+3 = (r0 cos, r0 sin, r1 cos, r1 sin), r = sqrt(-2 ln u1),
    angle 2 pi u2, u = (word + 0.5) / 2^32 from word pairs (0,1) and (2,3); computed in float64, then cast."""
    w = philox_words(seed, stream, (n + 3) // 4).astype(np.float64)
    u = (w + 0.5) * 2.0 ** -32
    out = np.empty((w.shape[0], 4))
    for h in range(2):
        r = np.sqrt(-2.0 * np.log(u[:, 2 * h]))
        out[:, 2 * h] = r * np.cos(2 * np.pi * u[:, 2 * h + 1])
        out[:, 2 * h + 1] = r * np.sin(2 * np.pi * u[:, 2 * h + 1])
    return out.reshape(-1)[:n].astype(dtype)


def uniform(seed, stream, n):
    """n uniforms in (0,1): word j of the same Philox sequence -> (w + 0.5)/2^32 (accept/reject draws)."""
    w = philox_words(seed, stream, (n + 3) // 4).reshape(-1)[:n].astype(np.float64)
    return (w + 0.5) * 2.0 ** -32
