"""Host side of the engine's counter-based generator (Philox4x32-10, the same one `cmbl_randn` runs on the device): scalar
uniforms for accept/reject steps (`log(rand(rng))`, src/sampling.jl:414) and the stream-id convention of the drivers.  Maps are
always drawn on the device (ProjLambert.randn)."""

_M0, _M1, _W0, _W1, _M32 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF

# stream ids: (draw kind) + 16 * (running step index) -- independent sequences of one chain's key
STREAM_F, STREAM_N, STREAM_P, STREAM_U = 0, 1, 2, 3         # STREAM_U + 1 + j: uniform of the j-th θ pass
STREAM_INIT = 12                                         # starting point of a chain: step 0 = ϕ ~ prior, 1 + j = θ_j ~ prior


def stream_id(kind, step):
    return kind + 16 * int(step)


def philox4x32_10(c, k):
    c, k = list(c), list(k)
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & _M32, (p0 >> 32) ^ c[3] ^ k[1], p0 & _M32]
        k = [(k[0] + _W0) & _M32, (k[1] + _W1) & _M32]
    return c


def uniform(seed, stream, n=1):
    """n uniforms in (0,1) of sequence (seed, stream): word j -> (w + 0.5)/2^32"""
    out = []
    for c in range((n + 3) // 4):
        out += philox4x32_10([c & _M32, c >> 32, stream & _M32, (stream >> 32) & _M32], [seed & _M32, (seed >> 32) & _M32])
    return [(w + 0.5) / 4294967296.0 for w in out[:n]]
