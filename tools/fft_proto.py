"""NumPy emulation of the in-LDS FFT schedule used by csrc/fft_lds.hpp (index-math prototype).

Forward: in-place radix-2 DIF (natural in, bit-reversed out). Inverse: in-place radix-2 DIT
(bit-reversed in, natural out).  Real transforms use the packed trick (length-N real as
length-N/2 complex) with the frequency data kept at bit-reversed LDS positions.
Run:  python tools/fft_proto.py
"""
import numpy as np


def brev(i, bits):
    r = 0
    for b in range(bits):
        r |= ((i >> b) & 1) << (bits - 1 - b)
    return r


def dif_forward(x):            # e^{-i}, natural -> bitrev
    x = x.copy(); N = x.size; h = N // 2
    while h >= 1:
        for a in range(N):
            if (a // h) % 2 == 0:
                j = a % h
                w = np.exp(-2j * np.pi * j / (2 * h))
                u, v = x[a], x[a + h]
                x[a], x[a + h] = u + v, (u - v) * w
        h //= 2
    return x


def dit_inverse(x):            # e^{+i}, bitrev -> natural, unnormalised
    x = x.copy(); N = x.size; h = 1
    while h < N:
        for a in range(N):
            if (a // h) % 2 == 0:
                j = a % h
                w = np.exp(2j * np.pi * j / (2 * h))
                u, t = x[a], w * x[a + h]
                x[a], x[a + h] = u + t, u - t
        h *= 2
    return x


def r2c_packed(f):
    """real f[N] -> A[0..N/2] stored at LDS slots: slot[brev(k)] for k<M, slot[M] for k=M."""
    N = f.size; M = N // 2; bits = int(np.log2(M))
    z = dif_forward(f[0::2] + 1j * f[1::2])            # Z[k] at z[brev(k)]
    lds = np.zeros(M + 1, complex); lds[:M] = z
    for k in range(0, M // 2 + 1):
        k2 = (M - k) % M
        a, b = lds[brev(k, bits)], lds[brev(k2, bits)]
        w = np.exp(-2j * np.pi * k / N)
        Ak = 0.5 * (a + np.conj(b)) - 0.5j * w * (a - np.conj(b))
        w2 = np.exp(-2j * np.pi * (M - k) / N)
        Ak2 = 0.5 * (b + np.conj(a)) - 0.5j * w2 * (b - np.conj(a))
        if k == 0:
            lds[brev(0, bits)] = Ak.real                 # A[0]  (real)
            lds[M] = Ak2.real                            # A[M]  (real): uses Z[M]=Z[0]
        else:
            lds[brev(k, bits)] = Ak
            lds[brev(k2, bits)] = Ak2
    return lds


def c2r_packed(lds, N):
    """inverse of the above incl. FFTW semantics: Im A[0], Im A[M] ignored; normalised by 1/N... returns f*N? no: returns f (normalised)."""
    M = N // 2; bits = int(np.log2(M))
    lds = lds.copy()
    for k in range(0, M // 2 + 1):
        k2 = M - k
        Ak = lds[brev(k, bits)] if k < M else lds[M]
        Ak2 = lds[M] if k2 == M else lds[brev(k2, bits)]
        if k == 0:
            Ak, Ak2 = Ak.real, Ak2.real                  # c2r ignores these imaginary parts
        w = np.exp(2j * np.pi * k / N)
        Zk = (Ak + np.conj(Ak2)) + 1j * w * (Ak - np.conj(Ak2))
        w2 = np.exp(2j * np.pi * k2 / N)
        Zk2 = (Ak2 + np.conj(Ak)) + 1j * w2 * (Ak2 - np.conj(Ak))
        lds[brev(k, bits)] = Zk
        if 0 < k2 < M:
            lds[brev(k2, bits)] = Zk2
    z = dit_inverse(lds[:M]) / N                         # (1/2)·(1/M) folded: Z above carries factor 2
    f = np.empty(N); f[0::2] = z.real; f[1::2] = z.imag
    return f


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for N in (8, 16, 64, 256):
        bits = int(np.log2(N))
        x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        X = dif_forward(x)
        ref = np.fft.fft(x)
        assert np.allclose([X[brev(k, bits)] for k in range(N)], ref)
        assert np.allclose(dit_inverse(X) / N, x)
        f = rng.standard_normal(N)
        lds = r2c_packed(f)
        M = N // 2; mb = int(np.log2(M))
        A = np.array([lds[brev(k, mb)] for k in range(M)] + [lds[M]])
        assert np.allclose(A, np.fft.rfft(f)), N
        assert np.allclose(c2r_packed(lds, N), f)
        # non-Hermitian junk in Im A[0], Im A[M] must be ignored like numpy/FFTW
        G = A * (1j * np.arange(M + 1))
        lds2 = np.zeros(M + 1, complex)
        for k in range(M):
            lds2[brev(k, mb)] = G[k]
        lds2[M] = G[M]
        assert np.allclose(c2r_packed(lds2, N), np.fft.irfft(G, N)), N
    print("fft prototype OK")
