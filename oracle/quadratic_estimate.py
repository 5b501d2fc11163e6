"""Hu-Okamoto quadratic estimator, restated from src/quadratic_estimate.jl:29-200.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Fields are S0 Fourier arrays (B,1,Nx,Nyh); covariances are [x,ky] planes.
"""
import itertools

import numpy as np

from .flatsky import rfft2, irfft2, nan2zero, pinv, grad_mults

__all__ = ["quadratic_estimate", "qe_leg"]


def qe_leg(proj, C, inds, cache=None):
    """`QE_leg(C, inds...)` (:83-91): Map(nan2zero(C * ∇[1]^p1 * ∇[2]^p2 / sqrt(∇²)^n)); an index given as a
    1-tuple `(i,)` is the wave-vector l[i] (Julia `[i]`), a bare int is the unit vector l̂[i]."""
    n = sum(1 for x in inds if isinstance(x, int))
    first = [x[0] if isinstance(x, tuple) else x for x in inds]
    p1, p2 = first.count(1), first.count(2)
    key = (id(C), n, p1, p2)
    if cache is not None and key in cache:
        return cache[key][1]
    ilx, ily = grad_mults(proj)
    Cin = C
    if C.ndim == 2:
        C = C[None, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        v = C * ilx ** p1 * ily ** p2 / (proj.lmag.astype(np.float64) ** n if n else 1.0)
    out = irfft2(nan2zero(v.astype(np.complex128)), proj.Ny)
    if cache is not None:
        cache[key] = (Cin, out)                     # keep C alive so id() stays unique
    return out


def eps3(a, b):
    """levicivita([a, b, 3]) for a, b in {1, 2}"""
    return 1 if (a, b) == (1, 2) else (-1 if (a, b) == (2, 1) else 0)


def _grad(proj, i, fl):
    return grad_mults(proj)[i - 1] * fl


def _inds(D):
    # collect(product(repeated(1:2, D)...))[:]  -- first index varies fastest (irrelevant for the sums)
    return [t[::-1] for t in itertools.product((1, 2), repeat=D)]


def quadratic_estimate(proj, which, d1, d2, Cf, Cft, Cn, Cphi, TF, wiener_filtered=True, AL=None):
    """which in {'TT','EE','EB'}.  d1, d2: data legs as dicts of S0 Fourier fields {'T'} or {'E','B'};
    Cf, Cft, Cn, TF: dicts of [x,ky] planes keyed the same way (weights = :unlensed, :95,131,165).
    Returns (phiqe, AL, Nphi)."""
    cache = {}
    L = lambda C, *inds: qe_leg(proj, C, inds, cache)
    F = rfft2
    if which == "TT":
        S = TF["T"] ** 2 * Cft["T"] + Cn["T"]                                        # :97
        CT = Cf["T"]
        with np.errstate(divide="ignore", invalid="ignore"):
            a = nan2zero(TF["T"] * d1["T"] / S)
            b = CT * nan2zero(TF["T"] * d2["T"] / S)
        un = -sum(_grad(proj, i, F(L(a) * L(b, (i,)))) for i in (1, 2))                # :101
        if AL is None:
            w1, w2, w3 = TF["T"] ** 2 * CT ** 2 * pinv(S), TF["T"] ** 2 * pinv(S), TF["T"] ** 2 * CT * pinv(S)
            A = lambda i, j: L(w1, (i,), (j,)) * L(w2) + L(w3, (i,)) * L(w3, (j,))       # :106-109
    elif which == "EE":
        TF2 = TF["E"] ** 2
        S = TF2 * Cft["E"] + Cn["E"]
        CE = Cf["E"]
        with np.errstate(divide="ignore", invalid="ignore"):
            a1 = CE * nan2zero(TF["E"] * d1["E"] / S)
            a2 = nan2zero(TF["E"] * d2["E"] / S)
        I = lambda i: -(2 * sum(L(a1, (i,), j, k) * L(a2, j, k) for (j, k) in _inds(2)) - L(a1, (i,)) * L(a2))   # :133-136
        un = sum(_grad(proj, i, F(I(i))) for i in (1, 2))
        if AL is None:
            w1, w2, w3 = TF2 * CE ** 2 * pinv(S), TF2 * pinv(S), TF2 * CE * pinv(S)
            def A(i, j):
                A1 = -4 * sum(eps3(m, p) * eps3(n, q) * (L(w1, (i,), (j,), k, l, m, n) * L(w2, k, l, p, q)
                                                          + L(w3, (i,), k, l, m, n) * L(w3, (j,), k, l, p, q))
                              for (k, l, m, n, p, q) in _inds(6) if eps3(m, p) * eps3(n, q) != 0)
                A2 = L(w1, (i,), (j,)) * L(w2) + L(w3, (i,)) * L(w3, (j,))
                return A1 + A2
    elif which == "EB":
        CE, CB = Cf["E"], Cf["B"]
        TF2E, TF2B = TF["E"] ** 2, TF["B"] ** 2
        SE = TF2E * Cft["E"] + Cn["E"]
        SB = TF2B * Cft["B"] + Cn["B"]
        with np.errstate(divide="ignore", invalid="ignore"):
            e1 = nan2zero(TF["E"] * d1["E"] / SE)
            b2 = nan2zero(TF["B"] * d2["B"] / SB)
        ce1, cb2 = CE * e1, CB * b2
        I = lambda i: 2 * sum(eps3(k, l) * (L(ce1, (i,), j, k) * L(b2, j, l) - L(e1, j, k) * L(cb2, (i,), j, l))
                              for (j, k, l) in _inds(3) if eps3(k, l) != 0)                      # :175-179
        un = sum(_grad(proj, i, F(I(i))) for i in (1, 2))
        if AL is None:
            wE2, wE1, wE0 = TF2E * CE ** 2 * pinv(SE), TF2E * CE * pinv(SE), TF2E * pinv(SE)
            wB0, wB1, wB2 = TF2B * pinv(SB), TF2B * CB * pinv(SB), TF2B * CB ** 2 * pinv(SB)
            A = lambda i, j: 4 * sum(eps3(m, p) * eps3(n, q) * (
                L(wE2, (i,), (j,), k, l, m, n) * L(wB0, k, l, p, q)
                - 2 * L(wE1, (i,), k, l, m, n) * L(wB1, (j,), k, l, p, q)
                + L(wE0, k, l, m, n) * L(wB2, (i,), (j,), k, l, p, q))
                for (k, l, m, n, p, q) in _inds(6) if eps3(m, p) * eps3(n, q) != 0)              # :186-192
    else:
        raise ValueError(which)
    if AL is None:
        tot = 0
        for (i, j) in _inds(2):
            gi, gj = grad_mults(proj)[i - 1], grad_mults(proj)[j - 1]
            tot = tot + np.abs(gi * gj * F(A(i, j)))[0, 0]                                   # :111,153,193
        AL = pinv(tot)
    Nphi = AL
    phiqe = AL * un
    if wiener_filtered:
        phiqe = (Cphi * pinv(Cphi + Nphi)) * phiqe
    return phiqe, AL, Nphi
