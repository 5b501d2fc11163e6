"""θ layer: parameter-dependent operators and the Gibbs θ pass (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference
    src/dataset.jl:272-274,316-328   Cf(r) = Cfs + (r/r₀) Cft,  Cϕ(Aϕ) = Aϕ Cϕ₀,
                                     G(Aϕ) = G₀⁻¹ sqrt(I + 2 Nϕ Cϕ(Aϕ)⁻¹), G₀ = sqrt(I + 2 Nϕ Cϕ₀⁻¹),
                                     D(r)  = sqrt((Cf(r) + σ²len + 2 Cn̂) Cf(r)⁻¹)
    src/dataset.jl:84-87             logpdf(Mixed; θ) = logpdf(ds; unmix(θ)) − logdet(D,θ) − logdet(G,θ)
    src/generic.jl:264-271           logdet(L,θ) = logdet(L()⁻¹ L(θ)) if L depends on a parameter named in θ, else 0
    src/specialops.jl:314-340        ParamDependentOp: recomputed only for the parameters it names
    src/sampling.jl:80-135,427-437   grid_and_sample / gibbs_sample_slice_θ!
The smoothing inside `grid_and_sample` is Loess.jl (local quadratic, tricube weights, span = 0.25) and the quadrature / root
finding are QuadGK / Roots in the reference: third-party numerics, **parity unpinned**; restated here as plain local-quadratic
LOESS + trapezoid CDF on a fine grid, checked against analytic densities (tests/test_oracle_theta.py).
"""
import copy

import numpy as np

from .flatsky import pinv, logdet_fourier

__all__ = ["ThetaDataSet", "loess", "grid_and_sample"]


class ThetaDataSet:
    """BaseDataSet with the ParamDependentOps of `load_sim`; `base` is the fiducial oracle DataSet (G₀-normalised G = I)."""

    def __init__(self, base, Cfs, Cten, r0=0.2, Aphi0=1.0):
        self.base, self.Cfs, self.Cten, self.r0, self.Aphi0 = base, Cfs, Cten, r0, Aphi0
        self.Cphi0 = base.Cphi / Aphi0
        self.s2len = base.proj.T(np.deg2rad(5 / 60) ** 2)

    def D(self, r):
        Cf = self.Cfs + self.Cten.scale(r / self.r0)
        return ((Cf + (self.base.Cnhat.scale(2) + self.s2len)) @ Cf.pinv()).sqrt(), Cf

    def G(self, Aphi):
        g0 = np.sqrt(1 + 2 * self.base.Nphi * pinv(self.Cphi0 * self.Aphi0))
        return pinv(g0) * np.sqrt(1 + 2 * self.base.Nphi * pinv(self.Cphi0 * Aphi))

    def at(self, r=None, Aphi=None):
        """(dataset at θ, logdet(D,θ), logdet(G,θ)); a parameter left None is 'not in θ' (operators stay fiducial, logdet term 0)"""
        ds = copy.copy(self.base)
        ds._L = None
        proj = ds.proj
        ldD = ldG = 0.0
        if r is not None:
            ds.D, ds.Cf = self.D(r)
            D0, _ = self.D(self.r0)
            ldD = (D0.pinv() @ ds.D).logdet(proj)
        if Aphi is not None:
            ds.Cphi = self.Cphi0 * Aphi
            ds.G = self.G(Aphi)
            ldG = logdet_fourier(proj, ds.G[None, None])              # G() = I at the fiducial point
        return ds, ldD, ldG

    def logpdf_mixed(self, fo, po, r=None, Aphi=None):
        ds, ldD, ldG = self.at(r, Aphi)
        return ds.logpdf_mixed(fo, po) - ldD - ldG


def loess(xs, ys, x, span=0.25, degree=2):
    """local polynomial regression with tricube weights over the ceil(span·n) nearest points, evaluated at x (array)"""
    xs, ys, x = np.asarray(xs, float), np.asarray(ys, float), np.atleast_1d(np.asarray(x, float))
    n = len(xs)
    q = int(min(n, max(degree + 1, np.ceil(span * n))))
    out = np.empty_like(x)
    for i, x0 in enumerate(x):
        d = np.abs(xs - x0)
        idx = np.argpartition(d, q - 1)[:q]
        h = d[idx].max()
        w = (1 - (d[idx] / h) ** 3) ** 3 if h > 0 else np.ones(q)
        w = np.maximum(w, 1e-12)
        A = np.vander(xs[idx] - x0, degree + 1, increasing=True)
        coef, *_ = np.linalg.lstsq(A * np.sqrt(w)[:, None], ys[idx] * np.sqrt(w), rcond=None)
        out[i] = coef[0]
    return out


def grid_and_sample(logpdfs, xs, u, span=0.25, nfine=2001):
    """`grid_and_sample(logpdfs, xs)` (sampling.jl:91-131): trim non-finite ends, subtract the maximum, smooth the LOG pdf, normalise,
    draw by inverse-transform with the uniform `u`.  Returns (sample, (x_fine, smoothed normalised log pdf), normalised log pdf at xs)."""
    xs, lp = np.asarray(xs, float), np.asarray(logpdfs, float)
    fin = np.flatnonzero(np.isfinite(lp))
    xs, lp = xs[fin[0]:fin[-1] + 1], lp[fin[0]:fin[-1] + 1]
    lp = lp - lp.max()
    xf = np.linspace(xs[0], xs[-1], nfine)
    sm = loess(xs, lp, xf, span)
    p = np.nan_to_num(np.exp(sm))
    cdf = np.concatenate([[0.0], np.cumsum((p[1:] + p[:-1]) / 2 * np.diff(xf))])
    logA = np.log(cdf[-1])
    cdf = cdf / cdf[-1]
    sample = float(np.interp(u, cdf, xf))
    return sample, (xf, sm - logA), loess(xs, lp, xs, span) - logA
