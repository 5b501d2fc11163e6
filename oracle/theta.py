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

__all__ = ["ThetaDataSet", "loess", "grid_and_sample", "findbin", "bandpower_rescale"]


def findbin(ledges, l):
    """`findbin` (src/proj_lambert.jl:402-404), 0-based: out of range -> len(ledges) - 1; else (first edge > ℓ) - 1"""
    ledges = list(ledges)
    out = np.empty(np.shape(l), int)
    for i, x in np.ndenumerate(np.asarray(l, float)):
        if x < ledges[0] or x >= ledges[-1]:
            out[i] = len(ledges) - 1
        else:
            out[i] = next(k for k, e in enumerate(ledges) if e > x) - 1
    return out


def bandpower_rescale(arr, bin_idx, amplitudes):
    """src/proj_lambert.jl:405-408"""
    return np.concatenate([np.asarray(amplitudes, float), [1.0]])[bin_idx] * arr


class ThetaDataSet:
    """BaseDataSet with the ParamDependentOps of `load_sim`; `base` is the fiducial oracle DataSet (G₀-normalised G = I)."""

    def __init__(self, base, Cfs, Cten, r0=0.2, Aphi0=1.0, bands=None):
        """bands: {θname: (plane indices of Cfs to rescale, ℓedges)} -- bandpower amplitudes of the scalar part (src/proj_lambert.jl:374-400)"""
        self.base, self.Cfs0, self.Cfs, self.Cten, self.r0, self.Aphi0 = base, Cfs, Cfs, Cten, r0, Aphi0
        self.bands = {k: (pl, findbin(le, base.proj.lmag), len(le) - 1) for k, (pl, le) in (bands or {}).items()}
        self.Cphi0 = base.Cphi / Aphi0
        self.s2len = base.proj.T(np.deg2rad(5 / 60) ** 2)

    def D(self, r):
        """(D(r), Cf(r, amplitudes)): D is built from the covariance `load_sim` created (src/dataset.jl:322-328: it closes over that
        Cf and names r only), so bandpower amplitudes of a Cf assigned later do not enter it"""
        Cf0 = self.Cfs0 + self.Cten.scale(r / self.r0)
        return ((Cf0 + (self.base.Cnhat.scale(2) + self.s2len)) @ Cf0.pinv()).sqrt(), self.Cfs + self.Cten.scale(r / self.r0)

    def G(self, Aphi):
        g0 = np.sqrt(1 + 2 * self.base.Nphi * pinv(self.Cphi0 * self.Aphi0))
        return pinv(g0) * np.sqrt(1 + 2 * self.base.Nphi * pinv(self.Cphi0 * Aphi))

    def _rescaled_Cfs(self, amps):
        C = copy.deepcopy(self.Cfs0)
        for name, (planes, idx, nb) in self.bands.items():
            a = amps.get(name)
            a = np.ones(nb) if a is None else a
            for k in planes:                                   # plane k of the operator's array form (HarmOp.arrays order)
                if C.P == 3:
                    te = list(C.te)
                    te[k] = bandpower_rescale(self.Cfs0.te[k], idx, a)
                    C.te = tuple(te)
                else:
                    C.d[k] = bandpower_rescale(self.Cfs0.d[k], idx, a)
        return C

    def at(self, r=None, Aphi=None, **amps):
        """(dataset at θ, logdet(D,θ), logdet(G,θ)); a parameter left None is 'not in θ' (operators stay fiducial, logdet term 0)"""
        ds = copy.copy(self.base)
        ds._L = None
        proj = ds.proj
        ldD = ldG = 0.0
        if amps:
            self.Cfs = self._rescaled_Cfs(amps)
            ds.Cf = self.Cfs + self.Cten                        # D does not depend on the amplitudes (src/dataset.jl:322-328: r only)
        else:
            self.Cfs = self.Cfs0
        if r is not None:
            ds.D, ds.Cf = self.D(r)
            D0, _ = self.D(self.r0)
            ldD = (D0.pinv() @ ds.D).logdet(proj)
        if Aphi is not None:
            ds.Cphi = self.Cphi0 * Aphi
            ds.G = self.G(Aphi)
            ldG = logdet_fourier(proj, ds.G[None, None])              # G() = I at the fiducial point
        return ds, ldD, ldG

    def logpdf_mixed(self, fo, po, r=None, Aphi=None, **amps):
        ds, ldD, ldG = self.at(r, Aphi, **amps)
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
