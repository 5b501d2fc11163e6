"""θ layer: the parameter-dependent operators of `load_sim` and the Gibbs θ pass of `sample_joint`.

    src/dataset.jl:272-274,316-328   Cf(r) = Cfs + (r/r₀) Cft,  Cϕ(Aϕ) = Aϕ Cϕ₀,  G(Aϕ) = G₀⁻¹ sqrt(I + 2 Nϕ Cϕ(Aϕ)⁻¹),
                                     D(r) = sqrt((Cf(r) + σ²len + 2 Cn̂) Cf(r)⁻¹)
    src/dataset.jl:84-87             logpdf(Mixed; θ) = logpdf(ds; unmix(θ)) − logdet(D,θ) − logdet(G,θ)
    src/generic.jl:264-271           logdet(L,θ) = logdet(L()⁻¹ L(θ)) when θ names a parameter of L, else 0
    src/sampling.jl:80-135,427-437   grid_and_sample, gibbs_sample_slice_θ!
Like the reference's ParamDependentOp (src/specialops.jl:314-330: recompute on the host side of the storage, then adapt) the ℓ-space
planes are recomputed on the host for each θ and uploaded (a handful of (Nx, Ny/2+1) planes); every field operation stays on the
device.  The smoothing / quadrature inside grid_and_sample are Loess.jl / QuadGK / Roots in the reference -- third-party numerics
whose outputs are not pinned; here: local-quadratic LOESS with tricube weights and a trapezoid CDF on a fine grid.
"""
import numpy as np

from .sim import HarmOp, _pinv


# ---- bandpower amplitudes (src/proj_lambert.jl:374-411) ----------------------------------------------------------------------
def findbin(ledges, l):
    """bin index (0-based) of every ℓ in the half-open bins [ℓedges[i], ℓedges[i+1]); out of range -> len(ledges) - 1, the slot of
    the constant amplitude 1 that `bandpower_rescale` appends (src/proj_lambert.jl:402-404)"""
    ledges, l = np.asarray(ledges, float), np.asarray(l, float)
    idx = np.searchsorted(ledges, l, side="right") - 1               # findfirst(>(ℓ), ℓedges) - 1
    return np.where((l < ledges[0]) | (l >= ledges[-1]), len(ledges) - 1, idx)


def bandpower_rescale(plane, bin_idx, amplitudes):
    """`[amplitudes; 1][ℓbin_indices] .* arr` (src/proj_lambert.jl:405-408)"""
    a = np.append(np.asarray(amplitudes, float), 1.0)
    return a[bin_idx] * plane


class BinRescaledCov:
    """`Cℓ_to_Cov(pol, proj, (Cℓ, ℓedges, θname), ...)`: a covariance whose TT / EE / TE planes are rescaled by one amplitude per
    ℓ-bin (ParamDependentOp over the amplitude vectors; BB is never rescaled, like the reference).  `bands` maps a spectrum
    ("TT", "EE", "TE") to (ℓedges, θname); `op(**θ)` gives the HarmOp at the named amplitudes (default: all ones)."""

    def __init__(self, pol, proj, cls, bands):
        self.C0 = HarmOp.from_cls(pol, proj, cls)
        self.pol = pol
        plane_of = {"I": {"TT": [0]}, "P": {"EE": [0]}, "IP": {"TT": [0], "TE": [1, 2], "EE": [3]}}[pol]
        self.bands = {}
        for spec, (ledges, name) in bands.items():
            if spec not in plane_of:
                raise ValueError(f"no rescalable {spec} spectrum for pol={pol} (BB is fixed, src/proj_lambert.jl:385-393)")
            self.bands[name] = (plane_of[spec], findbin(ledges, proj.lmag), len(ledges) - 1)
        self.names = list(self.bands)

    def __call__(self, **theta):
        p = self.C0.p.copy()
        for name, (planes, idx, nb) in self.bands.items():
            amps = theta.get(name)
            amps = np.ones(nb) if amps is None else np.asarray(amps, float)
            assert amps.shape == (nb,), f"{name}: expected {nb} amplitudes"
            for k in planes:
                p[k] = bandpower_rescale(self.C0.p[k], idx, amps)
        return HarmOp(p)


def _ops(ds, r, Aphi, bands=None):
    h = ds.host
    proj = ds.proj
    out, logdet_mix = {}, 0.0
    Cfs = h["Cfs_bands"](**(bands or {})) if "Cfs_bands" in h else h["Cfs"]       # bandpower amplitudes rescale the scalar part
    Cf = Cfs + h["Cten"].scale((h["r0"] if r is None else r) / h["r0"])
    Cphi0 = np.asarray(h.get("Cphi0", h["Cphi"]), float)
    Cphi = Cphi0 * (h["Aphi0"] if Aphi is None else Aphi) / h["Aphi0"]
    Dof = lambda C: ((C + (h["Cn"].scale(2) + h["s2len"])) @ C.pinv()).sqrt()
    # D closes over the covariance load_sim built and names r only (src/dataset.jl:322-328): bandpower amplitudes do not enter it
    D = Dof(h["Cfs"] + h["Cten"].scale((h["r0"] if r is None else r) / h["r0"]))
    if r is not None:
        D0 = Dof(h["Cfs"] + h["Cten"])
        logdet_mix += (D0.pinv() @ D).logdet(proj)
    Nphi = np.asarray(h["Nphi"], float)
    G = np.ones_like(Cphi0) if "G_user" not in h else np.asarray(h["G_user"], float)
    if Aphi is not None and "G_user" not in h:      # a user-supplied G stays constant: `if G == nothing` (src/dataset.jl:317-320), logdet(G,θ) = 0
        g0 = np.sqrt(1 + 2 * Nphi * _pinv(Cphi0))
        G = _pinv(g0) * np.sqrt(1 + 2 * Nphi * _pinv(Cphi))
        logdet_mix += HarmOp([G]).logdet(proj)
    precond = Cf.pinv() + (h["B"].T() @ h["Mf"].T() @ h["Cn"].pinv() @ h["Mf"] @ h["B"])
    out = dict(Cf_inv=Cf.pinv().p, D=D.p, D_inv=D.pinv().p, precond_inv=precond.pinv().p, Cphi_inv=_pinv(Cphi)[None], G_inv=_pinv(G)[None])
    logdet_sum = Cf.logdet(proj) + h["Cn"].logdet(proj) + HarmOp([Cphi]).logdet(proj)
    return out, logdet_sum, logdet_mix, dict(Cf=Cf, Cphi=Cphi, D=D, G=G, precond=precond)


def use_bandpowers(ds, bands, cls_scalar):
    """Make the scalar part of `ds.Cf` a bandpower-rescaled covariance: `bands` = {"EE": (ℓedges, "AEE"), ...}
    (`ds.Cf = Cℓ_to_Cov(pol, proj, (Cℓ, ℓedges, :AEE), ...)` in the reference).  The amplitude vectors then are parameters of
    `set_theta` / `logpdf_mixed_theta` / the Gibbs θ pass, next to r and Aϕ."""
    pol = {1: "I", 2: "P", 3: "IP"}[ds.P]
    ds.host["Cfs_bands"] = BinRescaledCov(pol, ds.proj, cls_scalar, bands)


def set_theta(ds, r=None, Aphi=None, **bands):
    """Evaluate the dataset's ParamDependentOps at θ = (r, Aϕ[, bandpower amplitude vectors]) (`ds(θ)`, src/dataset.jl:23-31) and
    make them current on the device; a parameter left None is 'not named in θ': its operators stay fiducial and its logdet term is
    0.  `set_theta(ds)` restores the fiducial dataset."""
    h = ds.host
    h.setdefault("Cphi0", np.asarray(h["Cphi"], float).copy())
    if bands and "Cfs_bands" not in h:
        raise ValueError("bandpower amplitudes given but the dataset has no bandpower covariance (theta.use_bandpowers)")
    ops, logdet_sum, logdet_mix, host = _ops(ds, r, Aphi, bands)
    for k, v in ops.items():
        ds.set_op(k, v)
    ds.set_logdet(logdet_sum)
    ds.logdet_mix = float(logdet_mix)
    h.update(host)
    ds.L.invalidate()
    ds.theta = dict(r=r, Aphi=Aphi, **bands)


def logpdf_mixed_theta(ds, fo, po, r=None, Aphi=None, **bands):
    """logpdf(Mixed(ds); f°, ϕ°, θ) (src/dataset.jl:84-87), per batch slot; leaves the dataset at θ"""
    set_theta(ds, r, Aphi, **bands)
    return ds.logpdf_mixed(fo, po)


def loess(xs, ys, x, span=0.25, degree=2):
    """local polynomial regression, tricube weights over the ceil(span·n) nearest points, evaluated at x"""
    xs, ys, x = np.asarray(xs, float), np.asarray(ys, float), np.atleast_1d(np.asarray(x, float))
    n = len(xs)
    q = int(min(n, max(degree + 1, np.ceil(span * n))))
    out = np.empty_like(x)
    for i, x0 in enumerate(x):
        d = np.abs(xs - x0)
        idx = np.argpartition(d, q - 1)[:q]
        hmax = d[idx].max()
        w = np.maximum((1 - (d[idx] / hmax) ** 3) ** 3 if hmax > 0 else np.ones(q), 1e-12)
        A = np.vander(xs[idx] - x0, degree + 1, increasing=True)
        out[i] = np.linalg.lstsq(A * np.sqrt(w)[:, None], ys[idx] * np.sqrt(w), rcond=None)[0][0]
    return out


def grid_and_sample(logpdfs, xs, u, span=0.25, nfine=2001):
    """`grid_and_sample(logpdfs, xs)` (src/sampling.jl:91-131) with the uniform draw `u` injected: trim non-finite ends, subtract
    the maximum, smooth the log pdf, normalise, inverse-transform sample.  -> (sample, (x_fine, log pdf), log pdf at xs)"""
    xs, lp = np.asarray(xs, float), np.asarray(logpdfs, float)
    fin = np.flatnonzero(np.isfinite(lp))
    xs, lp = xs[fin[0]:fin[-1] + 1], lp[fin[0]:fin[-1] + 1]
    lp = lp - lp.max()
    xf = np.linspace(xs[0], xs[-1], nfine)
    sm = loess(xs, lp, xf, span)
    p = np.nan_to_num(np.exp(sm))
    cdf = np.concatenate([[0.0], np.cumsum((p[1:] + p[:-1]) / 2 * np.diff(xf))])
    logA = np.log(cdf[-1])
    return float(np.interp(u, cdf / cdf[-1], xf)), (xf, sm - logA), loess(xs, lp, xs, span) - logA


def gibbs_sample_theta(ds, fo, po, theta, key, xs, u, span=0.25):
    """`gibbs_sample_slice_θ!(k)` (src/sampling.jl:427-437): conditional of θ[key] given (f°, ϕ°) and the other parameters, on the grid
    `xs`, one draw per batch slot with uniforms `u`.  Returns (new value per slot, log pdf grid (nslots, len(xs)))."""
    lps = []
    for x in xs:
        th = dict(theta, **{key: float(x)})
        lps.append(logpdf_mixed_theta(ds, fo, po, **th))
    lps = np.array(lps).T                                       # (B, nx)
    out = [grid_and_sample(lps[b], xs, u[b], span) for b in range(lps.shape[0])]
    return np.array([o[0] for o in out]), np.array([o[2] for o in out])
