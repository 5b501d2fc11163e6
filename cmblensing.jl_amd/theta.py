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


def _ops(ds, r, Aphi):
    h = ds.host
    proj = ds.proj
    out, logdet_mix = {}, 0.0
    Cf = h["Cfs"] + h["Cten"].scale((h["r0"] if r is None else r) / h["r0"])
    Cphi0 = np.asarray(h.get("Cphi0", h["Cphi"]), float)
    Cphi = Cphi0 * (h["Aphi0"] if Aphi is None else Aphi) / h["Aphi0"]
    Dof = lambda C: ((C + (h["Cn"].scale(2) + h["s2len"])) @ C.pinv()).sqrt()
    D = Dof(Cf)
    if r is not None:
        D0 = Dof(h["Cfs"] + h["Cten"])
        logdet_mix += (D0.pinv() @ D).logdet(proj)
    Nphi = np.asarray(h["Nphi"], float)
    G = np.ones_like(Cphi0) if "G_user" not in h else np.asarray(h["G_user"], float)
    if Aphi is not None:
        g0 = np.sqrt(1 + 2 * Nphi * _pinv(Cphi0))
        G = _pinv(g0) * np.sqrt(1 + 2 * Nphi * _pinv(Cphi))
        logdet_mix += HarmOp([G]).logdet(proj)
    precond = Cf.pinv() + (h["B"].T() @ h["Mf"].T() @ h["Cn"].pinv() @ h["Mf"] @ h["B"])
    out = dict(Cf_inv=Cf.pinv().p, D=D.p, D_inv=D.pinv().p, precond_inv=precond.pinv().p, Cphi_inv=_pinv(Cphi)[None], G_inv=_pinv(G)[None])
    logdet_sum = Cf.logdet(proj) + h["Cn"].logdet(proj) + HarmOp([Cphi]).logdet(proj)
    return out, logdet_sum, logdet_mix, dict(Cf=Cf, Cphi=Cphi, D=D, G=G, precond=precond)


def set_theta(ds, r=None, Aphi=None):
    """Evaluate the dataset's ParamDependentOps at θ = (r, Aϕ) (`ds(θ)`, src/dataset.jl:23-31) and make them current on the device;
    a parameter left None is 'not named in θ': its operators stay fiducial and its logdet term is 0.  `set_theta(ds)` restores the
    fiducial dataset."""
    h = ds.host
    h.setdefault("Cphi0", np.asarray(h["Cphi"], float).copy())
    ops, logdet_sum, logdet_mix, host = _ops(ds, r, Aphi)
    for k, v in ops.items():
        ds.set_op(k, v)
    ds.set_logdet(logdet_sum)
    ds.logdet_mix = float(logdet_mix)
    h.update(host)
    ds.L.invalidate()
    ds.theta = dict(r=r, Aphi=Aphi)


def logpdf_mixed_theta(ds, fo, po, r=None, Aphi=None):
    """logpdf(Mixed(ds); f°, ϕ°, θ) (src/dataset.jl:84-87), per batch slot; leaves the dataset at θ"""
    set_theta(ds, r, Aphi)
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
