"""MAP_joint step (coordinate descent + line search) restated from src/maximization.jl:116-233.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The 1-D minimiser is SciPy's bounded Brent (`fminbound`); the
reference uses Optim.jl's Brent (src/maximization.jl:195-199) -- only the converged minimiser (to abs_tol = αtol)
is comparable, the iterate sequence is not pinned.
"""
import numpy as np
from scipy.optimize import minimize_scalar

from .flatsky import pinv, dot_fourier

__all__ = ["map_joint_step", "map_joint", "map_marg"]


def map_joint_step(ds, phi_l, fstart=None, alpha_prev=1.0, alpha_tol=1e-4, alpha_max=None, cg_tol=1e-1, cg_nsteps=500):
    """One iteration of the MAP_joint loop body (:160-206) at θ = fiducial, G = I (:146).
    Returns dict(f, phi, f_mixed, phi_mixed, grad_phi, dphi, alpha, logpdf, cg_hist)."""
    G_save = ds.G
    ds.G = np.ones_like(ds.Cphi)
    try:
        f, hist = ds.argmaxf_logpdf(phi_l, fstart=fstart, tol=cg_tol, nsteps=cg_nsteps)              # :164-169
        fo, po = ds.mix(f, phi_l)                                                                   # :176
        lp0, gfo, gpo = ds.grad_logpdf_mixed(fo, po)                                                # :178 (∇ wrt ϕ° only is used)
        Hinv = pinv(pinv(ds.Cphi) + pinv(ds.Nphi))                                                  # dataset.jl:134-137
        dphi = Hinv * gpo                                                                           # :188
        amax = 2 * alpha_prev if alpha_max is None else alpha_max                                   # :193
        neg = lambda a: -float(np.sum(ds.logpdf_mixed(fo, po + a * dphi)))
        res = minimize_scalar(neg, bounds=(0.0, amax), method="bounded", options=dict(xatol=alpha_tol))   # :194-199
        alpha = float(res.x)
        po2 = po + alpha * dphi                                                                     # :201
        lp = ds.logpdf_mixed(fo, po2)                                                               # :205
        f2, phi2 = ds.unmix(fo, po2)                                                                # :206
        return dict(f=f, phi=phi2, f_mixed=fo, phi_mixed=po2, grad_phi=gpo, dphi=dphi, alpha=alpha, logpdf=lp, logpdf_before=lp0,
                    cg_hist=hist, dphi_norm=float(np.sqrt(np.sum(dot_fourier(ds.proj, dphi, dphi)))))
    finally:
        ds.G = G_save


def map_joint(ds, nsteps=3, phi_start=None, **kw):
    """`MAP_joint(ds; nsteps)` from ϕ = 0 (:119), carrying f (fstart) and α across steps (:226)."""
    phi = np.zeros((ds.d.shape[0], 1, ds.proj.Nx, ds.proj.Nyh), dtype=ds.d.dtype) if phi_start is None else phi_start
    f, alpha, hist = None, 1.0, []
    for _ in range(nsteps):
        st = map_joint_step(ds, phi, fstart=f, alpha_prev=alpha, **kw)
        f, phi, alpha = st["f"], st["phi"], st["alpha"]
        hist.append(dict(logpdf=st["logpdf"], alpha=alpha, ncg=len(st["cg_hist"]), dphi_norm=st["dphi_norm"]))
    return f, phi, hist


def map_marg(ds, white_f, white_n, nsteps=10, nsteps_with_meanfield_update=4, alpha=0.2, sims_per_batch=1, phi_start=None,
             cg_tol=1e-1, cg_nsteps=500):
    """`MAP_marg(ds)` (src/maximization.jl:245-343) at θ = fiducial.  white_f / white_n: (Nsims,P,Nx,Ny) unit white maps, the SAME
    draws every step (`_rng = copy(rng)`, :283).  ϕ ← ϕ + α Hϕ⁻¹ (g_data − ḡ − Cϕ⁻¹ϕ) (:319-322), Hϕ⁻¹ = pinv(Cϕ⁻¹ + Nϕ⁻¹) (:268);
    ḡ = mean over sims of ∂logpdf/∂ϕ at their Wiener-filtered f (:304-311), refreshed in the first
    `nsteps_with_meanfield_update` steps, CG warm-started from the previous step (:303).  `sims_per_batch` sims are filtered as
    batch slots of one CG (1 = the reference).  Returns (ϕ, trace)."""
    proj = ds.proj
    Nsims = white_f.shape[0]
    Hinv = pinv(pinv(ds.Cphi) + pinv(ds.Nphi))
    phi = np.zeros((1, 1, proj.Nx, proj.Nyh), dtype=ds.d.dtype) if phi_start is None else phi_start
    batches = [list(range(i, min(i + sims_per_batch, Nsims))) for i in range(0, Nsims, sims_per_batch)]
    f_prev, f_prev_sims, gbar, trace = None, [None] * len(batches), None, []

    def gmap(d, fprev):
        f_wf, hist = ds.argmaxf_logpdf(phi, d=d, fstart=fprev, tol=cg_tol, nsteps=cg_nsteps)
        return ds.gradientphi_logpdf(f_wf, phi, d=d), f_wf, hist

    for step in range(1, nsteps + 1):
        g_data, f_prev, hist = gmap(ds.d, f_prev)
        ncg = [len(hist)]
        if step <= nsteps_with_meanfield_update:
            tot = 0
            for k, ids in enumerate(batches):
                d_sim = ds.simulate_data(phi, white_f[ids], white_n[ids])
                g, f_prev_sims[k], hist = gmap(d_sim, f_prev_sims[k])
                tot = tot + g.sum(axis=0, keepdims=True)
                ncg.append(len(hist))
            gbar = tot / Nsims
        with np.errstate(divide="ignore", invalid="ignore"):
            g = g_data - gbar - np.nan_to_num(phi / ds.Cphi, nan=0.0, posinf=0.0, neginf=0.0)
        phi = phi + alpha * Hinv * g
        trace.append(dict(step=step, g_norm=float(np.sqrt(np.sum(dot_fourier(proj, g, g)))), ncg=ncg, phi=phi))
    return phi, trace
