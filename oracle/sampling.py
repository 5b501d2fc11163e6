"""HMC / Gibbs passes of sample_joint, restated from src/sampling.jl:14-46,388-464 and src/maximization.jl:56-62.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Randomness is injected (momenta, uniforms, simulation white noise) so that
the device implementation can be driven with identical draws.
"""
import numpy as np

from .flatsky import pinv, rfft2, dot_fourier

__all__ = ["symplectic_integrate", "hmc_step", "mass_matrix_phi", "sample_f", "gibbs_step"]


def symplectic_integrate(x0, p0, Lam, U, dUdx, N=50, eps=0.1, dot=None):
    """src/sampling.jl:14-46.  Λ is a diagonal (array); H(x,p) = U(x) − p·(Λ\\p)/2.  Returns (ΔH, x, p)."""
    H = lambda x, p: U(x) - dot(p, pinv(Lam) * p) / 2
    x, p = x0, p0
    g = dUdx(x)
    for _ in range(N):
        x1 = x - eps * (pinv(Lam) * (p - eps / 2 * g))
        g1 = dUdx(x1)
        p = p - eps / 2 * (g1 + g)
        x, g = x1, g1
    return H(x, p) - H(x0, p0), x, p


def mass_matrix_phi(ds):
    """src/sampling.jl:422-425: pinv(G)^2 (pinv(Cϕ) + pinv(Nϕ))"""
    return pinv(ds.G) ** 2 * (pinv(ds.Cphi) + pinv(ds.Nphi))


def hmc_step(ds, fo, po, white_p, log_u, N=25, eps=0.01, always_accept=False, alias_quirk=False):
    """One pass of `hmc_step` (src/sampling.jl:405-418) over ϕ° with U = logpdf(Mixed(ds); f°, ϕ°).
    white_p: white-noise map (B,1,Nx,Ny) for the momentum p = sqrt(Λ)·rfft(white); log_u: log of the uniform draws (B,)."""
    proj = ds.proj
    Lam = mass_matrix_phi(ds)
    p0 = np.sqrt(Lam) * rfft2(white_p)
    U = lambda x: ds.logpdf_mixed(fo, x)
    dU = lambda x: ds.grad_logpdf_mixed(fo, x, alias_quirk=alias_quirk)[2]
    dot = lambda a, b: dot_fourier(proj, a, b)
    dH, xt, _ = symplectic_integrate(po, p0, Lam, U, dU, N=N, eps=eps, dot=dot)
    accept = np.logical_or(always_accept, log_u < dH)
    x = np.where(accept.reshape(-1, 1, 1, 1), xt, po)
    return x, dH, accept


def sample_f(ds, phi_l, white_f, white_n, fstart=None, tol=1e-1, nsteps=500):
    """`sample_f` (src/maximization.jl:56-62): sim = simulate(ds; ϕ); Δf = argmaxf(d − sim.d; offset) ; f = sim.f + Δf"""
    L = ds.L(phi_l)
    fs = ds.Cf.sqrt()(rfft2(white_f))
    ns = ds.Cn.sqrt()(rfft2(white_n))
    dsim = ds.mean(L, fs) + ns
    df, hist = ds.argmaxf_logpdf(phi_l, d=ds.d - dsim, fstart=fstart, tol=tol, nsteps=nsteps, offset=True)
    return fs + df, hist


def gibbs_step(ds, phi_l, white_f, white_n, white_p, log_u, N=25, eps=0.01, always_accept=False):
    """One sample_joint step at fixed θ (src/sampling.jl:187-193): sample f | ϕ ; mix ; HMC ϕ° | f° ; unmix ; postprocess."""
    f, hist = sample_f(ds, phi_l, white_f, white_n)
    fo, po = ds.mix(f, phi_l)
    po2, dH, accept = hmc_step(ds, fo, po, white_p, log_u, N=N, eps=eps, always_accept=always_accept)
    f2, phi2 = ds.unmix(fo, po2)
    lp = ds.logpdf(f2, phi2)
    return dict(f=f2, phi=phi2, dH=dH, accept=accept, logpdf=lp, cg_hist=hist)
