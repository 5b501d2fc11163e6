"""Oracle pinning, part 2: LenseFlow (test/runtests.jl:533-581) and the posterior
(test/runtests.jl:585-621) at the reference's sizes and tolerances, plus tighter checks."""
import numpy as np
import pytest

import oracle as O
from oracle.lenseflow import LenseFlow

NSIDES_BIG = [(128, 128), (64, 128), (128, 64)]


def _sims(camb, Ny, Nx, P, T):
    proj = O.Proj(Ny, Nx, 1.0, T)                       # ProjLambert default θpix = 1 (runtests.jl:546)
    cl = camb["unlensed_total"]
    Cphi = O.cl_to_2d(cl["pp"], proj)
    C = (O.cl_to_2d(cl["TT"], proj)[None] if P == 1
         else np.stack([O.cl_to_2d(cl["EE"], proj), O.cl_to_2d(cl["BB"], proj)]))
    simf = lambda seed: O.from_harm(proj, (np.sqrt(C) * O.rfft2(O.white_noise(seed, (1, P, Nx, Ny), T))))
    simp = lambda seed: O.irfft2(np.sqrt(Cphi) * O.rfft2(O.white_noise(seed, (1, 1, Nx, Ny), T)), Ny).astype(T)
    return proj, simf, simp


@pytest.mark.parametrize("Ny,Nx", NSIDES_BIG)
@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("P", [1, 2])
def test_lenseflow_adjoint_and_gradient(camb, Ny, Nx, T, P):
    proj, simf, simp = _sims(camb, Ny, Nx, P, T)
    phi, dphi = simp(2), simp(5)
    f, g, df = simf(1), simf(11), simf(4)
    L = LenseFlow(proj, phi, 7)
    Lg = L.apply(g)
    assert Lg.dtype == T and Lg.shape == g.shape
    # adjoint identity  f'(Lϕ g) ≈ (f'Lϕ) g   (runtests.jl:556, 570 rtol = sqrt(eps(T)))
    lhs = O.dot_map(f, Lg)[0]
    rhs = O.dot_fourier(proj, L.adj(O.rfft2(f)), O.rfft2(g))[0]
    assert abs(lhs - rhs) <= np.sqrt(np.finfo(T).eps) * abs(lhs)
    # inverse round trips
    assert np.sqrt(O.dot_map(L.inv(Lg) - g, L.inv(Lg) - g) / O.dot_map(g, g))[0] < 2e-5
    gl = O.rfft2(g)
    back = L.invadj(L.adj(gl))
    assert np.sqrt(O.dot_fourier(proj, back - gl, back - gl) / O.dot_fourier(proj, gl, gl))[0] < 5e-3
    if T == np.float32:
        return
    # directional gradient of α ↦ ‖L(ϕ+αδϕ)(f+αδf)‖ at α=0 vs central differences (runtests.jl:559,573; atol 0.2)
    fun = lambda a: np.sqrt(O.dot_map(*(2 * [LenseFlow(proj, phi + a * dphi, 7).apply(f + a * df)])))[0]
    eps = 0.01
    fd = (fun(-2 * eps) - 8 * fun(-eps) + 8 * fun(eps) - fun(2 * eps)) / (12 * eps)
    ft = L.apply(f)
    delta = O.rfft2(ft / np.sqrt(O.dot_map(ft, ft)))
    for quirk, tol in ((False, 1e-5 * abs(fd) + 1e-6), (True, 0.2)):
        f0, gf, gp = L.grad_apply(ft, delta, alias_quirk=quirk)
        an = (O.dot_fourier(proj, gf, O.rfft2(df)) + O.dot_fourier(proj, gp, O.rfft2(dphi)))[0]
        assert abs(an - fd) < tol, (quirk, an, fd)
        # the δ-flow also carries f̃ back to f
        assert np.sqrt(O.dot_map(f0 - f, f0 - f) / O.dot_map(f, f))[0] < 2e-5
    # with ϕ held constant the pullback is the plain adjoint flow (flowops.jl:45-46)
    np.testing.assert_allclose(gf, L.adj(delta), rtol=0, atol=1e-9 * np.abs(gf).max())


def test_grad_inv_matches_fd(camb):
    proj, simf, simp = _sims(camb, 64, 128, 2, np.float64)
    phi, dphi, f, df = simp(2), simp(5), simf(1), simf(4)
    L = LenseFlow(proj, phi, 7)
    fun = lambda a: np.sqrt(O.dot_map(*(2 * [LenseFlow(proj, phi + a * dphi, 7).inv(f + a * df)])))[0]
    eps = 0.01
    fd = (fun(-2 * eps) - 8 * fun(-eps) + 8 * fun(eps) - fun(2 * eps)) / (12 * eps)
    fi = L.inv(f)
    _, gf, gp = L.grad_inv(fi, O.rfft2(fi / np.sqrt(O.dot_map(fi, fi))))
    an = (O.dot_fourier(proj, gf, O.rfft2(df)) + O.dot_fourier(proj, gp, O.rfft2(dphi)))[0]
    assert abs(an - fd) < 1e-5 * abs(fd) + 1e-6


def test_batched_flow_equals_loop(camb):
    # batching along dim 4 (src/proj_lambert.jl:446-459): every slot independent
    proj, simf, simp = _sims(camb, 64, 64, 2, np.float64)
    phis = np.concatenate([simp(2), simp(3)], axis=0)
    fs = np.concatenate([simf(1), simf(7)], axis=0)
    Lb = LenseFlow(proj, phis, 7)
    out = Lb.apply(fs)
    for b in range(2):
        np.testing.assert_allclose(out[b:b + 1], LenseFlow(proj, phis[b:b + 1], 7).apply(fs[b:b + 1]), rtol=0, atol=1e-10)
    # one ϕ, many f
    out = LenseFlow(proj, phis[:1], 7).apply(fs)
    np.testing.assert_allclose(out[1:], LenseFlow(proj, phis[:1], 7).apply(fs[1:]), rtol=0, atol=1e-10)


@pytest.mark.parametrize("Nside", NSIDES_BIG)
@pytest.mark.parametrize("pol", ["I", "P", "IP"])
def test_posterior(Nside, pol):
    # runtests.jl:585-621: LenseFlow(7), Float64, θpix=3, beamFWHM=3, border mask (edge padding 1°)
    s = O.load_sim(3.0, Nside, pol, np.float64, beam_fwhm=3.0, pixel_mask=dict(pad_deg=1.0, apod_deg=0.6))
    ds, proj, f, phi = s["ds"], s["proj"], s["f"], s["phi"]
    fo, po = ds.mix(f, phi)
    lp, lpm = ds.logpdf(f, phi)[0], ds.logpdf_mixed(fo, po)[0]
    assert abs(lp - lpm) <= 3e-4 * abs(lp)                                     # :609
    P = ds.P
    df = ds.Cf.sqrt()(O.rfft2(O.white_noise(4, (1, P, proj.Nx, proj.Ny), np.float64)))
    dp = np.sqrt(ds.Cphi) * O.rfft2(O.white_noise(5, (1, 1, proj.Nx, proj.Ny), np.float64))
    dfm = O.from_harm(proj, df)
    atol = 30 if pol == "IP" else 3                                            # :613
    eps = 1e-3
    fun = lambda a: ds.logpdf_mixed(fo + a * dfm, po + a * dp)[0]
    fd = (fun(-2 * eps) - 8 * fun(-eps) + 8 * fun(eps) - fun(2 * eps)) / (12 * eps)
    for quirk in (False, True):
        lp2, gfo, gpo = ds.grad_logpdf_mixed(fo, po, alias_quirk=quirk)
        assert abs(lp2[0] - lpm) < 1e-9 * abs(lpm)
        an = (O.dot_map(gfo, dfm) + O.dot_fourier(proj, gpo, dp))[0]
        assert abs(an - fd) < atol, (quirk, an, fd)                            # :615
    # the f°-part alone is an exact transpose (discrete adjoint): tight check
    funf = lambda a: ds.logpdf_mixed(fo + a * dfm, po)[0]
    fdf = (funf(-2 * eps) - 8 * funf(-eps) + 8 * funf(eps) - funf(2 * eps)) / (12 * eps)
    assert abs(O.dot_map(gfo, dfm)[0] - fdf) < 1e-6 * abs(fdf) + 1e-5


def test_gradientf_logpdf_and_wiener_filter():
    # dataset.jl:76-80 is the f-gradient of logpdf(ds;…); maximization.jl:17-42 solves it to zero
    s = O.load_sim(3.0, (64, 64), "P", np.float64, beam_fwhm=3.0, pixel_mask=dict(pad_deg=0.5, apod_deg=0.5))
    ds, proj, f, phi = s["ds"], s["proj"], s["f"], s["phi"]
    L = ds.L(phi)
    g = ds.gradientf_logpdf(f, L, ds.d)
    df = ds.Cf.sqrt()(O.rfft2(O.white_noise(4, (1, 2, 64, 64), np.float64)))
    fun = lambda a: ds.logpdf(f + a * df, phi)[0]
    eps = 1e-2
    fd = (fun(-2 * eps) - 8 * fun(-eps) + 8 * fun(eps) - fun(2 * eps)) / (12 * eps)
    an = ds.dot(g, df)[0]
    assert abs(an - fd) < 1e-7 * abs(fd) + 1e-6
    fwf, hist = ds.argmaxf_logpdf(phi, tol=1e-1, nsteps=500)
    assert hist[-1][1][0] < 1e-1 and len(hist) < 500
    res = [h[1][0] for h in hist]
    assert res[-1] < 1e-3 * res[0]
    # a tighter solve really is the maximiser: gradient ≈ 0 relative to the rhs
    fwf2, hist2 = ds.argmaxf_logpdf(phi, tol=1e-10, nsteps=500)
    g = ds.gradientf_logpdf(fwf2, L, ds.d)
    b = ds.gradientf_logpdf(np.zeros_like(f), L, ds.d)
    assert ds.dot(g, g)[0] < 1e-8 * ds.dot(b, b)[0]
    # Wiener-filtered map correlates with the truth
    r = ds.dot(fwf, f)[0] / np.sqrt(ds.dot(fwf, fwf)[0] * ds.dot(f, f)[0])
    assert r > 0.5


# ---------------------------------------------------------------------------------------------------------------------
# Independent pin of the conventions: LenseFlow(ϕ)*f against the exact remapping f(x + ∇ϕ(x)) evaluated by direct Fourier
# summation (tests/_known.py).  The reference's own properties (adjoint identity, FD gradients, round trips) are blind to a
# global sign / convention error such as f(x − ∇ϕ); this one is not (the wrong sign gives an O(1) error).
def _remap_case(Nx, Ny, P, rms_pix, theta=2.0):
    from _known import bandlimited, remap_exact, deflection
    f = bandlimited(1, Nx, Ny, 0.35, 1.5, (1, P))
    phi0 = bandlimited(2, Nx, Ny, 0.25, 3.0, ())
    ax, ay = deflection(phi0, np.deg2rad(theta / 60))
    phi = phi0 * rms_pix / np.sqrt(np.mean(ax ** 2 + ay ** 2))
    want, rms = remap_exact(f, phi, theta, +1.0)
    wrong, _ = remap_exact(f, phi, theta, -1.0)
    assert abs(rms - rms_pix) < 1e-6 * rms_pix
    return f, phi[None, None], want, wrong


@pytest.mark.parametrize("Ny,Nx,P", [(32, 32, 1), (64, 32, 2), (32, 64, 1)])
def test_lenseflow_is_the_exact_remap(Ny, Nx, P):
    f, phi, want, wrong = _remap_case(Nx, Ny, P, 0.55)
    proj = O.Proj(Ny, Nx, 2.0, np.float64)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    for n, tol in ((7, 3e-5), (10, 3e-5)):                 # measured 3e-6 .. 8e-6: the floor is aliasing of the (not band-limited) lensed field
        L = LenseFlow(proj, phi, n)
        got = L.apply(f)
        assert rel(got, want) < tol, (n, rel(got, want))
        assert rel(got, wrong) > 0.1                       # f(x − ∇ϕ) is NOT what the flow computes
        # the inverse flow undoes the exact remap
        assert rel(L.inv(want), f) < 3 * tol
        # adjoint against the exact remap:  <g, remap(f)> = <L'g, f>
        g = np.random.default_rng(3).standard_normal(f.shape)
        lhs = O.dot_map(g, want)[0]
        rhs = O.dot_fourier(proj, L.adj(O.rfft2(g)), O.rfft2(f))[0]
        assert abs(lhs - rhs) < 30 * tol * np.sqrt(O.dot_map(g, g)[0] * O.dot_map(want, want)[0])
