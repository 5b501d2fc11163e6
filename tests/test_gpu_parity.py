r"""GPU parity: every C-ABI entry point of libcmblens_hip.so against the NumPy oracle on identical inputs.

Tolerances (relative L2 per field; fp32 against the float64 oracle on the fp32-rounded inputs).  Two layers (tests/_tol.py):
  * the class bounds in TOL below = 3 x the LARGEST error that class showed on MI355X (profiles/r04_parity_measured.txt):
      fp32: transforms 5e-7 -> 1.5e-6; forward-type flows (L*f, L\f, the f part of a pullback) 6e-6 -> 2e-5; adjoint-type flows (L'g, L'\g,
            the δf part: Fourier-space state, 7x less accurate) 2.3e-5 (3.4e-5 on the any-size path) -> 5e-5 / 1e-4; δϕ 5.8e-5 (7.3e-5) -> 1.8e-4
      fp64: 1e-12 transforms / pointwise, 1e-10 flows, 1e-9 gradients (measured 1e-16..3e-11; the largest are the 4096-point rows)
  * every single comparison is additionally held to 3 x ITS OWN measured error (tests/golden/parity_measured.json).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as O                                   # the checker (tests only)
from oracle.lenseflow import LenseFlow as OLenseFlow


def _pkg():
    import cmblensing_jl_amd as C
    return C


from _tol import close, rel, scalars_close            # assertions that log what they measured (tests/_tol.py)


DT = {"f32": (torch.float32, np.float32), "f64": (torch.float64, np.float64)}
LPTOL = {"f32": 5e-8, "f64": 1e-10}      # logpdf: fp32 terms, float64 sums; measured 1.3e-8 (64²) .. 1.8e-8 (1024² T+QU)
TOL = {"f32": dict(fft=1.5e-6, flow=2e-5, adj=5e-5, grad=1.8e-4, cg=1e-3), "f64": dict(fft=1e-12, flow=1e-10, adj=1e-10, grad=1e-9, cg=1e-7)}


@pytest.fixture(scope="module")
def camb():
    return O.load_camb()


def sims(camb, Ny, Nx, P, B=1, theta=2.0):
    """float64 CMB-like inputs: f (map), g (map), phi (map)"""
    proj = O.Proj(Ny, Nx, theta, np.float64)
    cl = camb["unlensed_total"]
    Cphi = O.cl_to_2d(cl["pp"], proj)
    if P == 1:
        C = O.cl_to_2d(cl["TT"], proj)[None]
    elif P == 2:
        C = np.stack([O.cl_to_2d(cl["EE"], proj), O.cl_to_2d(cl["BB"], proj) + 0.05 * O.cl_to_2d(cl["EE"], proj)])
    else:
        C = np.stack([O.cl_to_2d(cl["TT"], proj), O.cl_to_2d(cl["EE"], proj), 0.05 * O.cl_to_2d(cl["EE"], proj)])
    simf = lambda seed: O.from_harm(proj, np.sqrt(C) * O.rfft2(O.white_noise(seed, (B, P, Nx, Ny), np.float64)))
    simp = lambda seed, b=B: O.irfft2(np.sqrt(Cphi) * O.rfft2(O.white_noise(seed, (b, 1, Nx, Ny), np.float64)), Ny)
    return proj, simf, simp


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("Ny,Nx", [(64, 128), (128, 64), (256, 256), (32, 32)])
def test_geometry_and_basis_transforms(prec, Ny, Nx):
    C = _pkg()
    tT, nT = DT[prec]
    p = C.ProjLambert(Ny, Nx, 3.0, tT)
    op = O.Proj(Ny, Nx, 3.0, nT)
    np.testing.assert_allclose(p.lx, op.lx, rtol=1e-6)
    np.testing.assert_allclose(p.ly, op.ly, rtol=1e-6)
    np.testing.assert_allclose(p.lam, op.lam)
    np.testing.assert_allclose(p.sin2phi, op.sin2phi, atol=2e-6)
    np.testing.assert_allclose(p.cos2phi, op.cos2phi, atol=2e-6)
    rng = np.random.default_rng(4)
    for P, B in [(1, 1), (2, 2), (3, 1)]:
        m = rng.standard_normal((B, P, Nx, Ny)).astype(nT)
        fl = p.rfft(p.tensor(m))
        ref = O.rfft2(m.astype(np.float64))
        close(("rfft", P, B), fl.cpu().numpy(), ref, TOL[prec]["fft"])
        back = p.irfft(fl)
        close(("irfft∘rfft", P, B), back.cpu().numpy(), m, TOL[prec]["fft"])
        # irfft of NON-Hermitian input must follow FFTW/pocketfft semantics (Im of ky=0,Nyq after the x pass dropped)
        junk = (rng.standard_normal(ref.shape) + 1j * rng.standard_normal(ref.shape))
        out = p.irfft(p.tensor(junk))
        close(("irfft non-hermitian", P, B), out.cpu().numpy(), O.irfft2(junk, Ny), TOL[prec]["fft"])
        # the whole basis lattice (src/proj_lambert.jl:245-300)
        oproj = O.Proj(Ny, Nx, 3.0, np.float64)
        h = p.convert(p.tensor(m), C.MAP, C.HARMONIC)
        close("h.cpu().numpy()", h.cpu().numpy(), O.to_harm(oproj, m.astype(np.float64)), 3 * TOL[prec]["fft"] + 1e-6 * (prec == "f32"))
        q = p.convert(h, C.HARMONIC, C.FOURIER)
        close("q.cpu().numpy()", q.cpu().numpy(), ref, 3 * TOL[prec]["fft"] + 1e-6 * (prec == "f32"))
        mm = p.convert(h, C.HARMONIC, C.MAP)
        close("mm.cpu().numpy()", mm.cpu().numpy(), m, 3 * TOL[prec]["fft"] + 1e-6 * (prec == "f32"))


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_reductions_and_diag_ops(prec):
    C = _pkg()
    tT, nT = DT[prec]
    Ny, Nx = 128, 64
    p = C.ProjLambert(Ny, Nx, 2.0, tT)
    op = O.Proj(Ny, Nx, 2.0, np.float64)
    rng = np.random.default_rng(1)
    tol = 1e-5 if prec == "f32" else 1e-12
    for P, B in [(1, 2), (2, 3), (3, 1)]:
        a = rng.standard_normal((B, P, Nx, Ny)).astype(nT)
        b = rng.standard_normal((B, P, Nx, Ny)).astype(nT)
        np.testing.assert_allclose(p.dot(p.tensor(a), p.tensor(b), C.MAP), O.dot_map(a.astype(float), b.astype(float)), rtol=tol, atol=tol * a.size ** 0.5)
        al, bl = O.rfft2(a.astype(float)), O.rfft2(b.astype(float))
        got = p.dot(p.tensor(al), p.tensor(bl), C.FOURIER)
        np.testing.assert_allclose(got, O.dot_fourier(op, al, bl), rtol=10 * tol, atol=tol * a.size ** 0.5)
        np.testing.assert_allclose(got, O.dot_map(a.astype(float), b.astype(float)), rtol=10 * tol, atol=10 * tol * a.size ** 0.5)   # Parseval
        # DiagOp * and \ in the harmonic basis applied to a map, result as map (src/specialops.jl:9-10)
        d = (rng.random((P, Nx, Ny // 2 + 1)) + 0.5).astype(nT)
        d[0, 0, 0] = 0                                       # exercises nan2zero
        fh = O.to_harm(op, a.astype(float))
        want = O.from_harm(op, d.astype(float) * fh)
        got = p.diag_apply(d, p.tensor(a), C.HARMONIC, C.MAP, C.MAP)
        close("got.cpu().numpy()", got.cpu().numpy(), want, (5e-6 if prec == "f32" else 1e-12))
        want = O.from_harm(op, O.diag_div(d.astype(float), fh))
        got = p.diag_apply(d, p.tensor(a), C.HARMONIC, C.MAP, C.MAP, kind=3)
        close("got.cpu().numpy()", got.cpu().numpy(), want, (5e-6 if prec == "f32" else 1e-12))
        ld = p.logdet(d)
        np.testing.assert_allclose(ld, O.logdet_fourier(op, d.astype(float)[None])[0], rtol=1e-6 if prec == "f32" else 1e-12)
    # BlockDiagIEB (src/specialops.jl:80-83)
    te = (rng.random((5, Nx, Ny // 2 + 1)) + 0.5).astype(nT)
    a = rng.standard_normal((2, 3, Nx, Ny)).astype(nT)
    H = O.HarmOp(3, te=tuple(te[:4].astype(float)), bb=te[4].astype(float))
    want = H(O.to_harm(op, a.astype(float)))
    got = p.diag_apply(te, p.tensor(a), C.HARMONIC, C.MAP, C.HARMONIC)
    close("got.cpu().numpy()", got.cpu().numpy(), want, (5e-6 if prec == "f32" else 1e-12))


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("Ny,Nx,P,B,Bphi", [(128, 128, 1, 1, 1), (64, 128, 2, 1, 1), (128, 64, 3, 1, 1), (64, 64, 2, 3, 3),
                                            (64, 64, 2, 2, 1), (256, 256, 2, 1, 1)])
@pytest.mark.parametrize("n", [7, 10])
def test_lenseflow_ops(camb, prec, Ny, Nx, P, B, Bphi, n):
    C = _pkg()
    tT, nT = DT[prec]
    oproj, simf, simp = sims(camb, Ny, Nx, P, B)
    f = simf(1).astype(nT).astype(np.float64)
    g = simf(11).astype(nT).astype(np.float64)
    phi = simp(2, Bphi).astype(nT).astype(np.float64)
    OL = OLenseFlow(oproj, phi, n)
    p = C.ProjLambert(Ny, Nx, 2.0, tT)
    L = C.LenseFlow(p, n)(C.Field(p, p.tensor(phi), C.MAP))
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    tol = TOL[prec]["flow"]
    out = (L * F(f, C.MAP)).arr.cpu().numpy()
    close("L*f", out, OL.apply(f), tol)
    out = L.ldiv(F(f, C.MAP)).arr.cpu().numpy()
    close("L\\f", out, OL.inv(f), tol)
    gl = O.rfft2(g)
    out = (L.adjoint * F(gl, C.FOURIER)).arr.cpu().numpy()
    close("L'*g", out, OL.adj(gl), TOL[prec]["adj"])
    out = L.adjoint.ldiv(F(gl, C.FOURIER)).arr.cpu().numpy()
    close("L'\\g", out, OL.invadj(gl), TOL[prec]["adj"])
    # basis plumbing: harmonic in, map out == explicit conversion
    fh = O.to_harm(oproj, f)
    out = L._apply(C.FLOW_FWD, F(fh, C.HARMONIC), C.HARMONIC).arr.cpu().numpy()
    close("out", out, O.to_harm(oproj, OL.apply(f)), 2 * tol)
    # adjoint identity on the device itself (test/runtests.jl:556,570)
    lhs = p.dot(p.tensor(f), (L * F(g, C.MAP)).arr, C.MAP)
    rhs = (L.adjoint * F(O.rfft2(f), C.FOURIER)).dot(F(gl, C.FOURIER))
    scalars_close("adjoint identity", lhs, rhs, rtol=2.5e-5 if prec == "f32" else 1e-10)              # measured up to 8.3e-6 (a batch slot whose dot nearly cancels)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("Ny,Nx,P,B,Bphi", [(128, 128, 1, 1, 1), (64, 128, 2, 1, 1), (128, 64, 3, 1, 1), (64, 64, 2, 2, 2), (64, 64, 2, 2, 1)])
@pytest.mark.parametrize("mode", ["fwd", "inv"])
@pytest.mark.parametrize("n", [7, 10])
def test_lenseflow_gradient(camb, prec, Ny, Nx, P, B, Bphi, mode, n):
    C = _pkg()
    tT, nT = DT[prec]
    oproj, simf, simp = sims(camb, Ny, Nx, P, B)
    f = simf(1).astype(nT).astype(np.float64)
    phi = simp(2, Bphi).astype(nT).astype(np.float64)
    OL = OLenseFlow(oproj, phi, n)
    fe = OL.apply(f) if mode == "fwd" else OL.inv(f)
    fe = fe.astype(nT).astype(np.float64)
    delta = O.rfft2(simf(7)).astype(np.complex64 if prec == "f32" else np.complex128).astype(np.complex128)
    p = C.ProjLambert(Ny, Nx, 2.0, tT)
    L = C.LenseFlow(p, n)(C.Field(p, p.tensor(phi), C.MAP))
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    for quirk in (False, True):
        f0, df, dp = (OL.grad_apply if mode == "fwd" else OL.grad_inv)(fe, delta, alias_quirk=quirk)
        gdp, gdf, gf0 = L.gradient(C.FLOW_FWD if mode == "fwd" else C.FLOW_INV, F(fe, C.MAP), F(delta, C.FOURIER), alias_quirk=quirk)
        close(("f", quirk), gf0.arr.cpu().numpy(), f0, TOL[prec]["flow"])
        close(("df", quirk), gdf.arr.cpu().numpy(), df, TOL[prec]["adj"])
        close(("dphi", quirk), gdp.arr.cpu().numpy(), dp, TOL[prec]["grad"])
    # the two variants must differ (the flag is live)
    a = L.gradient(C.FLOW_FWD, F(fe, C.MAP), F(delta, C.FOURIER), alias_quirk=False)[0].arr
    b = L.gradient(C.FLOW_FWD, F(fe, C.MAP), F(delta, C.FOURIER), alias_quirk=True)[0].arr
    assert rel(a.cpu().numpy(), b.cpu().numpy()) > 1e-6


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_lenseflow_gradient_many_steps(camb, prec):
    """n = 20 RK steps = 80 stages: beyond the 64 (t_s, c_s) pairs that ride in the kernel arguments of the δϕ quadrature, so the
    table goes through device memory; n = 16 is the last that does not"""
    for n in (16, 20):
        test_lenseflow_gradient(camb, prec, 64, 32, 2, 1, 1, "fwd", n)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("Ny,Nx,P", [(32, 32, 1), (64, 32, 2), (32, 64, 3)])
def test_lenseflow_is_the_exact_remap(prec, Ny, Nx, P, tol32=2.5e-5):
    """Independent known answer (tests/_known.py, no oracle involved): L(ϕ)*f = f(x + ∇ϕ(x)) by direct Fourier summation.
    Pins the deflection sign / axis conventions, which the reference's self-consistency properties cannot see."""
    from _known import bandlimited, remap_exact, deflection
    C = _pkg()
    tT, nT = DT[prec]
    theta = 2.0
    f = bandlimited(1, Nx, Ny, 0.35, 1.5, (1, P))
    phi0 = bandlimited(2, Nx, Ny, 0.25, 3.0, ())
    ax, ay = deflection(phi0, np.deg2rad(theta / 60))
    phi = phi0 * 0.55 / np.sqrt(np.mean(ax ** 2 + ay ** 2))             # 0.55 pixel rms deflection
    want, _ = remap_exact(f, phi, theta, +1.0)
    wrong, _ = remap_exact(f, phi, theta, -1.0)
    p = C.ProjLambert(Ny, Nx, theta, tT)
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    tol = tol32 if prec == "f32" else 2.5e-5                             # fp64 floor 3e-6..8e-6 = aliasing of the lensed field; fp32 measured 8.3e-6
    for n in (7, 10):
        L = C.LenseFlow(p, n)(F(phi[None, None], C.MAP))
        got = (L * F(f, C.MAP)).arr.cpu().numpy()
        close(("exact remap, n", n), got, want, tol)
        assert rel(got, wrong) > 0.1
        close(("L\\ of the exact remap, n", n), L.ldiv(F(want, C.MAP)).arr.cpu().numpy(), f, 3 * tol)
        g = np.random.default_rng(3).standard_normal(f.shape)
        lhs = p.dot(p.tensor(g), p.tensor(want), C.MAP)[0]
        rhs = (L.adjoint * F(g, C.MAP).to(C.FOURIER)).dot(F(f, C.MAP).to(C.FOURIER))[0]
        assert abs(lhs - rhs) < 30 * tol * np.linalg.norm(g) * np.linalg.norm(want)


def _dataset_pair(prec, pol, Nside, theta=3.0, mask=True, beam=3.0, B=1):
    """identical dataset on both sides: oracle (float64) and device (precision `prec`)"""
    C = _pkg()
    tT, nT = DT[prec]
    pm = dict(pad_deg=0.4, apod_deg=0.4) if mask else None
    so = O.load_sim(theta, Nside, pol, np.float64, beam_fwhm=beam, pixel_mask=pm, Nbatch=B)
    camb = so["cls"]
    cls = {g: {k: C.Cls(v.ell, v.cl) for k, v in camb[g].items()} for g in ("unlensed_scalar", "tensor", "total")}
    sd = C.load_sim(theta, Nside, pol, cls, T=tT, beam_fwhm=beam, pixel_mask=pm, Nbatch=B, Nphi=so["ds"].Nphi * 2)
    return C, so, sd


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol,Nside,mask", [("I", (64, 64), True), ("P", (64, 128), True), ("IP", (128, 64), True), ("P", (64, 64), False)])
def test_dataset_gradientf_and_wiener(prec, pol, Nside, mask):
    C, so, sd = _dataset_pair(prec, pol, Nside, mask=mask)
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    tol = TOL[prec]
    # the simulated fields agree (same PCG64 seeds, same operators)
    close("simulated f", sd["f"].arr.cpu().numpy(), so["f"], 8e-7 if prec == "f32" else 1e-11)               # measured 2.4e-7
    close("simulated phi", sd["phi"].arr.cpu().numpy(), so["phi"], 5e-7 if prec == "f32" else 1e-11)           # 1.4e-7
    close("simulated d", sd["d"].arr.cpu().numpy(), so["d"], 2.5e-6 if prec == "f32" else 1e-10)               # 7.1e-7
    # run both sides from the ORACLE's fields so that only the operator under test differs
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    f, phi, d = so["f"], so["phi"], so["d"]
    ds.set_data(F(d, C.HARMONIC))
    OL = ods.L(phi)
    want = ods.gradientf_logpdf(f, OL, d)
    got = ds.gradientf_logpdf(F(f, C.HARMONIC), F(phi, C.FOURIER))
    close("gradientf_logpdf", got.arr.cpu().numpy(), want, 1e-4 if prec == "f32" else 4e-10)                  # measured 3.4e-5
    # Wiener filter: same tolerance-based stop; compare solution and history loosely, tight solve tightly
    fw_o, h_o = ods.argmaxf_logpdf(phi, tol=1e-1, nsteps=500)
    fw_g, h_g = ds.argmaxf_logpdf(F(phi, C.FOURIER), tol=1e-1, nsteps=500)
    # iteration count depends on round-off (SURVEY §7 'CG reproducibility'): ±1 in fp64, within 5 % in fp32
    assert abs(len(h_g) - len(h_o)) <= (1 if prec == "f64" else max(2, len(h_o) // 20)), (len(h_g), len(h_o))
    n = min(len(h_g), len(h_o)) - 1
    scalars_close("cg first residual", h_g[0][1], h_o[0][1], rtol=2e-6 if prec == "f32" else 1e-8)            # measured 5.9e-7
    scalars_close("cg mid-run residual", h_g[n // 2][1], h_o[n // 2][1], rtol=9e-3 if prec == "f32" else 1e-5)   # measured 2.9e-3
    # hundreds of CG iterations amplify round-off (loss of conjugacy): converged solutions agree loosely ...
    close("cg converged solution", fw_g.arr.cpu().numpy(), fw_o, (3.9e-3 if prec == "f32" else 1e-4))       # measured 8.7e-4 (1.3e-3 any-size)
    # ... while a fixed, short run (no early stop) must agree tightly, iterate by iterate
    fw_o, h_o = ods.argmaxf_logpdf(phi, tol=0.0, nsteps=8)
    fw_g, h_g = ds.argmaxf_logpdf(F(phi, C.FOURIER), tol=0.0, nsteps=8)
    cgtol = dict(res=1e-3, x=4e-6) if prec == "f32" else dict(res=1e-8, x=1e-9)          # fp32 measured: residuals 3.2e-4, iterate 1.4e-6
    assert len(h_g) == len(h_o) == 8
    for (i, r_g), (_, r_o) in zip(h_g, h_o):
        scalars_close(("cg 8-step residual", i), r_g, r_o, rtol=cgtol["res"])
    close("cg 8-step iterate", fw_g.arr.cpu().numpy(), fw_o, cgtol["x"])
    # fstart (maximization.jl:26,37): restarting from the 8-step iterate continues to converge
    fw_g2, h_g2 = ds.argmaxf_logpdf(F(phi, C.FOURIER), fstart=fw_g, tol=1e-1, nsteps=500)
    assert h_g2[0][1][0] < h_g[0][1][0] and min(h[1][0] for h in h_g2) < h_g2[0][1][0]


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol,Nside", [("I", (64, 64)), ("P", (64, 128)), ("IP", (128, 64))])
def test_logpdf_mixed_and_gradient(prec, pol, Nside, scale32=1.0):
    """scale32: factor on the single-precision class bounds (survey patch sizes: the rounding error of the mixing operator and of the
    flows grows with the number of modes; tests/test_gpu_anysize.py)"""
    C, so, sd = _dataset_pair(prec, pol, Nside)
    s32 = scale32
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    oproj = so["proj"]
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    ds.set_data(F(so["d"], C.HARMONIC))
    # a non-trivial G exercises the ϕ re-parametrisation on both sides
    G = np.sqrt(1 + 2 * ods.Nphi * O.pinv(ods.Cphi))
    ods.G = G
    ds.ops["G_inv"] = p.tensor(O.pinv(G)[None])
    import ctypes
    from cmblensing_jl_amd.lib import check
    check(ds.lib.cmbl_dataset_set_op(ds._h, 8, ctypes.c_void_p(ds.ops["G_inv"].data_ptr()), 1))
    fo, po = ods.mix(so["f"], so["phi"])
    gfo_d, gpo_d = ds.mix(F(so["f"], C.HARMONIC), F(so["phi"], C.FOURIER))
    close("mix: f°", gfo_d.arr.cpu().numpy(), fo, 2.7e-6 * s32 if prec == "f32" else 2e-10)                       # measured 8.8e-7
    close("mix: ϕ°", gpo_d.arr.cpu().numpy(), po, 2.1e-7 * s32 if prec == "f32" else 1e-11)                       # 6.9e-8
    lp_o = ods.logpdf_mixed(fo, po)
    lp_g = ds.logpdf_mixed(F(fo, C.MAP), F(po, C.FOURIER))
    scalars_close("logpdf_mixed", lp_g, lp_o, rtol=LPTOL[prec] * (s32 if prec == "f32" else 1))
    for quirk in (False, True):
        lp2, gf, gp = ods.grad_logpdf_mixed(fo, po, alias_quirk=quirk)
        lp3, gf_g, gp_g = ds.gradient_logpdf_mixed(F(fo, C.MAP), F(po, C.FOURIER), alias_quirk=quirk)
        scalars_close("logpdf from the gradient call", lp3, lp2, rtol=LPTOL[prec] * (s32 if prec == "f32" else 1))
        close(("grad f°", quirk), gf_g.arr.cpu().numpy(), gf, 2e-4 * s32 if prec == "f32" else 1e-9)               # measured 9.3e-6 (QU) .. 6.6e-5 (64² T)
        close(("grad ϕ°", quirk), gp_g.arr.cpu().numpy(), gp, 6e-6 * s32 if prec == "f32" else 3e-9)               # 2.0e-6


def test_errors_are_status_codes():
    C = _pkg()
    with pytest.raises(C.CmblError) as e:
        C.ProjLambert(8192, 64, 1.0)                # sides above 4096 do not fit the in-LDS transforms
    assert e.value.code == 2                       # CMBL_ERR_SHAPE
    p = C.ProjLambert(64, 64, 1.0)
    L = C.LenseFlow(p, 7)
    f = C.Field(p, p.tensor(np.zeros((1, 1, 64, 64))), C.MAP)
    with pytest.raises(C.CmblError) as e:
        L * f                                       # set_phi not called
    assert e.value.code == 5                       # CMBL_ERR_STATE


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol", ["I", "P"])
def test_wiener_filter_and_logpdf_closed_form(prec, pol):
    """Oracle-independent known answer for the data model, the operator chain, the reductions and the PCG (src/dataset.jl:59-66,76-80,
    129-132, src/maximization.jl:17-42, src/numerical_algorithms.jl:73-134): without lensing (ϕ = 0: the flow is the identity) and without a
    pixel mask every operator is diagonal in ℓ, so the Wiener filter is f = pinv(Cf⁻¹ + T²Cn⁻¹)·T·Cn⁻¹·d per mode (T = Mf·B), the `:diag`
    preconditioner is the exact inverse (one CG step), and logpdf is a sum over modes -- all formed here in NumPy from the operator
    planes, with nothing from `oracle/`."""
    C = _pkg()
    from bench import synthetic_cls
    tT, nT = DT[prec]
    s = C.load_sim(3.0, (64, 128), pol, synthetic_cls(), T=tT, beam_fwhm=3.0, pixel_mask=None, Nphi="flat")
    ds, p, h = s["ds"], s["proj"], s["ds"].host
    P = ds.P
    pinv = lambda a: np.where(a != 0, 1.0 / np.where(a != 0, a, 1.0), 0.0)
    Cf, Cn, T = np.asarray(h["Cf"].p, float), np.asarray(h["Cn"].p, float), np.asarray(h["Mf"].p, float) * np.asarray(h["B"].p, float)   # (P, Nx, Nyh)
    d = s["d"].arr.cpu().numpy().astype(np.complex128)                                                       # (1, P, Nx, Nyh), harmonic
    A = pinv(Cf) + T ** 2 * pinv(Cn)
    want = pinv(A) * T * pinv(Cn) * d
    phi0 = C.Field(p, torch.zeros_like(s["phi"].arr), C.FOURIER)
    got, hist = ds.argmaxf_logpdf(phi0, tol=1e-12 if prec == "f64" else 1e-4, nsteps=20)
    assert len(hist) <= 3, len(hist)                                          # the preconditioner is the exact inverse here
    close("Wiener filter, closed form", got.arr.cpu().numpy(), want, 3e-6 if prec == "f32" else 1e-12)
    # logpdf(f, ϕ = 0) = -1/2 [ z'Cn⁻¹z + f'Cf⁻¹f + logdets ],  z = T f - d,  <a, b> = Σ λ Re(conj a · b) / (Ny Nx)  (src/proj_lambert.jl:322-325)
    lam = np.asarray(p.lam, float)[None, None, None, :]
    dotF = lambda a, b: float(np.sum(lam * (np.conj(a) * b).real) / (p.Ny * p.Nx))
    f = want
    z = T * f - d
    lp_want = -0.5 * (dotF(z, pinv(Cn) * z) + dotF(f, pinv(Cf) * f) + ds.logdet_sum)
    lp_got = ds.logpdf(C.Field(p, p.tensor(f), C.HARMONIC), phi0)
    scalars_close("logpdf, closed form", lp_got, [lp_want], rtol=1e-6 if prec == "f32" else 1e-12)
    # and the gradient with respect to f vanishes at the Wiener-filtered field (src/dataset.jl:76-80)
    g = ds.gradientf_logpdf(C.Field(p, p.tensor(f), C.HARMONIC), phi0)
    rhs = T * pinv(Cn) * d
    assert rel(g.arr.cpu().numpy() + rhs, rhs) < (1e-5 if prec == "f32" else 1e-11)
