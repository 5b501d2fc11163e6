"""The five BASELINE.json configurations as GPU tests, plus the reference's reduction known answers on the device.

  config 1  256² T, one LenseFlow forward pass                       -> vs the oracle (the reference's own CPU-runnable case)
  config 2  512² QU fp32: L*f, L'g, one Wiener-filter CG solve        -> vs the oracle (flows tightly, CG count within 5 %)
  config 3  1024² T+QU fp32: one MAP_joint gradient step (inner CG)   -> properties (logpdf rises, FD of the f°-gradient)
  config 4  1024² T+QU sample_joint chains                            -> one Gibbs step through properties; the chain partition
                                                                        itself is covered by tests/test_chains_gloo.py
  config 5  2048² QU fp64, n = 10 + quadratic_estimate(:EB)           -> properties at full size, QE vs the oracle at 256²
Sizes the oracle cannot do in seconds are checked through size-independent properties, as the reference's own tests do
(test/runtests.jl:556-621).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as O
from oracle.lenseflow import LenseFlow as OLenseFlow
from test_gpu_parity import _dataset_pair, rel, sims, close, scalars_close
from bench import synthetic_cls


def _pkg():
    import cmblensing_jl_amd as C
    return C


# ---------------------------------------------------------------------------------------------------------------------
# reductions: test/runtests.jl:249-285 on the device, in every accumulation mode (src/util.jl:288-316)
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("mode", ["float64", "working", "kahan"])
def test_logdet_tr_known_answers(prec, mode):
    C = _pkg()
    T = torch.float32 if prec == "f32" else torch.float64
    p = C.ProjLambert(32, 32, 1.0, T)
    p.set_sum_accuracy_mode(mode)
    # the reference's 2x2 known answers, embedded in the smallest map the engine takes: ones elsewhere (log 1 = 0; even number
    # of negatives) for logdet, zeros elsewhere for tr
    x = np.array([[1, -2], [3, -4]], float)
    for P in (1, 2, 3):
        m = np.ones((2, P, 32, 32)); m[:, :, :2, :2] = x              # batched (runtests.jl:255)
        np.testing.assert_allclose(p.logdet_diag(p.tensor(m), C.MAP), [P * np.log(24)] * 2, rtol=1e-6)
        z = np.zeros((2, P, 32, 32)); z[:, :, :2, :2] = x
        np.testing.assert_allclose(p.tr_diag(p.tensor(z), C.MAP), [-2 * P] * 2, rtol=1e-6)
    odd = np.ones((1, 1, 32, 32)); odd[0, 0, 0, 0] = -1.0              # log(prod(sign)) = log(-1): Julia throws, here NaN
    assert np.isnan(p.logdet_diag(p.tensor(odd), C.MAP)[0])
    odd[0, 0, 0, 0] = 0.0
    assert p.logdet_diag(p.tensor(odd), C.MAP)[0] == -np.inf
    # Fourier: against the dense fft (runtests.jl:258-265, 275-282) at the reference's sizes
    rt = 2e-5 if (prec == "f32" and mode == "working") else (2e-6 if prec == "f32" else 1e-11)
    for Ny, Nx in ((128, 128), (64, 128), (128, 64)):
        q = C.ProjLambert(Ny, Nx, 1.0, T)
        q.set_sum_accuracy_mode(mode)
        xm = np.random.default_rng(4).random((Nx, Ny))
        full = np.fft.fft2(xm)
        ld, tr = np.sum(np.log(np.abs(full))), np.sum(full).real
        for P in (1, 2, 3):
            xs = q.rfft(q.tensor(np.broadcast_to(xm, (2, P, Nx, Ny)).copy()))
            np.testing.assert_allclose(q.logdet_diag(xs, C.FOURIER), [P * ld] * 2, rtol=rt * 10)
            np.testing.assert_allclose(q.tr_diag(xs, C.FOURIER), [P * tr] * 2, rtol=rt * 10)
        # norm and the Map / Fourier dots (Parseval)
        a = q.tensor(np.random.default_rng(5).standard_normal((1, 2, Nx, Ny)))
        n_map, n_f = q.norm(a, C.MAP), q.norm(q.rfft(a), C.FOURIER)
        np.testing.assert_allclose(n_map, np.sqrt(q.dot(a, a, C.MAP)), rtol=1e-12)
        np.testing.assert_allclose(n_map, n_f, rtol=rt * 10)
        np.testing.assert_allclose(n_map, np.linalg.norm(a.double().cpu().numpy()), rtol=rt * 10)


def test_sum_accuracy_modes_differ_as_documented():
    """fp32: plain working-precision accumulation loses digits that the Float64 and Kahan modes keep (src/util.jl:288-316)"""
    C = _pkg()
    p = C.ProjLambert(512, 512, 1.0, torch.float32)
    a = (1.0 + 1e-3 * np.random.default_rng(0).standard_normal((1, 1, 512, 512))).astype(np.float32)
    exact = np.sum(a.astype(np.float64) ** 2)                          # terms are formed in fp32 by all modes: compare on fp32 products
    exact32 = np.sum((a * a).astype(np.float64))
    t = p.tensor(a)
    err = {}
    for mode in ("working", "float64", "kahan"):
        p.set_sum_accuracy_mode(mode)
        err[mode] = abs(p.dot(t, t, C.MAP)[0] - exact32) / exact32
    assert err["float64"] < 1e-9 and err["kahan"] < 2e-7 and err["working"] < 1e-4
    assert err["working"] >= err["float64"]
    assert abs(exact - exact32) / exact < 1e-6
    with pytest.raises(KeyError):
        p.set_sum_accuracy_mode("nonsense")


# ---------------------------------------------------------------------------------------------------------------------
def test_config1_256_T_lenseflow_forward(camb):
    """BASELINE config 1: 256² T-only, one LenseFlow forward pass -- the oracle's CPU run is the reference's own plumbing case"""
    C = _pkg()
    oproj, simf, simp = sims(camb, 256, 256, 1, 1)
    for prec, T, nT, tol in (("f32", torch.float32, np.float32, 5e-5), ("f64", torch.float64, np.float64, 1e-10)):
        f = simf(1).astype(nT).astype(np.float64)
        phi = simp(2, 1).astype(nT).astype(np.float64)
        want = OLenseFlow(oproj, phi, 7).apply(f)
        p = C.ProjLambert(256, 256, 2.0, T)
        got = C.LenseFlow(p, 7)(C.Field(p, p.tensor(phi), C.MAP)) * C.Field(p, p.tensor(f), C.MAP)
        close("got.arr.cpu().numpy()", got.arr.cpu().numpy(), want, tol)


def test_config2_512_QU_fwd_adjoint_wiener():
    """BASELINE config 2: 512² QU fp32 -- L*f and L'g against the oracle run here; the argmaxf_logpdf solve (tol 1e-1, <= 500 its, true
    ϕ) against the float64 oracle's solve of the same problem, which takes the oracle minutes and is committed as data
    (tests/golden/config2_cg.json, tools/make_config2_golden.py): iteration count within 5 %, same residual history, same solution norm."""
    import json, os
    C, so, sd = _dataset_pair("f32", "P", (512, 512), theta=2.0, mask=True, beam=0.0)
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    f, phi, d = so["f"], so["phi"], so["d"]
    ds.set_data(F(d, C.HARMONIC))
    OL = ods.L(phi)
    L = ds.L(F(phi, C.FOURIER))
    fm = O.from_harm(so["proj"], f)
    close("L*f", (L * F(fm, C.MAP)).arr.cpu().numpy(), OL.apply(fm), 5e-5)
    gl = O.rfft2(fm[:, ::-1].copy())
    close("L'g", (L.adjoint * F(gl, C.FOURIER)).arr.cpu().numpy(), OL.adj(gl), 5e-5)
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config2_cg.json")))
    fw_g, h_g = ds.argmaxf_logpdf(F(phi, C.FOURIER), tol=1e-1, nsteps=500)
    assert abs(len(h_g) - gold["ncg"]) <= max(2, gold["ncg"] // 20), (len(h_g), gold["ncg"])          # CG count within 5 %
    res_g = np.array([h[1][0] for h in h_g])
    n = min(len(res_g), gold["ncg"])
    np.testing.assert_allclose(res_g[0], gold["res"][0], rtol=1e-3)
    np.testing.assert_allclose(res_g[:10], gold["res"][:10], rtol=1e-3)                                 # fp32 round-off is amplified from there on
    assert res_g[n - 1] < 1e-3 * res_g[0] and gold["res"][n - 1] < 1e-3 * gold["res"][0]
    np.testing.assert_allclose(float(fw_g.arr.abs().pow(2).sum().sqrt()), gold["f_l2"], rtol=1e-2)


def _fd_check_f_gradient(C, ds, fo, po, lp0, gfo, seed=3, e=0.05):
    """directional derivative of logpdf(Mixed) along a random f° direction vs central differences of the device logpdf"""
    p = ds.proj
    u = C.Field(p, p.tensor(np.random.default_rng(seed).standard_normal(tuple(fo.arr.shape))), C.MAP)
    # scale the direction so that the logpdf changes by O(1): ‖u‖ ~ rms(f°)·1e-3
    s = 1e-3 * float(fo.arr.std())
    lp = lambda a: float(ds.logpdf_mixed(p.axpby(1.0, fo, a * s, u), po)[0])
    fd = (lp(-2 * e) - 8 * lp(-e) + 8 * lp(e) - lp(2 * e)) / (12 * e)
    an = s * float(gfo.dot(u)[0])
    return fd, an


@pytest.mark.parametrize("pol", ["IP"])
def test_config3_1024_IQU_map_joint_step(pol):
    """BASELINE config 3: 1024² T+QU fp32, one full MAP_joint gradient step (inner Wiener CG + ∇logpdf(Mixed) + line search).
    Properties: the CG residual falls by orders of magnitude, the step increases the posterior, α is inside its bracket, the
    f°-gradient agrees with finite differences of the device logpdf (test/runtests.jl:585-621, src/maximization.jl:160-226)."""
    C = _pkg()
    s = C.load_sim(2.0, 1024, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
    ds, p = s["ds"], s["proj"]
    assert ds.P == 3
    phi0 = C.Field(p, torch.zeros_like(s["phi"].arr), C.FOURIER)
    st = C.MAP_joint_step(ds, phi0, cg_tol=1e-1, cg_nsteps=60)
    res = [float(h[1][0]) for h in st["cg_hist"]]
    assert len(res) >= 2 and min(res) < 1e-2 * res[0]
    assert np.isfinite(st["logpdf"][0]) and st["logpdf"][0] > st["logpdf_before"][0]
    assert 0 < st["alpha"] <= 2.0
    # the ϕ step goes towards the truth
    r = st["phi"].dot(s["phi"]) / np.sqrt(st["phi"].dot(st["phi"]) * s["phi"].dot(s["phi"]))
    assert r[0] > 0.2
    # f°-gradient vs finite differences at the stepped point
    lp, gfo, gpo = ds.gradient_logpdf_mixed(st["f_mixed"], st["phi_mixed"])
    fd, an = _fd_check_f_gradient(C, ds, st["f_mixed"], st["phi_mixed"], lp, gfo)
    assert abs(fd - an) < 0.05 * abs(fd) + 3.0, (fd, an)               # fp32 logpdf ~1e6: differences resolve ~O(1) (runtests.jl atol 3/30)


def test_config4_1024_IQU_gibbs_step():
    """BASELINE config 4 (per-GPU share): one sample_joint Gibbs pass of a 1024² T+QU chain -- f | ϕ (Wiener CG), HMC over ϕ°,
    unmix, logpdf -- all finite, |ΔH| moderate at ϵ = 0.005 from the true ϕ, state shapes right."""
    C = _pkg()
    s = C.load_sim(2.0, 1024, "IP", synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), Nphi="flat")
    ds, p = s["ds"], s["proj"]
    p.set_sum_accuracy_mode("float64")
    out = C.sample_joint(ds, 1, chain_ids=(0,), base_seed=3, N=5, eps=0.005, rng="device", phi_start=s["phi"], nburnin_always_accept=0)
    assert out["logpdf"].shape == (1, 1) and np.isfinite(out["logpdf"]).all()
    assert np.isfinite(out["dH"]).all() and abs(out["dH"][0, 0]) < 200
    assert tuple(out["f"].arr.shape) == (1, 3, 1024, 513) and tuple(out["phi"].arr.shape) == (1, 1, 1024, 513)


def test_config5_2048_QU_f64_n10_and_qe():
    """BASELINE config 5: 2048² QU fp64, LenseFlow with N = 10 RK steps + quadratic_estimate(:EB) at full size (properties), and the
    same estimator against the oracle at 256²."""
    C = _pkg()
    from test_gpu_fullsize import _fields
    proj = C.ProjLambert(2048, 2048, 2.0, torch.float64)
    f, g, phi, dphi = _fields(C, proj, 2)
    L = C.LenseFlow(proj, 10)(phi)
    Lg = L * g
    np.testing.assert_allclose(f.dot(Lg), (L.adjoint * f.to(C.FOURIER)).dot(g.to(C.FOURIER)), rtol=1e-9)
    back = L.ldiv(Lg)
    assert float((back.arr - g.arr).norm() / g.arr.norm()) < 3e-4       # RK4 n = 10 discretisation (1e-3 at n = 7)
    # n = 10 is closer to n = 20 than n = 7 is: the step count is live at this size
    ref = (C.LenseFlow(proj, 20)(phi) * g).arr
    e10 = float((Lg.arr - ref).norm() / ref.norm())
    e7 = float(((C.LenseFlow(proj, 7)(phi) * g).arr - ref).norm() / ref.norm())
    assert e10 < 0.5 * e7
    del L, Lg, back, ref, f, g, phi, dphi
    torch.cuda.empty_cache()
    # quadratic_estimate(:EB) at 2048²: N⁰ is real, non-negative, isotropic (same along the two axes) and the estimate correlates with ϕ
    s = C.load_sim(2.0, 2048, "P", synthetic_cls(), T=torch.float64, nsteps=10, Nphi="flat")
    qe = C.quadratic_estimate(s["ds"], "EB")
    N0 = qe["Nphi"]
    assert np.all(np.isfinite(N0)) and np.all(N0 >= 0) and np.any(N0 > 0)
    lx, ly = s["proj"].lx, s["proj"].ly
    kx = np.argmin(np.abs(lx - 500.0)); ky = np.argmin(np.abs(ly - lx[kx]))
    np.testing.assert_allclose(N0[kx, 0], N0[0, ky], rtol=1e-6)           # N⁰(ℓ along x) == N⁰(ℓ along y)
    pq = qe["phiqe"]
    r = pq.dot(s["phi"]) / np.sqrt(pq.dot(pq) * s["phi"].dot(s["phi"]))
    assert r[0] > 0.5
    del s, qe
    torch.cuda.empty_cache()
    # against the oracle at 256² (fp64, n irrelevant for the estimator)
    C2, so, sd = _dataset_pair("f64", "P", (256, 256), theta=2.0, mask=False, beam=1.0)
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    ds.set_data(C2.Field(p, p.tensor(so["d"]), C2.HARMONIC))
    planes = lambda op: {k: op.d[i] for i, k in enumerate(["E", "B"])}
    TF = {k: planes(ods.Mf)[k] * planes(ods.B)[k] for k in "EB"}
    dd = {k: so["d"][:, i:i + 1] for i, k in enumerate("EB")}
    pq_o, AL, _ = O.quadratic_estimate(so["proj"], "EB", dd, dd, planes(ods.Cf), planes(ods.Cftilde), planes(ods.Cn), ods.Cphi, TF)
    got = C2.quadratic_estimate(ds, "EB")
    # inside the band the estimator uses (LowPass(3000)); beyond it the normalisation is 1/(round-off) on both sides
    m = (ods.Cphi > 0) & (so["proj"].lmag < 2500)
    np.testing.assert_allclose(got["AL"][m], AL[m], rtol=1e-7)
    close("got['phiqe'].arr.cpu().numpy()[...", got["phiqe"].arr.cpu().numpy()[..., m], pq_o[..., m], 1e-7)


def test_more_than_64_batch_slots():
    """80 chains as batch slots in one call (the per-slot scalar buffers of reductions, CG and posterior hold MAXBATCH = 256): slots 0, 1
    equal the same two slots of a 2-chain dataset (the generator fills slot-major, so their draws coincide); 257 slots are refused."""
    import torch
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    kw = dict(T=torch.float64, beam_fwhm=2.0, pixel_mask=dict(pad_deg=0.2, apod_deg=0.2), Nphi="flat")
    big = C.load_sim(3.0, (32, 64), "P", synthetic_cls(), Nbatch=80, **kw)
    two = C.load_sim(3.0, (32, 64), "P", synthetic_cls(), Nbatch=2, **kw)
    out = []
    for s in (big, two):
        ds = s["ds"]
        fo, po = ds.mix(s["f"], s["phi"])
        lp, gf, gp = ds.gradient_logpdf_mixed(fo, po)
        fw, hist = ds.argmaxf_logpdf(s["phi"], tol=0.0, nsteps=6)
        out.append((np.asarray(lp), gf.arr.cpu().numpy(), gp.arr.cpu().numpy(), fw.arr.cpu().numpy(), np.array([h[1] for h in hist]), ds.proj.dot(fo.arr, fo.arr, C.MAP)))
    b, t = out
    assert len(b[0]) == 80 and np.all(np.isfinite(b[0]))
    np.testing.assert_allclose(b[0][:2], t[0], rtol=1e-11)
    for k in (1, 2, 3):
        np.testing.assert_allclose(b[k][:2], t[k], rtol=0, atol=1e-10 * np.abs(t[k]).max())
    np.testing.assert_allclose(b[4][:, :2], t[4], rtol=1e-9)
    np.testing.assert_allclose(b[5][:2], t[5], rtol=1e-12)
    p = big["proj"]
    with pytest.raises(C.CmblError):
        p.dot(torch.zeros(257, 1, 64, 32, dtype=torch.float64, device="cuda"), torch.zeros(257, 1, 64, 32, dtype=torch.float64, device="cuda"), C.MAP)
