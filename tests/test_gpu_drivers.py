"""GPU parity of the drivers built on the hot path: quadratic_estimate, get_max_lensing_step, one MAP_joint step,
one HMC / Gibbs step -- each against the NumPy oracle on identical inputs and identical random draws."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as O
from test_gpu_parity import _dataset_pair, rel, DT, close, scalars_close


def test_brent_minimizer_cpu_logic():
    import cmblensing_jl_amd as C
    x, fx, n = C.brent_minimize(lambda a: (a - 0.3) ** 2 + 1, 0.0, 2.0, abs_tol=1e-6)
    assert abs(x - 0.3) < 1e-5 and n < 40
    x, _, _ = C.brent_minimize(lambda a: -a, 0.0, 2.0, abs_tol=1e-4)          # monotone: converges to the upper end
    assert x > 2.0 - 1e-3


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol,which", [("I", "TT"), ("P", "EB"), ("P", "EE"), ("IP", "EB")])
def test_quadratic_estimate(prec, pol, which):
    C, so, sd = _dataset_pair(prec, pol, (64, 128), mask=False, beam=1.0)
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    ds.set_data(C.Field(p, p.tensor(so["d"]), C.HARMONIC))
    key = {1: ["T"], 2: ["E", "B"], 3: ["T", "E", "B"]}[ods.P]
    if ods.P == 3:
        planes = lambda op: dict(T=op.te[0], E=op.te[3], B=op.bb)
    else:
        planes = lambda op: {k: op.d[i] for i, k in enumerate(key)}
    TF = {k: planes(ods.Mf)[k] * planes(ods.B)[k] for k in key}
    dd = {k: so["d"][:, i:i + 1] for i, k in enumerate(key)}
    pq, AL, Nphi = O.quadratic_estimate(so["proj"], which, dd, dd, planes(ods.Cf), planes(ods.Cftilde), planes(ods.Cn), ods.Cphi, TF)
    got = C.quadratic_estimate(ds, which)
    m = ods.Cphi > 0
    scalars_close("QE normalisation AL", got["AL"][m], AL[m], rtol=2e-3 if prec == "f32" else 1e-9)
    close("got['phiqe'].arr.cpu().numpy()", got["phiqe"].arr.cpu().numpy(), pq, (2e-3 if prec == "f32" else 1e-9))
    nat = C.quadratic_estimate_native(ds, which)                        # cmbl_quadratic_estimate, directly against the oracle
    scalars_close("cmbl_quadratic_estimate vs oracle: AL", nat["AL"][m], AL[m], rtol=2e-3 if prec == "f32" else 1e-9)
    close("cmbl_quadratic_estimate vs oracle: phiqe", nat["phiqe"].arr.cpu().numpy(), pq, (2e-3 if prec == "f32" else 1e-9))
    # the plane arguments of cmbl_quadratic_estimate may be host or device pointers (include/cmblens.h): the wrapper above passes the
    # dataset's device-resident copies, here the same planes go in as plain host arrays -- identical results
    import ctypes
    pl = next(iter(ds._qe_planes.values()))
    hostp = {k: np.ascontiguousarray(pl[k].cpu().numpy()) for k in ("Cf", "Cft", "Cn", "TF", "Cphi")}
    pd = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    out_h, AL_h = p.empty(C.FOURIER, 1, 1), np.zeros_like(hostp["Cphi"])
    rc = ds.lib.cmbl_quadratic_estimate(ds._h, {"TT": 0, "EE": 1, "EB": 2}[which], pd(hostp["Cf"]), pd(hostp["Cft"]), pd(hostp["Cn"]), pd(hostp["TF"]),
                                        pd(hostp["Cphi"]), 1, None, ctypes.c_void_p(out_h.data_ptr()), pd(AL_h), 1)
    assert rc == 0 and np.array_equal(AL_h, nat["AL"]) and torch.equal(out_h, nat["phiqe"].arr)
    # it is an estimate of ϕ: correlates with the truth
    phi = so["phi"]
    r = O.dot_fourier(so["proj"], pq, phi) / np.sqrt(O.dot_fourier(so["proj"], pq, pq) * O.dot_fourier(so["proj"], phi, phi))
    assert r[0] > 0.5


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_max_lensing_step(prec):
    C, so, sd = _dataset_pair(prec, "I", (64, 64), mask=False)
    p, proj = sd["proj"], so["proj"]
    phi = O.irfft2(so["phi"], proj.Ny)
    eta = O.irfft2(np.sqrt(so["ds"].Cphi) * O.rfft2(O.white_noise(9, (1, 1, 64, 64), np.float64)), proj.Ny)
    want = O.get_max_lensing_step(proj, phi, eta)
    L = C.LenseFlow(p, 7)
    got = L.max_lensing_step(C.Field(p, p.tensor(phi), C.MAP), C.Field(p, p.tensor(eta), C.MAP))
    np.testing.assert_allclose(got[0], want, rtol=1e-3 if prec == "f32" else 1e-9)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol", ["P", "I", "IP"])
def test_map_joint_step(prec, pol):
    C, so, sd = _dataset_pair(prec, pol, (64, 64), mask=False, beam=1.0)
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    ds.set_data(C.Field(p, p.tensor(so["d"]), C.HARMONIC))
    # Nϕ as load_sim builds it (src/dataset.jl:316): quadratic_estimate(ds).Nϕ / Nϕ_fac, on both sides
    Nphi = C.quadratic_estimate(ds)["Nphi"] / 2
    ods.Nphi = Nphi
    ds.host["Nphi"] = Nphi
    phi0 = np.zeros_like(so["phi"])
    # a fixed, short CG (no tolerance stop) keeps the f-step identical on both sides, so the ϕ-step can be compared tightly
    st_o = O.map_joint_step(ods, phi0, alpha_tol=1e-4, cg_tol=0.0, cg_nsteps=10)
    st_g = C.MAP_joint_step(ds, C.Field(p, p.tensor(phi0), C.FOURIER), alpha_tol=1e-4, cg_tol=0.0, cg_nsteps=10)
    tol = 5e-3 if prec == "f32" else 1e-6
    close("st_g['f'].arr.cpu().numpy()", st_g["f"].arr.cpu().numpy(), st_o["f"], (2e-3 if prec == "f32" else 1e-8))
    close("st_g['grad_phi'].arr.cpu().numpy()", st_g["grad_phi"].arr.cpu().numpy(), st_o["grad_phi"], tol)
    close("st_g['dphi'].arr.cpu().numpy()", st_g["dphi"].arr.cpu().numpy(), st_o["dphi"], tol)
    # Brent (ours) vs SciPy's bounded Brent (oracle): same minimiser within the tolerance, same objective value
    assert abs(st_g["alpha"] - st_o["alpha"]) < 5e-3, (st_g["alpha"], st_o["alpha"])
    scalars_close("MAP_joint step logpdf", st_g["logpdf"], st_o["logpdf"], rtol=2e-5)
    assert st_g["logpdf"][0] > st_g["logpdf_before"][0]
    close("st_g['phi'].arr.cpu().numpy()", st_g["phi"].arr.cpu().numpy(), st_o["phi"], 2e-2)
    # the same loop body as ONE library call (cmbl_map_joint_step), directly against the oracle
    st_n = C.MAP_joint_step_native(ds, C.Field(p, p.tensor(phi0), C.FOURIER), alpha_tol=1e-4, cg_tol=0.0, cg_nsteps=10)
    close("cmbl_map_joint_step vs oracle: f", st_n["f"].arr.cpu().numpy(), st_o["f"], (2e-3 if prec == "f32" else 1e-8))
    assert abs(st_n["alpha"] - st_o["alpha"]) < 5e-3, (st_n["alpha"], st_o["alpha"])
    scalars_close("cmbl_map_joint_step vs oracle: logpdf", st_n["logpdf"], st_o["logpdf"], rtol=2e-5)
    close("cmbl_map_joint_step vs oracle: phi", st_n["phi"].arr.cpu().numpy(), st_o["phi"], 2e-2)
    # two more steps keep increasing the posterior and approach the true ϕ
    f, phi, hist = C.MAP_joint(ds, nsteps=3)
    lps = [h["logpdf"][0] for h in hist]
    assert lps[0] < lps[1] < lps[2]
    r = phi.dot(C.Field(p, p.tensor(so["phi"]), C.FOURIER)) / np.sqrt(phi.dot(phi) * O.dot_fourier(so["proj"], so["phi"], so["phi"]))
    assert r[0] > 0.8


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol", ["P", "IP"])
def test_hmc_and_gibbs_step(prec, pol):
    C, so, sd = _dataset_pair(prec, pol, (64, 64), mask=True, beam=1.0)
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    ds.set_data(C.Field(p, p.tensor(so["d"]), C.HARMONIC))
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    B, P, Nx, Ny = 1, ods.P, 64, 64
    wf, wn, wp = (O.white_noise(s, (B, P if s < 9 else 1, Nx, Ny), np.float64) for s in (7, 8, 9))
    logu = np.log(np.random.default_rng(3).random(B))
    # HMC alone from the truth: ΔH, proposal and acceptance agree
    fo, po = ods.mix(so["f"], so["phi"])
    x_o, dH_o, acc_o = O.hmc_step(ods, fo, po, wp, logu, N=5, eps=0.01)
    x_g, dH_g, acc_g = C.hmc_step(ds, F(fo, C.MAP), F(po, C.FOURIER), wp, logu, N=5, eps=0.01)
    # ΔH is a difference of two H ~ 1e5: agreement is limited by the 1e-10 (fp64) / 2e-5 (fp32) relative accuracy of logpdf
    np.testing.assert_allclose(dH_g, dH_o, atol=5.0 if prec == "f32" else 2e-3)
    if prec == "f64":
        assert bool(acc_g[0]) == bool(acc_o[0])
    close("x_g.arr.cpu().numpy()", x_g.arr.cpu().numpy(), x_o, (2e-3 if prec == "f32" else 1e-7))
    # the same update as ONE library call (cmbl_hmc_step), directly against the oracle
    x_n, dH_n, acc_n = C.hmc_step_native(ds, F(fo, C.MAP), F(po, C.FOURIER), white_p=wp, log_u=logu, N=5, eps=0.01)
    np.testing.assert_allclose(dH_n, dH_o, atol=5.0 if prec == "f32" else 2e-3)
    close("cmbl_hmc_step vs oracle: phi°", x_n.arr.cpu().numpy(), x_o, (2e-3 if prec == "f32" else 1e-7))
    if prec == "f64":
        assert bool(acc_n[0]) == bool(acc_o[0])
    # posterior sample of f (src/maximization.jl:56-62): fixed short CG so both sides stop at the same iterate
    f_o, _ = O.sample_f(ods, so["phi"], wf, wn, tol=0.0, nsteps=6)
    f_g, _ = C.sample_f(ds, F(so["phi"], C.FOURIER), wf, wn, tol=0.0, nsteps=6)
    close("f_g.arr.cpu().numpy()", f_g.arr.cpu().numpy(), f_o, (2e-3 if prec == "f32" else 1e-8))
    # unmixed logpdf (gibbs_postprocess!, src/sampling.jl:455-464)
    scalars_close("logpdf", ds.logpdf(F(so["f"], C.HARMONIC), F(so["phi"], C.FOURIER)), ods.logpdf(so["f"], so["phi"]),
                  rtol=2e-5 if prec == "f32" else 1e-10)
    st = C.gibbs_step(ds, F(so["phi"], C.FOURIER), wf, wn, wp, logu, N=3, eps=0.01)
    assert np.all(np.isfinite(st["logpdf"])) and st["f"].arr.shape == (B, P, Nx, Ny // 2 + 1)


def test_sample_joint_chains_are_partition_independent():
    """two chains as two batch slots == the same two chains run one at a time (seed = base + chain id)"""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    kw = dict(T=torch.float64, beam_fwhm=1.0, Nphi="flat")
    both = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), Nbatch=2, **kw)
    # identical data in every slot: chains differ only by their random streams
    d0 = both["d"].arr[:1].repeat(2, 1, 1, 1).contiguous()
    both["ds"].set_data(C.Field(both["proj"], d0, C.HARMONIC))
    r2 = C.sample_joint(both["ds"], 2, chain_ids=(0, 1), base_seed=40, N=3, eps=0.01)
    assert r2["logpdf"].shape == (2, 2) and np.all(np.isfinite(r2["logpdf"]))
    for c in (0, 1):
        one = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), Nbatch=1, **kw)
        one["ds"].set_data(C.Field(one["proj"], d0[:1].contiguous(), C.HARMONIC))
        r1 = C.sample_joint(one["ds"], 2, chain_ids=(c,), base_seed=40, N=3, eps=0.01)
        np.testing.assert_allclose(r1["logpdf"][:, 0], r2["logpdf"][:, c], rtol=1e-6)
        np.testing.assert_allclose(r1["accept"][:, 0], r2["accept"][:, c])
    assert not np.allclose(r2["logpdf"][:, 0], r2["logpdf"][:, 1])


def test_load_sim_uses_quadratic_estimate_noise():
    """load_sim's Nϕ is the QE N⁰ / 2 (src/dataset.jl:316), matching the oracle's estimator on the same data"""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    s = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), T=torch.float64, beam_fwhm=1.0)
    np.testing.assert_allclose(s["ds"].host["Nphi"], C.quadratic_estimate(s["ds"], "EB")["Nphi"] / 2, rtol=1e-12)
    assert s["ds"].host["Nphi"][1, 1] > 0


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_device_randn_matches_oracle(prec):
    """cmbl_randn == the oracle's Philox4x32-10 + Box-Muller, element by element, for every slot / stream; the fp32 draw is the
    rounding of the fp64 draw."""
    import cmblensing_jl_amd as C
    from oracle import rng as R
    T, npT = DT[prec]
    proj = C.ProjLambert(32, 64, 3.0, T, 0)
    seeds, stream, P = [7, 2 ** 40 + 123, 2 ** 64 - 1], 2 ** 33 + 5, 3
    w = proj.randn(seeds, stream, P).cpu().numpy()
    assert w.shape == (3, P, 64, 32) and w.dtype == npT
    n = P * 64 * 32
    for b, s in enumerate(seeds):
        want = R.randn(s, stream, n, np.float64)
        np.testing.assert_allclose(w[b].reshape(-1), want, rtol=2e-7 if prec == "f32" else 1e-13, atol=1e-7 if prec == "f32" else 1e-14)
    assert not np.allclose(w[0], proj.randn(seeds[:1], stream + 1, P).cpu().numpy()[0])
    # ragged tail: a slot size that is not a multiple of 4 still gets every element
    p2 = C.ProjLambert(32, 32, 3.0, T, 0)
    z = p2.randn([5], 0, 1).cpu().numpy().reshape(-1)
    np.testing.assert_allclose(z, R.randn(5, 0, 1024), rtol=2e-7 if prec == "f32" else 1e-13, atol=1e-7)


def test_device_simulate_has_the_requested_spectrum():
    """load_sim(rng="device"): sqrt(C)·rfft(white) with white noise drawn on the GPU -- band powers of f follow Cf"""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    s = C.load_sim(2.0, 256, "P", synthetic_cls(), T=torch.float32, rng="device", Nphi="flat", Nbatch=2)
    proj, ds = s["proj"], s["ds"]
    fh = s["f"].to(C.HARMONIC).arr.cpu().numpy()                         # (B, 2, Nx, Nyh): E, B
    Cee = ds.host["Cf"].p[0] * proj.Nx * proj.Ny                        # <|f_l|^2> = C * Npix (unnormalised rfft)
    lmag = proj.lmag
    for lo, hi in ((200, 600), (600, 1200), (1200, 2500)):
        sel = (lmag > lo) & (lmag < hi) & (Cee > 0)
        ratio = (np.abs(fh[:, 0][:, sel]) ** 2 / Cee[sel]).mean()
        assert abs(ratio - 1) < 6 * np.sqrt(1.0 / (2 * sel.sum())), (lo, hi, ratio)
    assert not np.allclose(fh[0], fh[1])                                 # batch slots are independent draws


def test_sample_joint_device_rng_partition_independent_and_resumable():
    """device-RNG chains: same chain in any batch slot / process gives the same samples; a run resumed at step k continues the
    streams of the uninterrupted run"""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    kw = dict(T=torch.float64, beam_fwhm=1.0, Nphi="flat")
    both = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), Nbatch=2, **kw)
    d0 = both["d"].arr[:1].repeat(2, 1, 1, 1).contiguous()
    both["ds"].set_data(C.Field(both["proj"], d0, C.HARMONIC))
    r2 = C.sample_joint(both["ds"], 3, chain_ids=(0, 1), base_seed=40, N=3, eps=0.01, rng="device")
    one = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), Nbatch=1, **kw)
    one["ds"].set_data(C.Field(one["proj"], d0[:1].contiguous(), C.HARMONIC))
    r1 = C.sample_joint(one["ds"], 3, chain_ids=(1,), base_seed=40, N=3, eps=0.01, rng="device")
    np.testing.assert_allclose(r1["logpdf"][:, 0], r2["logpdf"][:, 1], rtol=1e-6)
    assert not np.allclose(r2["logpdf"][:, 0], r2["logpdf"][:, 1])
    ra = C.sample_joint(one["ds"], 2, chain_ids=(1,), base_seed=40, N=3, eps=0.01, rng="device")
    rb = C.sample_joint(one["ds"], 1, chain_ids=(1,), base_seed=40, N=3, eps=0.01, rng="device", phi_start=ra["phi"], first_step=2)
    np.testing.assert_allclose(rb["logpdf"][0, 0], r1["logpdf"][2, 0], rtol=1e-6)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol", ["P", "IP"])
def test_gradientphi_and_map_marg(prec, pol):
    """∂logpdf/∂ϕ at fixed f and two MAP_marg steps (src/maximization.jl:245-343) against the oracle on identical data and
    identical simulation draws; fixed-length CGs so that both sides stop at the same iterate."""
    C, so, sd = _dataset_pair(prec, pol, (64, 64), mask=True, beam=1.0)
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    ds.set_data(F(so["d"], C.HARMONIC))
    P = ds.P
    g_o = ods.gradientphi_logpdf(so["f"], so["phi"])
    g_g = ds.gradientphi_logpdf(F(so["f"], C.HARMONIC), F(so["phi"], C.FOURIER))
    close("g_g.arr.cpu().numpy()", g_g.arr.cpu().numpy(), g_o, (3e-4 if prec == "f32" else 1e-9))
    # simulated data for a given ϕ
    Nsims = 4
    wf, wn = (O.white_noise(s, (Nsims, P, 64, 64), np.float64) for s in (50, 51))
    d_o = ods.simulate_data(so["phi"], wf[:2], wn[:2])
    d_g = C.simulate_data(ds, F(so["phi"], C.FOURIER), wf[:2], wn[:2])
    close("d_g.arr.cpu().numpy()", d_g.arr.cpu().numpy(), d_o, (5e-5 if prec == "f32" else 1e-10))
    # the iteration.  Hϕ⁻¹ = (Cϕ⁻¹ + Nϕ⁻¹)⁻¹ must not under-estimate the curvature or the fixed-α step diverges: with T data in
    # play the EB-only N⁰ that load_sim uses is far too large, so IP combines the TT and EB estimator noises
    Nphi = C.quadratic_estimate(ds, "EB")["Nphi"] / 2
    if pol == "IP":
        Nphi = O.pinv(O.pinv(Nphi) + O.pinv(C.quadratic_estimate(ds, "TT")["Nphi"] / 2))
    ods.Nphi = Nphi
    ds.host["Nphi"] = Nphi
    # (IP: a single step -- an 8-iteration CG is far from the Wiener filter of the high-S/N T map and a second fixed-α step
    # from that gradient diverges on both sides; P runs the second step at ϕ ≠ 0 with the frozen mean field)
    nst = 2 if pol == "P" else 1
    kw = dict(nsteps=nst, nsteps_with_meanfield_update=1, alpha=0.2, sims_per_batch=2, cg_tol=0.0, cg_nsteps=8)
    phi_o, tr_o = O.map_marg(ods, wf, wn, **kw)
    phi_g, tr_g = C.MAP_marg(ds, Nsims=Nsims, whites={C.rng.STREAM_F: wf, C.rng.STREAM_N: wn}, **kw)
    assert [len(t["ncg"]) for t in tr_g] == [3, 1][:nst]
    close("tr_g[0]['phi'].arr.cpu().numpy()", tr_g[0]["phi"].arr.cpu().numpy(), tr_o[0]["phi"], (2e-3 if prec == "f32" else 1e-7))
    close("phi_g.arr.cpu().numpy()", phi_g.arr.cpu().numpy(), phi_o, (2e-3 if prec == "f32" else 1e-7))
    scalars_close("MAP_marg gradient norms", [t["g_norm"] for t in tr_g], [t["g_norm"] for t in tr_o], rtol=2e-3 if prec == "f32" else 1e-7)


def test_map_marg_device_rng_converges():
    """MAP_marg end to end with sims drawn on the GPU: gradient norm falls, ϕ correlates with the truth, and the result does not
    depend on how sims are grouped into batches beyond the CG stopping rule"""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    s = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), T=torch.float64, beam_fwhm=1.0, pixel_mask=dict(pad_deg=0.3, apod_deg=0.4))
    ds, p = s["ds"], s["proj"]
    phi, tr = C.MAP_marg(ds, nsteps=5, nsteps_with_meanfield_update=3, alpha=0.2, Nsims=8, sims_per_batch=8, base_seed=11)
    gn = [t["g_norm"] for t in tr]
    assert all(b < a for a, b in zip(gn, gn[1:])), gn
    r = phi.dot(s["phi"]) / np.sqrt(phi.dot(phi) * s["phi"].dot(s["phi"]))
    assert r[0] > 0.9, r
    phi1, _ = C.MAP_marg(ds, nsteps=1, nsteps_with_meanfield_update=1, alpha=0.2, Nsims=8, sims_per_batch=1, base_seed=11)
    phi8, _ = C.MAP_marg(ds, nsteps=1, nsteps_with_meanfield_update=1, alpha=0.2, Nsims=8, sims_per_batch=8, base_seed=11)
    close("phi1.arr.cpu().numpy()", phi1.arr.cpu().numpy(), phi8.arr.cpu().numpy(), 0.05)


@pytest.mark.parametrize("ext", [".zip", ".jld2"])
def test_sample_joint_chain_file_and_resume(tmp_path, ext):
    """sample_joint(filename=...) writes chunks every nfilewrite steps; a run interrupted after 4 of 6 steps and resumed from the
    file gives the same chain as the uninterrupted run (device RNG streams are indexed by step), and load_chains reads both.  In
    both containers: the package's zip of .npy and the experimental JLD2 output (src/sampling.jl:311-320; jld2_writer.py)."""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    kw = dict(T=torch.float64, beam_fwhm=1.0, Nphi="flat")
    s = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), Nbatch=2, **kw)
    ds = s["ds"]
    run = dict(chain_ids=(0, 1), base_seed=7, N=3, eps=0.01, rng="device", nfilewrite=2, nsavemaps=3)
    fa, fb = str(tmp_path / ("a" + ext)), str(tmp_path / ("b" + ext))
    ra = C.sample_joint(ds, 6, filename=fa, **run)
    C.sample_joint(ds, 4, filename=fb, **run)
    with pytest.raises(ValueError):
        C.sample_joint(ds, 6, filename=fb, **run)                        # exists: resume must be stated
    rb = C.sample_joint(ds, 6, filename=fb, resume=True, **run)
    assert rb["logpdf"].shape == (2, 2)                                  # only the two remaining steps were run
    np.testing.assert_allclose(rb["logpdf"], ra["logpdf"][4:], rtol=1e-9)
    ca, cb = C.load_chains(fa), C.load_chains(fb)
    # a `.jld2` file carries the reference's step numbers: initial state = step 1, first Gibbs pass = step 2 (src/sampling.jl:263,288-290)
    o = 1 if ext == ".jld2" else 0
    assert len(ca) == 2 and ca["step"].tolist() == [[1 + o, 2 + o, 3 + o, 4 + o, 5 + o, 6 + o]] * 2 == cb["step"].tolist()
    np.testing.assert_allclose(ca["logpdf"], ra["logpdf"].T, rtol=1e-12)
    np.testing.assert_allclose(cb["logpdf"], ca["logpdf"], rtol=1e-9)
    np.testing.assert_allclose(cb["accept"], ca["accept"])
    # maps: first step, every 3rd step, and the last step of every chunk (2, 4, 6)
    assert [("phi" in smp) for smp in ca[0]] == [True, True, True, True, False, True]
    np.testing.assert_allclose(cb[1, -1]["phi"], ca[1, -1]["phi"], rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(ca[1, -1]["phi"], ra["phi"].arr[1, 0].cpu().numpy(), rtol=1e-12)
    assert C.load_chains(fa, thin="hasmaps")["step"].tolist() == [[1 + o, 2 + o, 3 + o, 4 + o, 6 + o]] * 2
    # a new file started from another file's last state (the way a chain the Julia package wrote is continued: read-only source)
    fc = str(tmp_path / ("c" + ext))
    C.sample_joint(ds, 4, filename=str(tmp_path / ("d" + ext)), **run)
    rc = C.sample_joint(ds, 6, filename=fc, resume=str(tmp_path / ("d" + ext)), **run)
    np.testing.assert_allclose(rc["logpdf"], ra["logpdf"][4:], rtol=1e-9)
    assert C.load_chains(fc)["step"].tolist() == [[5 + o, 6 + o]] * 2


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_theta_layer(prec):
    """logpdf(Mixed(ds); f°, ϕ°, θ) for θ = (r, Aϕ) (src/dataset.jl:84-87,272-274,316-328) and the Gibbs θ pass (src/sampling.jl:427-437)
    against the oracle's ParamDependentOp layer; set_theta(ds) restores the fiducial dataset."""
    from oracle.theta import ThetaDataSet, grid_and_sample as o_gas
    C, so, sd = _dataset_pair(prec, "P", (64, 64), mask=True, beam=1.0)
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    ds.set_data(F(so["d"], C.HARMONIC))
    th = ThetaDataSet(ods, so["Cfs"], so["Cten"])
    fo_o, po_o = ods.mix(so["f"], so["phi"])
    fo, po = F(fo_o, C.MAP), F(po_o, C.FOURIER)
    rt = 3e-5 if prec == "f32" else 1e-10
    base = ds.logpdf_mixed(fo, po)
    for kw in (dict(), dict(r=0.2, Aphi=1.0), dict(r=0.35), dict(Aphi=0.8), dict(r=0.1, Aphi=1.25)):
        np.testing.assert_allclose(C.logpdf_mixed_theta(ds, fo, po, **kw), th.logpdf_mixed(fo_o, po_o, **kw), rtol=rt, err_msg=str(kw))
    C.set_theta(ds)
    np.testing.assert_allclose(ds.logpdf_mixed(fo, po), base, rtol=1e-12)
    assert ds.logdet_mix == 0.0
    # Gibbs pass over Aϕ: same conditional on the grid, same draw for the same uniform
    xs = np.linspace(0.6, 1.6, 9)
    val, lp_grid = C.gibbs_sample_theta(ds, fo, po, dict(r=None, Aphi=None), "Aphi", xs, [0.37])
    lps_o = np.array([th.logpdf_mixed(fo_o, po_o, Aphi=a)[0] for a in xs])
    want, _, lp_o = o_gas(lps_o, xs, 0.37)
    np.testing.assert_allclose(lp_grid[0], lp_o, atol=5e-2 if prec == "f32" else 1e-6)
    assert abs(val[0] - want) < (2e-2 if prec == "f32" else 1e-6)
    assert xs[0] < val[0] < xs[-1]
    # the dataset is left at the last grid point: mixing with θ-dependent D, G is consistent with the oracle's
    C.set_theta(ds, r=0.3, Aphi=1.2)
    dso, _, _ = th.at(r=0.3, Aphi=1.2)
    f2_o, p2_o = dso.unmix(fo_o, po_o)
    f2, p2 = ds.unmix(fo, po)
    close("f2.arr.cpu().numpy()", f2.arr.cpu().numpy(), f2_o, (2e-4 if prec == "f32" else 1e-9))
    close("p2.arr.cpu().numpy()", p2.arr.cpu().numpy(), p2_o, (1e-5 if prec == "f32" else 1e-12))
    C.set_theta(ds)


def test_sample_joint_with_theta_pass():
    """sample_joint with a Gibbs pass over Aϕ: the chain runs, θ moves inside its grid and the dataset follows it"""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    s = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), T=torch.float64, beam_fwhm=1.0, Nphi="flat")
    ds = s["ds"]
    xs = np.linspace(0.5, 2.0, 12)
    out = C.sample_joint(ds, 3, chain_ids=(0,), base_seed=5, N=3, eps=0.01, rng="device", theta_ranges=dict(Aphi=xs), phi_start=s["phi"])
    th = [t["Aphi"] for t in out["theta"]]
    assert len(th) == 3 and all(xs[0] <= a <= xs[-1] for a in th) and len(set(th)) > 1
    assert ds.theta["Aphi"] == th[-1] and np.all(np.isfinite(out["logpdf"]))
    C.set_theta(ds)


@pytest.mark.parametrize("ext", [".zip", ".jld2"])
def test_sample_joint_theta_is_saved_and_resumed(tmp_path, ext):
    """θ is part of every saved sample and a resumed θ chain continues the saved one (src/sampling.jl:226-228,247-256):
    4 + 2 resumed steps == 6 uninterrupted steps, and load_chains gives the θ posterior samples."""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    s = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), T=torch.float64, beam_fwhm=1.0, Nphi="flat")
    ds = s["ds"]
    xs = np.linspace(0.5, 2.0, 10)
    run = dict(chain_ids=(0,), base_seed=5, N=3, eps=0.01, rng="device", theta_ranges=dict(Aphi=xs), nfilewrite=2, nsavemaps=2)
    fa, fb = str(tmp_path / ("a" + ext)), str(tmp_path / ("b" + ext))
    ra = C.sample_joint(ds, 6, filename=fa, **run)
    C.set_theta(ds)
    C.sample_joint(ds, 4, filename=fb, **run)
    C.set_theta(ds)                                                    # the resumed run must take θ from the file, not from the dataset
    rb = C.sample_joint(ds, 6, filename=fb, resume=True, **run)
    ca, cb = C.load_chains(fa), C.load_chains(fb)
    tha = [t["Aphi"] for t in ra["theta"]]
    np.testing.assert_allclose(ca["theta_Aphi"][0], tha, rtol=1e-12)
    np.testing.assert_allclose(cb["theta_Aphi"][0], tha, rtol=1e-6)
    np.testing.assert_allclose(cb["logpdf"], ca["logpdf"], rtol=1e-8)
    assert len(rb["theta"]) == 2 and xs[0] <= tha[0] <= xs[-1]
    C.set_theta(ds)


def test_nan_logpdf_is_a_value_not_an_error():
    """logpdf(Mixed) that evaluates to NaN is returned as NaN (src/maximization.jl:194-199 penalises it in the line search,
    src/sampling.jl:414 rejects the proposal); it must not abort the run."""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    s = C.load_sim(3.0, (64, 64), "P", synthetic_cls(), T=torch.float32, beam_fwhm=1.0, Nphi="flat")
    ds, p = s["ds"], s["proj"]
    fo, po = ds.mix(s["f"], s["phi"])
    bad = C.Field(p, po.arr * float("nan"), C.FOURIER)
    lp = ds.logpdf_mixed(fo, bad)
    assert np.isnan(lp[0])
    lp, gf, gp = ds.gradient_logpdf_mixed(fo, bad)
    assert np.isnan(lp[0])
    # a far too generous step bound sends the first line-search evaluations into NaN / garbage territory (|∇∇ϕ| >> 1): the
    # search backs off instead of aborting and the step still improves the posterior
    st = C.MAP_joint_step(ds, C.Field(p, torch.zeros_like(po.arr), C.FOURIER), alpha_max=50.0, cg_tol=0.0, cg_nsteps=5)
    assert np.isfinite(st["alpha"]) and 0 < st["alpha"] < 50
    assert np.isfinite(st["logpdf"][0]) and st["logpdf"][0] > st["logpdf_before"][0]
    # HMC: a NaN ΔH is a rejection
    x, dH, acc = C.hmc_step(ds, fo, bad, np.zeros((1, 1, 64, 64)), np.log([0.5]), N=2, eps=0.01)
    assert np.isnan(dH[0]) and not acc[0]


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_bandpower_theta_and_muse_adapter(prec):
    """(f) rank 4 remainder: bandpower-rescaled covariances as θ (src/proj_lambert.jl:374-411) against the oracle, and the MUSE problem
    interface (ext/CMBLensingMuseInferenceExt.jl:44-72) on the device dataset."""
    from oracle.theta import ThetaDataSet
    C, so, sd = _dataset_pair(prec, "P", (64, 64), mask=True, beam=1.0)
    ods, ds, p = so["ds"], sd["ds"], sd["proj"]
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    ds.set_data(F(so["d"], C.HARMONIC))
    le = [100, 600, 1500, 3000]
    cls = {k: C.Cls(v.ell, v.cl) for k, v in so["cls"]["unlensed_scalar"].items()}
    C.use_bandpowers(ds, {"EE": (le, "AEE")}, cls)
    th = ThetaDataSet(ods, so["Cfs"], so["Cten"], bands={"AEE": ([0], le)})
    fo_o, po_o = ods.mix(so["f"], so["phi"])
    fo, po = F(fo_o, C.MAP), F(po_o, C.FOURIER)
    rt = 3e-5 if prec == "f32" else 1e-10
    for kw in (dict(), dict(AEE=np.ones(3)), dict(AEE=np.array([1.3, 1.0, 0.8])), dict(AEE=np.array([0.9, 1.1, 1.0]), r=0.3, Aphi=1.1)):
        np.testing.assert_allclose(C.logpdf_mixed_theta(ds, fo, po, **kw), th.logpdf_mixed(fo_o, po_o, **kw), rtol=rt, err_msg=str(kw))
    C.set_theta(ds)
    # ---- MUSE adapter
    prob = C.CMBLensingMuseProblem(ds, MAP_joint_kwargs=dict(nsteps=2, cg_nsteps=20))
    z = dict(f=F(so["f"], C.HARMONIC), phi=F(so["phi"], C.FOURIER))
    d = F(so["d"], C.HARMONIC)
    ll = prob.logLike(d, z, dict(Aphi=1.0))
    np.testing.assert_allclose(ll, ods.logpdf(so["f"], so["phi"]), rtol=2e-5 if prec == "f32" else 1e-10)
    if prec == "f64":
        g = prob.grad_theta_logLike(d, z, dict(Aphi=1.0, r=0.2))
        o = lambda **kw: th.at(**kw)[0].logpdf(so["f"], so["phi"])[0]
        for k, x0 in (("Aphi", 1.0), ("r", 0.2)):
            h = 1e-3 * x0
            fd = (o(**{k: x0 + h}) - o(**{k: x0 - h})) / (2 * h)
            np.testing.assert_allclose(g[k], fd, rtol=1e-5)
    x, zs = prob.sample_x_z(7, dict(Aphi=1.2))
    assert tuple(x.arr.shape) == (1, 2, 64, 33) and np.isfinite(x.arr.abs().sum().item())
    x2, _ = prob.sample_x_z(7, dict(Aphi=1.2))
    assert torch.equal(x.arr, x2.arr)                                   # same seed, same simulation
    zh, hist = prob.zhat_at_theta(x, None, dict(Aphi=1.2))
    assert len(hist) == 2 and hist[1]["logpdf"][0] > hist[0]["logpdf"][0]
    assert abs(ds.d.arr - d.arr).max().item() == 0                      # the dataset's own data is restored
    C.set_theta(ds)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol", ["P", "IP"])
def test_native_driver_exports_equal_the_python_drivers(prec, pol):
    """`cmbl_hmc_step` / `cmbl_map_joint_step` (include/cmblens.h; the loop bodies of src/sampling.jl:405-418 and
    src/maximization.jl:160-206 as ONE library call each, for hosts that are neither Julia nor Python) against the Python drivers,
    which the tests above compare with the oracle: same launches in the same order, so agreement is to rounding -- and the library's own
    draws (white_p / log_u = NULL) are the ones `sample_joint(rng="device")` makes."""
    import cmblensing_jl_amd as C
    from cmblensing_jl_amd import rng as R
    from bench import synthetic_cls
    T = torch.float32 if prec == "f32" else torch.float64
    B = 2
    s = C.load_sim(3.0, (64, 128), pol, synthetic_cls(), T=T, beam_fwhm=2.0, pixel_mask=dict(pad_deg=0.4, apod_deg=0.4), Nbatch=B, Nphi="flat")
    ds, proj = s["ds"], s["proj"]
    fo, po = ds.mix(s["f"], s["phi"])
    rng = np.random.default_rng(5)
    wp = rng.standard_normal((B, 1, proj.Nx, proj.Ny))
    logu = np.log(rng.random(B))
    tight = 2e-5 if prec == "f32" else 1e-10
    # injected draws
    x_p, dH_p, acc_p = C.hmc_step(ds, fo, po, wp, logu, N=4, eps=0.02)
    x_n, dH_n, acc_n = C.hmc_step_native(ds, fo, po, white_p=wp, log_u=logu, N=4, eps=0.02)
    close("native hmc_step: phi°", x_n.arr.cpu().numpy(), x_p.arr.cpu().numpy(), tight)
    np.testing.assert_allclose(dH_n, dH_p, atol=(0.5 if prec == "f32" else 1e-6))       # ΔH: a difference of logpdfs ~ 1e6..1e7
    assert acc_n.tolist() == acc_p.tolist()
    # always_accept takes the proposal, a rejecting uniform keeps the start
    x_a, _, acc_a = C.hmc_step_native(ds, fo, po, white_p=wp, log_u=np.full(B, 1e30), N=2, eps=0.02, always_accept=True)
    x_r, _, acc_r = C.hmc_step_native(ds, fo, po, white_p=wp, log_u=np.full(B, 1e30), N=2, eps=0.02)
    assert acc_a.all() and not acc_r.any() and torch.equal(x_r.arr, po.arr) and not torch.equal(x_a.arr, po.arr)
    # the library's own draws = the device-RNG draws of sample_joint (rng.py stream ids)
    seeds, step = [11, 12], 3
    wdev = proj.randn(seeds, R.stream_id(R.STREAM_P, step), 1)
    ludev = np.log(np.array([R.uniform(sd, R.stream_id(R.STREAM_U, step))[0] for sd in seeds]))
    x_p2, dH_p2, acc_p2 = C.hmc_step(ds, fo, po, wdev, ludev, N=3, eps=0.02)
    x_n2, dH_n2, acc_n2 = C.hmc_step_native(ds, fo, po, seeds=seeds, step=step, N=3, eps=0.02)
    close("native hmc_step, library draws: phi°", x_n2.arr.cpu().numpy(), x_p2.arr.cpu().numpy(), tight)
    assert acc_n2.tolist() == acc_p2.tolist()
    # MAP_joint step
    p0 = C.Field(proj, torch.zeros_like(po.arr), C.FOURIER)
    # a G that is NOT the identity the step runs with (load_sim's default mixing matrix is 1): the restore below then proves something
    ds.set_op("G_inv", (0.5 + np.random.default_rng(9).random(np.asarray(ds.host["Cphi"]).shape))[None])
    st_p = C.MAP_joint_step(ds, p0, alpha_tol=1e-4, cg_tol=0.0, cg_nsteps=8)
    Ginv_before = ds.ops["G_inv"].clone()
    assert float((Ginv_before - 1).abs().max()) > 1e-3
    lp_G_before = np.asarray(ds.logpdf_mixed(fo, po)).copy()                # evaluated through the LIBRARY's G slot (a mixed-variable logpdf)
    st_n = C.MAP_joint_step_native(ds, p0, alpha_tol=1e-4, cg_tol=0.0, cg_nsteps=8)
    assert st_n["ncg"] == len(st_p["cg_hist"]) == 8
    close("native MAP_joint step: f", st_n["f"].arr.cpu().numpy(), st_p["f"].arr.cpu().numpy(), tight)
    assert abs(st_n["alpha"] - st_p["alpha"]) < (2e-3 if prec == "f32" else 1e-6) * max(1.0, st_p["alpha"]), (st_n["alpha"], st_p["alpha"])
    close("native MAP_joint step: phi", st_n["phi"].arr.cpu().numpy(), st_p["phi"].arr.cpu().numpy(), 5e-3 if prec == "f32" else 1e-6)
    scalars_close("native MAP_joint step: logpdf", st_n["logpdf"], st_p["logpdf"], rtol=2e-6 if prec == "f32" else 1e-10)
    assert st_n["linesearch_evals"] == st_p["linesearch_evals"] or prec == "f32"
    assert torch.equal(ds.ops["G_inv"], Ginv_before)                        # the host-side copy was never touched ...
    assert np.array_equal(np.asarray(ds.logpdf_mixed(fo, po)), lp_G_before)    # ... and the library's own G slot is back: same logpdf, bit for bit
    lp_chk = ds.logpdf_mixed(*ds.mix(st_n["f"], st_n["phi"]))               # and the library still evaluates with it
    assert np.all(np.isfinite(lp_chk))


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol,which", [("I", "TT"), ("P", "EE"), ("P", "EB"), ("IP", "TT"), ("IP", "EB")])
def test_native_quadratic_estimate_equals_the_python_driver(prec, pol, which):
    """`cmbl_quadratic_estimate` (the sums of leg products of src/quadratic_estimate.jl:95-175 inside the library) against the Python
    driver, which test_quadratic_estimate compares with the oracle: same legs, same order -> rounding-level agreement, batch of 2."""
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    T = torch.float32 if prec == "f32" else torch.float64
    s = C.load_sim(3.0, (64, 128), pol, synthetic_cls(), T=T, beam_fwhm=2.0, pixel_mask=dict(pad_deg=0.4, apod_deg=0.4), Nbatch=2, Nphi="flat")
    ds = s["ds"]
    a, b = C.quadratic_estimate(ds, which), C.quadratic_estimate_native(ds, which)
    m = a["AL"] > 0
    scalars_close(f"native QE {which}: AL", b["AL"][m], a["AL"][m], rtol=2e-5 if prec == "f32" else 1e-11)
    assert np.array_equal(b["AL"] > 0, m)
    close(f"native QE {which}: phiqe", b["phiqe"].arr.cpu().numpy(), a["phiqe"].arr.cpu().numpy(), 2e-5 if prec == "f32" else 1e-11)
    # a given normalisation is used as is; the unfiltered estimate differs from the filtered one
    c2 = C.quadratic_estimate_native(ds, which, wiener_filtered=False, AL=a["AL"])
    a2 = C.quadratic_estimate(ds, which, wiener_filtered=False, AL=a["AL"])
    close(f"native QE {which}: unfiltered, AL given", c2["phiqe"].arr.cpu().numpy(), a2["phiqe"].arr.cpu().numpy(), 2e-5 if prec == "f32" else 1e-11)
