"""Oracle sanity for the driver rows: the restated quadratic estimator recovers ϕ, a MAP_joint
step raises the posterior, HMC conserves H for a small step."""
import numpy as np

import oracle as O


def _planes(ods):
    key = {1: ["T"], 2: ["E", "B"]}[ods.P]
    pl = lambda op: {k: op.d[i] for i, k in enumerate(key)}
    return key, pl


def test_quadratic_estimate_recovers_phi():
    for pol, which in (("I", "TT"), ("P", "EB"), ("P", "EE")):
        s = O.load_sim(3.0, (64, 64), pol, np.float64, beam_fwhm=1.0)
        ds, proj = s["ds"], s["proj"]
        key, pl = _planes(ds)
        TF = {k: pl(ds.Mf)[k] * pl(ds.B)[k] for k in key}
        dd = {k: s["d"][:, i:i + 1] for i, k in enumerate(key)}
        pq, AL, Nphi = O.quadratic_estimate(proj, which, dd, dd, pl(ds.Cf), pl(ds.Cftilde), pl(ds.Cn), ds.Cphi, TF)
        assert np.all(np.isfinite(AL)) and np.all(AL >= 0) and AL[1, 1] > 0
        r = O.dot_fourier(proj, pq, s["phi"]) / np.sqrt(O.dot_fourier(proj, pq, pq) * O.dot_fourier(proj, s["phi"], s["phi"]))
        assert r[0] > 0.8, (which, r)
        # un-Wiener-filtered estimate with the normalisation supplied is consistent (AL reused, :40-46)
        pq2, AL2, _ = O.quadratic_estimate(proj, which, dd, dd, pl(ds.Cf), pl(ds.Cftilde), pl(ds.Cn), ds.Cphi, TF, wiener_filtered=False, AL=AL)
        np.testing.assert_allclose(pq, (ds.Cphi * O.pinv(ds.Cphi + AL)) * pq2, rtol=1e-10, atol=1e-30)


def test_map_joint_and_hmc():
    s = O.load_sim(3.0, (64, 64), "P", np.float64, beam_fwhm=1.0)
    ds, proj = s["ds"], s["proj"]
    key, pl = _planes(ds)
    TF = {k: pl(ds.Mf)[k] * pl(ds.B)[k] for k in key}
    dd = {k: s["d"][:, i:i + 1] for i, k in enumerate(key)}
    ds.Nphi = O.quadratic_estimate(proj, "EB", dd, dd, pl(ds.Cf), pl(ds.Cftilde), pl(ds.Cn), ds.Cphi, TF)[2] / 2   # dataset.jl:316
    f, phi, hist = O.map_joint(ds, nsteps=3)
    lps = [h["logpdf"][0] for h in hist]
    assert lps[0] < lps[1] < lps[2] and all(0 < h["alpha"] for h in hist)
    r = O.dot_fourier(proj, phi, s["phi"]) / np.sqrt(O.dot_fourier(proj, phi, phi) * O.dot_fourier(proj, s["phi"], s["phi"]))
    assert r[0] > 0.9
    # leapfrog: |ΔH| shrinks ~ eps² (symplectic, 2nd order) and small steps are accepted
    fo, po = ds.mix(f, phi)
    w = O.white_noise(9, (1, 1, 64, 64), np.float64)
    dH = [abs(O.hmc_step(ds, fo, po, w, np.array([-1e9]), N=4, eps=e)[1][0]) for e in (0.02, 0.01)]
    assert dH[1] < dH[0] and dH[1] < 0.1
    x, dH1, acc = O.hmc_step(ds, fo, po, w, np.array([np.log(0.5)]), N=4, eps=0.01)
    assert acc[0] and not np.allclose(x, po)
    x, _, acc = O.hmc_step(ds, fo, po, w, np.array([1e9]), N=2, eps=0.01)           # never accepted -> state unchanged
    assert (not acc[0]) and np.array_equal(x, po)


def test_gradientphi_logpdf_finite_difference_and_map_marg():
    """∂logpdf/∂ϕ at fixed f agrees with a finite difference of logpdf (the property the reference checks for its flow
    gradients, test/runtests.jl:566-579); MAP_marg's mean field removes the mask-induced bias and the iteration moves ϕ towards
    the truth."""
    s = O.load_sim(3.0, (32, 32), "P", np.float64, beam_fwhm=1.0, pixel_mask=dict(pad_deg=0.2, apod_deg=0.3), nsteps=14)
    ds, proj = s["ds"], s["proj"]
    f, phi = s["f"], s["phi"]
    g = ds.gradientphi_logpdf(f, phi)
    eta = O.rfft2(O.irfft2(s["phi"], proj.Ny)[..., ::-1, :].copy()) * 0.3            # some other ϕ-like direction
    eps = 1e-4
    fd = (ds.logpdf(f, phi + eps * eta) - ds.logpdf(f, phi - eps * eta)) / (2 * eps)
    an = O.dot_fourier(proj, g, eta) * (proj.Nx * proj.Ny)                           # Fourier cotangent convention: Σλ g*η  (autodiff.jl:31-34)
    an2 = O.dot_fourier(proj, g, eta)
    assert min(abs(fd[0] - an[0]), abs(fd[0] - an2[0])) < 2e-2 * abs(fd[0]), (fd, an, an2)

    # MAP_marg on the 64x64 masked simulation with the reference's Nϕ = N⁰(QE)/2 (dataset.jl:316)
    s = O.load_sim(3.0, (64, 64), "P", np.float64, beam_fwhm=1.0, pixel_mask=dict(pad_deg=0.3, apod_deg=0.4))
    ds, proj = s["ds"], s["proj"]
    key, pl = _planes(ds)
    TF = {k: pl(ds.Mf)[k] * pl(ds.B)[k] for k in key}
    dd = {k: s["d"][:, i:i + 1] for i, k in enumerate(key)}
    ds.Nphi = O.quadratic_estimate(proj, "EB", dd, dd, pl(ds.Cf), pl(ds.Cftilde), pl(ds.Cn), ds.Cphi, TF)[2] / 2
    Nsims = 4
    wf = O.white_noise(50, (Nsims, 2, 64, 64), np.float64)
    wn = O.white_noise(51, (Nsims, 2, 64, 64), np.float64)
    phi1, tr = O.map_marg(ds, wf, wn, nsteps=3, nsteps_with_meanfield_update=2, alpha=0.2, sims_per_batch=2)
    assert len(tr) == 3 and len(tr[0]["ncg"]) == 3 and len(tr[2]["ncg"]) == 1           # data + 2 sim batches, then data only
    assert tr[2]["g_norm"] < tr[1]["g_norm"] < tr[0]["g_norm"]
    r = O.dot_fourier(proj, phi1, s["phi"]) / np.sqrt(O.dot_fourier(proj, phi1, phi1) * O.dot_fourier(proj, s["phi"], s["phi"]))
    assert r[0] > 0.9, r
    # one sim per CG (the reference's Nbatch = 1) gives nearly the same step as the batched CG
    phi2, _ = O.map_marg(ds, wf, wn, nsteps=1, nsteps_with_meanfield_update=1, alpha=0.2, sims_per_batch=1)
    phi3, _ = O.map_marg(ds, wf, wn, nsteps=1, nsteps_with_meanfield_update=1, alpha=0.2, sims_per_batch=4)
    assert np.linalg.norm(phi2 - phi3) < 0.05 * np.linalg.norm(phi2)
