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
