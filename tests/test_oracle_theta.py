"""θ layer of the oracle: grid_and_sample against analytic densities (the reference's own docstring example,
src/sampling.jl:58-72), ParamDependentOp bookkeeping of logpdf(Mixed; θ) (src/dataset.jl:84-87, src/generic.jl:269)."""
import numpy as np

import oracle as O
from oracle.theta import ThetaDataSet, grid_and_sample, loess


def test_grid_and_sample_gaussian():
    xs = np.linspace(-3, 3, 100)
    lp = -(xs - 0.4) ** 2 / (2 * 0.5 ** 2) + 17.0                       # arbitrary offset: only differences matter
    us = (np.arange(2000) + 0.5) / 2000
    smp = np.array([grid_and_sample(lp, xs, u)[0] for u in us[::20]])
    assert abs(smp.mean() - 0.4) < 0.02 and abs(smp.std() - 0.5) < 0.03
    s, (xf, sm), at_xs = grid_and_sample(lp, xs, 0.5)
    assert abs(s - 0.4) < 1e-3                                            # median of a symmetric density
    assert abs(np.trapezoid(np.exp(sm), xf) - 1) < 1e-6                       # normalised
    np.testing.assert_allclose(sm.max(), -0.5 * np.log(2 * np.pi * 0.25), atol=2e-3)
    # non-finite ends are trimmed (zero-probability regions, :93-96)
    lp2 = lp.copy(); lp2[:10] = -np.inf; lp2[-5:] = np.nan
    s2 = grid_and_sample(lp2, xs, 0.5)[0]
    assert abs(s2 - 0.4) < 5e-3
    # LOESS reproduces a quadratic exactly
    np.testing.assert_allclose(loess(xs, lp, np.array([-1.234, 0.0, 2.5])), -(np.array([-1.234, 0.0, 2.5]) - 0.4) ** 2 / 0.5 + 17, atol=1e-9)


def test_logpdf_mixed_theta_bookkeeping():
    s = O.load_sim(3.0, (64, 64), "P", np.float64, beam_fwhm=1.0)
    ds, proj = s["ds"], s["proj"]
    th = ThetaDataSet(ds, s["Cfs"], s["Cten"])
    fo, po = ds.mix(s["f"], s["phi"])
    base = ds.logpdf_mixed(fo, po)
    # parameters not named in θ leave everything fiducial; naming them at their fiducial value changes nothing either
    np.testing.assert_allclose(th.logpdf_mixed(fo, po), base, rtol=1e-13)
    np.testing.assert_allclose(th.logpdf_mixed(fo, po, r=0.2, Aphi=1.0), base, rtol=1e-10)
    # θ-dependence: more tensor power -> different D, Cf; the conditional of Aϕ given (f°, ϕ°) peaks near the truth
    assert abs(th.logpdf_mixed(fo, po, r=0.4)[0] - base[0]) > 1.0
    xs = np.linspace(0.6, 1.5, 10)
    lps = np.array([th.logpdf_mixed(fo, po, Aphi=a)[0] for a in xs])
    assert np.all(np.isfinite(lps)) and 0.7 < xs[np.argmax(lps)] < 1.4
    # the G used in the mixed space is the identity at the fiducial amplitude and shrinks ϕ° weights otherwise
    assert np.allclose(th.G(1.0)[ds.Cphi > 0], 1.0) and not np.allclose(th.G(1.3)[ds.Cphi > 0], 1.0)


def test_bandpower_rescaling_semantics():
    """findbin / bandpower_rescale (src/proj_lambert.jl:402-408): half-open bins, out of range keeps amplitude 1; the product's host
    algebra (vectorised) equals the oracle's literal restatement"""
    from oracle.theta import findbin, bandpower_rescale
    import cmblensing_jl_amd as C
    le = [100, 500, 1000, 2500]
    l = np.array([0.0, 99.9, 100.0, 499.99, 500.0, 999.0, 1000.0, 2499.9, 2500.0, 9000.0])
    want = np.array([3, 3, 0, 0, 1, 1, 2, 2, 3, 3])                      # 3 = the appended unit amplitude
    np.testing.assert_array_equal(findbin(le, l), want)
    np.testing.assert_array_equal(C.findbin(le, l), want)
    amps = [2.0, 3.0, 5.0]
    np.testing.assert_allclose(bandpower_rescale(np.ones(10), want, amps), [1, 1, 2, 2, 3, 3, 5, 5, 1, 1])
    np.testing.assert_allclose(C.bandpower_rescale(np.ones(10), want, amps), [1, 1, 2, 2, 3, 3, 5, 5, 1, 1])
    # the covariance object on a geometry stand-in: only EE (and TE for IP) are rescaled, BB never
    proj = O.Proj(64, 64, 3.0, np.float64)
    cls = O.load_camb()["unlensed_scalar"]
    pcls = {k: C.Cls(v.ell, v.cl) for k, v in cls.items()}
    cov = C.BinRescaledCov("P", proj, pcls, {"EE": (le, "AEE")})
    base = O.HarmOp.from_cls("P", proj, cls)
    np.testing.assert_allclose(cov().p, base.d, rtol=1e-13)
    got = cov(AEE=amps).p
    idx = findbin(le, proj.lmag)
    np.testing.assert_allclose(got[0], bandpower_rescale(base.d[0], idx, amps), rtol=1e-13)
    np.testing.assert_allclose(got[1], base.d[1], rtol=1e-13)
    cov3 = C.BinRescaledCov("IP", proj, pcls, {"TT": (le, "ATT"), "TE": (le[:3], "ATE")})
    b3 = O.HarmOp.from_cls("IP", proj, cls)
    g3 = cov3(ATT=amps, ATE=[0.5, 0.25]).p
    np.testing.assert_allclose(g3[0], bandpower_rescale(b3.te[0], idx, amps), rtol=1e-13)
    np.testing.assert_allclose(g3[1], bandpower_rescale(b3.te[1], findbin(le[:3], proj.lmag), [0.5, 0.25]), rtol=1e-13)
    np.testing.assert_allclose(g3[2], g3[1]); np.testing.assert_allclose(g3[3], b3.te[3]); np.testing.assert_allclose(g3[4], b3.bb)


def test_bandpower_theta_changes_only_cf():
    s = O.load_sim(3.0, (64, 64), "P", np.float64, beam_fwhm=1.0)
    ds = s["ds"]
    le = [100, 600, 1500, 3000]
    th = ThetaDataSet(ds, s["Cfs"], s["Cten"], bands={"AEE": ([0], le)})
    fo, po = ds.mix(s["f"], s["phi"])
    base = ds.logpdf_mixed(fo, po)
    np.testing.assert_allclose(th.logpdf_mixed(fo, po, AEE=np.ones(3)), base, rtol=1e-12)
    assert abs(th.logpdf_mixed(fo, po, AEE=np.array([1.3, 1.0, 0.8]))[0] - base[0]) > 1.0
    dsθ, ldD, ldG = th.at(AEE=np.array([1.3, 1.0, 0.8]))
    assert ldD == 0.0 and ldG == 0.0                                     # D, G do not name the amplitudes (src/dataset.jl:316-328)
    np.testing.assert_allclose(th.logpdf_mixed(fo, po), base, rtol=1e-12)  # and the fiducial operators are restored
