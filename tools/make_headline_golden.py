#!/usr/bin/env python3
"""tests/golden/headline_*.npz: the float64 oracle's answers of tests/test_gpu_headline_parity.py as committed, sampled data.

Rounds 4-5 recomputed these on the GPU box's host cores inside the GPU test run (200 s of its 646 s).  The inputs are
now defined on the CPU alone -- seeded oracle simulations, ROUNDED to float32 where the device runs single precision -- so the oracle's
outputs can be computed once here and travel as 10⁴ seeded sample values per field (+ norms, scalars, input fingerprints):

    headline_grad_{P,IP}.npz   ∇logpdf(Mixed) at 1024² QU / T+QU (bench.py's workload: θpix 2′, 1° apodised border mask, n = 7), evaluated at
                               the float32-rounded (f°, ϕ°, d), both settings of the alias quirk
    headline_flow_2048.npz     L*f, L'g and the pullback of L*f at 2048² QU, n = 10 (BASELINE configs[4])
    headline_qe_2048.npz       quadratic_estimate(:EB) at 2048² QU on the oracle's simulated data (AL inside |l| < 5000, ϕqe)

    python tools/make_headline_golden.py [grad_P grad_IP flow qe]
tests/test_golden.py::test_headline_goldens_are_the_oracle re-derives the inputs' fingerprints and one cheap output on the CPU (drift guard)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from oracle.lenseflow import LenseFlow as OLenseFlow

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
NS = 10000
PM = dict(pad_deg=1.0, apod_deg=1.0)


def sample_idx(n, seed):
    return np.random.default_rng(seed).choice(n, size=NS, replace=False)


def fingerprint(a):
    a = np.asarray(a)
    return np.array([np.sqrt(np.sum(np.abs(a) ** 2)), np.abs(a.ravel()[:: max(1, a.size // 997)]).sum()])


def put(out, key, a, seed):
    a = np.asarray(a)
    idx = sample_idx(a.size, seed)
    out[key + "_idx"], out[key + "_val"], out[key + "_l2"] = idx, a.ravel()[idx], np.sqrt(np.sum(np.abs(a) ** 2))


def r32(a):
    """what the device holds after the upload of a float64 array into a float32 context"""
    a = np.asarray(a)
    return a.astype(np.complex64).astype(np.complex128) if np.iscomplexobj(a) else a.astype(np.float32).astype(np.float64)


def grad_inputs(pol):
    """(oracle dataset, f°, ϕ°) of the 1024² comparison: the oracle's simulation, mixed by the oracle, everything rounded to float32"""
    so = O.load_sim(2.0, 1024, pol, np.float64, pixel_mask=PM, nsteps=7)
    ods = so["ds"]
    fo, po = ods.mix(so["f"], so["phi"])
    ods.d = r32(so["d"])
    return so, ods, r32(fo), r32(po)


def make_grad(pol):
    t0 = time.time()
    so, ods, fo, po = grad_inputs(pol)
    out = dict(fp_d=fingerprint(ods.d), fp_fo=fingerprint(fo), fp_po=fingerprint(po), fp_Nphi=fingerprint(ods.Nphi))
    for quirk in (False, True):
        lp, gf, gp = ods.grad_logpdf_mixed(fo, po, alias_quirk=quirk)
        q = "q1" if quirk else "q0"
        out["lp_" + q] = np.asarray(lp)
        put(out, "gf_" + q, gf, 301 + quirk)
        put(out, "gp_" + q, gp, 303 + quirk)
        print(f"grad {pol} quirk={quirk}: lp {lp}  {time.time() - t0:.0f} s", flush=True)
    out["seconds"] = time.time() - t0
    np.savez_compressed(os.path.join(GOLD, f"headline_grad_{pol}.npz"), **out)


def flow_inputs():
    N, n = 2048, 10
    oproj = O.Proj(N, N, 2.0, np.float64)
    cl = O.load_camb()["unlensed_total"]
    Cphi = O.cl_to_2d(cl["pp"], oproj)
    Cf = np.stack([O.cl_to_2d(cl["EE"], oproj), O.cl_to_2d(cl["BB"], oproj) + 0.05 * O.cl_to_2d(cl["EE"], oproj)])
    f = O.from_harm(oproj, np.sqrt(Cf) * O.rfft2(O.white_noise(1, (1, 2, N, N), np.float64)))
    g = O.from_harm(oproj, np.sqrt(Cf) * O.rfft2(O.white_noise(4, (1, 2, N, N), np.float64)))
    phi = O.irfft2(np.sqrt(Cphi) * O.rfft2(O.white_noise(2, (1, 1, N, N), np.float64)), N)
    return oproj, n, f, O.rfft2(g), phi


def make_flow():
    t0 = time.time()
    oproj, n, f, gl, phi = flow_inputs()
    OL = OLenseFlow(oproj, phi, n)
    out = dict(fp_f=fingerprint(f), fp_gl=fingerprint(gl), fp_phi=fingerprint(phi), n=n)
    Lf = OL.apply(f)
    put(out, "Lf", Lf, 311)
    print(f"flow: L*f {time.time() - t0:.0f} s", flush=True)
    put(out, "adj", OL.adj(gl), 312)
    print(f"flow: L'g {time.time() - t0:.0f} s", flush=True)
    f0, df, dp = OL.grad_apply(Lf, gl)
    put(out, "f0", f0, 313); put(out, "df", df, 314); put(out, "dp", dp, 315)
    out["seconds"] = time.time() - t0
    print(f"flow: pullback {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(os.path.join(GOLD, "headline_flow_2048.npz"), **out)


def make_qe():
    t0 = time.time()
    so = O.load_sim(2.0, (2048, 2048), "P", np.float64, beam_fwhm=1.0, pixel_mask=None, Nbatch=1)     # = tests/test_gpu_parity.py::_dataset_pair
    ods = so["ds"]
    planes = lambda op: {k: op.d[i] for i, k in enumerate(["E", "B"])}
    TF = {k: planes(ods.Mf)[k] * planes(ods.B)[k] for k in ("E", "B")}
    dd = {k: so["d"][:, i:i + 1] for i, k in enumerate(("E", "B"))}
    pq, AL, Nphi = O.quadratic_estimate(so["proj"], "EB", dd, dd, planes(ods.Cf), planes(ods.Cftilde), planes(ods.Cn), ods.Cphi, TF)
    m = (ods.Cphi > 0) & (so["proj"].lmag < 5000)
    out = dict(fp_d=fingerprint(so["d"]), fp_Nphi=fingerprint(ods.Nphi))
    put(out, "phiqe", pq, 321)
    ALm = np.where(m, AL, 0.0)                                             # compared inside the support of the normalisation integral only
    put(out, "ALm", ALm, 322)
    out["seconds"] = time.time() - t0
    print(f"qe: {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(os.path.join(GOLD, "headline_qe_2048.npz"), **out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["grad_P", "grad_IP", "flow", "qe"]
    for w in what:
        if w.startswith("grad_"): make_grad(w[5:])
        elif w == "flow": make_flow()
        elif w == "qe": make_qe()
