#!/usr/bin/env python3
"""tests/golden/config4_gibbs_pass.npz: the float64 oracle's share of ONE chain of BASELINE config 4 (1024² T+QU `sample_joint`,
θpix 2′, 1° apodised border mask, LenseFlow n = 7) -- the two field updates of a Gibbs pass (src/sampling.jl:388-418) with every random
draw injected so that the device can be driven with the same ones:
    (a) `sample_f` (src/maximization.jl:56-62) at the true ϕ with a FIXED 10-iteration Wiener CG (tol = 0: the same iterate on both sides),
    (b) `hmc_step` (src/sampling.jl:405-418 over :14-46) from the mixed truth with N = 3 leapfrog steps, ϵ = 0.01: proposal ϕ°, ΔH, accept.
The oracle needs ~10 min for this on 8 cores -- too long for the GPU test run -- so its answer is committed as data: 10⁴ seeded sample
modes of each field (+ norms), the scalars, the CG residual history, and fingerprints of the simulated inputs and of the injected draws
(the GPU test regenerates both from the seeds and checks the fingerprints first).
    python tools/make_config4_golden.py            (tests/test_gpu_fullsize_golden.py reads the file)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O

N, POL, NS = 1024, "IP", 10000
PM = dict(pad_deg=1.0, apod_deg=1.0)
SEED_F, SEED_N, SEED_P, SEED_U = 7, 8, 9, 3            # injected draws (the seeds of tests/test_gpu_drivers.py::test_hmc_and_gibbs_step)
NLEAP, EPS, NCG = 3, 0.01, 10


def sample_idx(n, seed):
    return np.random.default_rng(seed).choice(n, size=NS, replace=False)


def fingerprint(a):
    a = np.asarray(a)
    return np.array([np.sqrt(np.sum(np.abs(a) ** 2)), np.abs(a.ravel()[:: max(1, a.size // 997)]).sum()])


def draws(P):
    wf, wn = (O.white_noise(s, (1, P, N, N), np.float64) for s in (SEED_F, SEED_N))
    wp = O.white_noise(SEED_P, (1, 1, N, N), np.float64)
    logu = np.log(np.random.default_rng(SEED_U).random(1))
    return wf, wn, wp, logu


if __name__ == "__main__":
    t0 = time.time()
    so = O.load_sim(2.0, N, POL, np.float64, pixel_mask=PM, nsteps=7)
    ods = so["ds"]
    print(f"load_sim {time.time() - t0:.0f} s", flush=True)
    wf, wn, wp, logu = draws(ods.P)
    out = dict(fp_d=fingerprint(so["d"]), fp_f=fingerprint(so["f"]), fp_phi=fingerprint(so["phi"]), fp_Nphi=fingerprint(ods.Nphi),
               fp_wf=fingerprint(wf), fp_wn=fingerprint(wn), fp_wp=fingerprint(wp), log_u=logu, nleap=NLEAP, eps=EPS, ncg=NCG)
    f_s, hist = O.sample_f(ods, so["phi"], wf, wn, tol=0.0, nsteps=NCG)
    out["cg_res"] = np.array([float(h[1][0]) for h in hist])
    print(f"sample_f {time.time() - t0:.0f} s", flush=True)
    fo, po = ods.mix(so["f"], so["phi"])
    x, dH, acc = O.hmc_step(ods, fo, po, wp, logu, N=NLEAP, eps=EPS)
    # the proposal itself (x equals it only when accepted): one more integration would double the run time, so store what decides it too
    out.update(dH=np.asarray(dH), accept=np.asarray(acc), H0=np.asarray(ods.logpdf_mixed(fo, po)))
    print(f"hmc_step {time.time() - t0:.0f} s  dH {dH} accept {acc}", flush=True)
    for k, a, seed in (("f_sample", f_s, 201), ("phio_out", x, 202), ("fo", fo, 203), ("phio_in", po, 204)):
        a = np.asarray(a)
        idx = sample_idx(a.size, seed)
        out[k + "_idx"], out[k + "_val"], out[k + "_l2"] = idx, a.ravel()[idx], np.sqrt(np.sum(np.abs(a) ** 2))
    out["seconds"] = time.time() - t0
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config4_gibbs_pass.npz")
    np.savez_compressed(path, **out)
    print(path, f"dH {dH} accept {acc}  {out['seconds']:.0f} s")
