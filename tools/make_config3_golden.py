#!/usr/bin/env python3
"""tests/golden/config3_map_joint_step.npz: the float64 oracle's first MAP_joint step (src/maximization.jl:160-206) on BASELINE config 3
(1024² T+QU, θpix 2′, 1° apodised border mask, LenseFlow n = 7 -- bench.py's T+QU workload), from ϕ = 0 with a FIXED 10-iteration Wiener
CG (cg_tol = 0: the f-step is then the same computation on both sides and the ϕ-step can be compared tightly).  The oracle needs ~10-20 min
for this on 8 cores, too long for the GPU test run, so its answer is committed as data: f after the CG, ∇ϕ° and the step direction at
10⁴ seeded sample modes each (+ their norms), α, the logpdf before / after, the CG residual history, and fingerprints of the simulated
inputs (the GPU test regenerates the inputs with oracle.load_sim -- simulation only, seconds -- and checks the fingerprints first).
    python tools/make_config3_golden.py            (tests/test_gpu_fullsize_golden.py reads the file)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O

N, POL, NS = 1024, "IP", 10000
PM = dict(pad_deg=1.0, apod_deg=1.0)


def sample_idx(n, seed):
    return np.random.default_rng(seed).choice(n, size=NS, replace=False)


def fingerprint(a):
    a = np.asarray(a)
    return np.array([np.sqrt(np.sum(np.abs(a) ** 2)), np.abs(a.ravel()[:: max(1, a.size // 997)]).sum()])


if __name__ == "__main__":
    t0 = time.time()
    so = O.load_sim(2.0, N, POL, np.float64, pixel_mask=PM, nsteps=7)
    ods = so["ds"]
    print(f"load_sim {time.time() - t0:.0f} s", flush=True)
    phi0 = np.zeros_like(so["phi"])
    st = O.map_joint_step(ods, phi0, alpha_tol=1e-4, cg_tol=0.0, cg_nsteps=10)
    out = dict(alpha=st["alpha"], logpdf=np.asarray(st["logpdf"]), logpdf_before=np.asarray(st["logpdf_before"]),
               cg_res=np.array([float(h[1][0]) for h in st["cg_hist"]]), dphi_norm=st["dphi_norm"],
               fp_d=fingerprint(so["d"]), fp_f=fingerprint(so["f"]), fp_phi=fingerprint(so["phi"]), fp_Nphi=fingerprint(ods.Nphi))
    for k, seed in (("f", 101), ("grad_phi", 102), ("dphi", 103), ("phi", 104)):
        a = np.asarray(st[k])
        idx = sample_idx(a.size, seed)
        out[k + "_idx"], out[k + "_val"], out[k + "_l2"] = idx, a.ravel()[idx], np.sqrt(np.sum(np.abs(a) ** 2))
    out["seconds"] = time.time() - t0
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config3_map_joint_step.npz")
    np.savez_compressed(path, **out)
    print(path, f"alpha {st['alpha']:.5f} logpdf {out['logpdf_before']} -> {out['logpdf']}  {out['seconds']:.0f} s")
