"""Full-size driver parity against COMMITTED oracle data (the tests/golden/config2_cg.json pattern): computations the float64 oracle
needs many minutes for are run once in the build container (tools/make_config3_golden.py) and travel as a small file of sampled values.

    one MAP_joint step, 1024² T+QU fp32 (BASELINE configs[2]; src/maximization.jl:160-206): f after a fixed 10-iteration Wiener CG,
    ∇ϕ°, the step direction, α, logpdf before / after -- Python driver and cmbl_map_joint_step.

The inputs are regenerated here with oracle.load_sim (simulation only) and checked against the fingerprints stored with the golden data,
so a drift of the NumPy generator or of the spectra shows up as a fingerprint mismatch, not as a parity failure."""
import os
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as O
from _tol import close, scalars_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config3_map_joint_step.npz")


def _fingerprint(a):
    a = np.asarray(a)
    return np.array([np.sqrt(np.sum(np.abs(a) ** 2)), np.abs(a.ravel()[:: max(1, a.size // 997)]).sum()])


def _rel_sample(got, idx, val):
    g = np.asarray(got).ravel()[idx]
    return float(np.linalg.norm(g - val) / np.linalg.norm(val))


@pytest.mark.skipif(not os.path.exists(GOLD), reason="tests/golden/config3_map_joint_step.npz not generated (tools/make_config3_golden.py)")
def test_map_joint_step_1024_IQU_fp32_vs_committed_oracle():
    import cmblensing_jl_amd as C
    g = np.load(GOLD)
    pm = dict(pad_deg=1.0, apod_deg=1.0)
    so = O.load_sim(2.0, 1024, "IP", np.float64, pixel_mask=pm, nsteps=7)                    # the inputs of the golden run (simulation only)
    ods = so["ds"]
    for k, a in (("d", so["d"]), ("f", so["f"]), ("phi", so["phi"]), ("Nphi", ods.Nphi)):
        np.testing.assert_allclose(_fingerprint(a), g["fp_" + k], rtol=1e-9, err_msg=f"simulated input {k} differs from the golden run's")
    camb = so["cls"]
    cls = {grp: {k: C.Cls(v.ell, v.cl) for k, v in camb[grp].items()} for grp in ("unlensed_scalar", "tensor", "total")}
    sd = C.load_sim(2.0, 1024, "IP", cls, T=torch.float32, pixel_mask=pm, nsteps=7, Nphi=ods.Nphi * 2)
    ds, p = sd["ds"], sd["proj"]
    ds.set_data(C.Field(p, p.tensor(so["d"]), C.HARMONIC))
    phi0 = C.Field(p, p.tensor(np.zeros_like(so["phi"])), C.FOURIER)
    for name, step in (("MAP_joint_step", C.MAP_joint_step), ("cmbl_map_joint_step", C.MAP_joint_step_native)):
        st = step(ds, phi0, alpha_tol=1e-4, cg_tol=0.0, cg_nsteps=10)
        # the f-step: a fixed 10-iteration CG is the same computation on both sides (8-step CG iterate class of tests/_tol.py: 4e-6 at 64²-256²;
        # T+QU at 1024² carries the TE block's cancellations like the f-gradient of test_gpu_headline_parity.py: 1.2e-4 class)
        e_f = _rel_sample(st["f"].arr.cpu().numpy(), g["f_idx"], g["f_val"])
        e_g = _rel_sample(st["grad_phi"].arr.cpu().numpy(), g["grad_phi_idx"], g["grad_phi_val"]) if "grad_phi" in st else None
        e_p = _rel_sample(st["phi"].arr.cpu().numpy(), g["phi_idx"], g["phi_val"])
        print(f"{name}: f {e_f:.2e}  grad_phi {e_g}  phi {e_p:.2e}  alpha {st['alpha']:.5f} vs {float(g['alpha']):.5f}")
        assert e_f < 2e-4, (name, e_f)
        if e_g is not None:
            assert e_g < 2e-4, (name, e_g)
        if "dphi" in st:
            assert _rel_sample(st["dphi"].arr.cpu().numpy(), g["dphi_idx"], g["dphi_val"]) < 2e-4
        # Brent here vs SciPy's bounded Brent in the oracle: the same minimiser to the tolerance of the search, the same objective value
        assert abs(st["alpha"] - float(g["alpha"])) < 5e-3 * max(1.0, float(g["alpha"])), (st["alpha"], float(g["alpha"]))
        scalars_close(f"{name} 1024² T+QU: logpdf after the step", st["logpdf"], g["logpdf"], rtol=2e-6)
        if "logpdf_before" in st:
            scalars_close(f"{name} 1024² T+QU: logpdf before the step", st["logpdf_before"], g["logpdf_before"], rtol=2e-6)
        assert e_p < 2e-2, (name, e_p)                                                        # ϕ = α · direction: carries the α tolerance
        if "cg_hist" in st and len(st["cg_hist"]) == len(g["cg_res"]):
            res = np.array([float(h[1][0]) for h in st["cg_hist"]])
            np.testing.assert_allclose(res, g["cg_res"], rtol=2e-3)


GOLD4 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_gibbs_pass.npz")


@pytest.mark.skipif(not os.path.exists(GOLD4), reason="tests/golden/config4_gibbs_pass.npz not generated (tools/make_config4_golden.py)")
def test_gibbs_pass_1024_IQU_fp32_vs_committed_oracle():
    """BASELINE configs[3], one chain's share: `sample_f` (fixed 10-iteration CG) and one `hmc_step` (N = 3, injected momenta / log u) at
    1024² T+QU fp32 against committed float64 oracle data -- Python driver and cmbl_hmc_step (src/sampling.jl:405-418, :14-46;
    src/maximization.jl:56-62)."""
    import cmblensing_jl_amd as C
    g = np.load(GOLD4)
    pm = dict(pad_deg=1.0, apod_deg=1.0)
    so = O.load_sim(2.0, 1024, "IP", np.float64, pixel_mask=pm, nsteps=7)
    ods = so["ds"]
    N, P = 1024, ods.P
    wf, wn = (O.white_noise(s, (1, P, N, N), np.float64) for s in (7, 8))
    wp = O.white_noise(9, (1, 1, N, N), np.float64)
    logu = np.log(np.random.default_rng(3).random(1))
    for k, a in (("d", so["d"]), ("f", so["f"]), ("phi", so["phi"]), ("Nphi", ods.Nphi), ("wf", wf), ("wn", wn), ("wp", wp)):
        np.testing.assert_allclose(_fingerprint(a), g["fp_" + k], rtol=1e-9, err_msg=f"input {k} differs from the golden run's")
    np.testing.assert_allclose(logu, g["log_u"], rtol=1e-12)
    camb = so["cls"]
    cls = {grp: {k: C.Cls(v.ell, v.cl) for k, v in camb[grp].items()} for grp in ("unlensed_scalar", "tensor", "total")}
    sd = C.load_sim(2.0, 1024, "IP", cls, T=torch.float32, pixel_mask=pm, nsteps=7, Nphi=ods.Nphi * 2)
    ds, p = sd["ds"], sd["proj"]
    ds.set_data(C.Field(p, p.tensor(so["d"]), C.HARMONIC))
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    nleap, eps, ncg = int(g["nleap"]), float(g["eps"]), int(g["ncg"])
    # (a) posterior sample of f at the true ϕ: the fixed short CG stops at the same iterate on both sides
    f_s, hist = C.sample_f(ds, F(so["phi"], C.FOURIER), wf, wn, tol=0.0, nsteps=ncg)
    e_f = _rel_sample(f_s.arr.cpu().numpy(), g["f_sample_idx"], g["f_sample_val"])
    print(f"sample_f 1024² T+QU: {e_f:.2e}")
    assert e_f < 2e-4, e_f                                                                   # the f-step class of the MAP_joint test above
    if len(hist) == len(g["cg_res"]):
        np.testing.assert_allclose(np.array([float(h[1][0]) for h in hist]), g["cg_res"], rtol=2e-3)
    # (b) HMC from the mixed truth.  The mix itself first (the state the golden trajectory started from) ...
    fo, po = ds.mix(F(so["f"], C.HARMONIC), F(so["phi"], C.FOURIER))
    assert _rel_sample(fo.to(C.MAP).arr.cpu().numpy(), g["fo_idx"], g["fo_val"]) < 2e-4
    assert _rel_sample(po.to(C.FOURIER).arr.cpu().numpy(), g["phio_in_idx"], g["phio_in_val"]) < 1e-5
    scalars_close("1024² T+QU: H0 = logpdf(Mixed) at the start of the trajectory", ds.logpdf_mixed(fo, po), g["H0"], rtol=2e-6)
    # ... then the proposal (the oracle accepted it: its ϕ° IS the proposal; always_accept here so that a ΔH within fp32 noise of log u
    # cannot turn the comparison into one against the unchanged state), ΔH and, where fp32 can decide it, the decision itself
    assert bool(g["accept"][0])
    dH_tol = 2e-6 * abs(float(g["H0"][0]))                                                   # ΔH = H1 - H0: two logpdfs of fp32 accuracy each
    for name, step in (("hmc_step", C.hmc_step), ("cmbl_hmc_step", lambda *a, **k: C.hmc_step_native(a[0], a[1], a[2], white_p=a[3], log_u=a[4], **k))):
        x, dH, acc = step(ds, fo, po, wp, logu, N=nleap, eps=eps, always_accept=True)
        e_x = _rel_sample(x.to(C.FOURIER).arr.cpu().numpy(), g["phio_out_idx"], g["phio_out_val"])
        print(f"{name} 1024² T+QU: proposal {e_x:.2e}  dH {float(dH[0]):.4f} vs {float(g['dH'][0]):.4f} (tolerance {dH_tol:.2f})")
        assert e_x < 2e-4, (name, e_x)
        assert abs(float(dH[0]) - float(g["dH"][0])) < dH_tol, (name, dH, g["dH"])
        if abs(float(g["dH"][0]) - float(logu[0])) > 2 * dH_tol:
            _, _, acc2 = step(ds, fo, po, wp, logu, N=nleap, eps=eps)
            assert bool(acc2[0]) == bool(g["accept"][0])
