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
