"""Golden vectors (tests/golden/*.npz, made by tools/make_golden.py from the float64 oracle).
CPU: the oracle still reproduces them (drift guard).  GPU: the HIP engine reproduces them in fp64 and fp32."""
import os

import numpy as np
import pytest

import oracle as O
from _tol import sample_close, scalars_close
from oracle.lenseflow import LenseFlow as OLF

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_sample(z, key, arr, rtol):
    flat = np.asarray(arr).ravel()
    idx, val, l2 = z[f"{key}.idx"], z[f"{key}.val"], z[f"{key}.l2"]
    err = float(np.linalg.norm(flat[idx] - val) / np.linalg.norm(val))
    sample_close("golden " + key, err, rtol)
    assert abs(np.sqrt(np.sum(np.abs(flat) ** 2)) - l2) < 10 * rtol * l2, key


def flow_inputs(Ny, Nx, P):
    camb = O.load_camb()
    proj = O.Proj(Ny, Nx, 2.0, np.float64)
    cl = camb["unlensed_total"]
    Cphi = O.cl_to_2d(cl["pp"], proj)
    C = (O.cl_to_2d(cl["TT"], proj)[None] if P == 1 else
         np.stack([O.cl_to_2d(cl["EE"], proj), O.cl_to_2d(cl["BB"], proj) + 0.05 * O.cl_to_2d(cl["EE"], proj)]))
    f = O.from_harm(proj, np.sqrt(C) * O.rfft2(O.white_noise(1, (1, P, Nx, Ny), np.float64)))
    g = O.from_harm(proj, np.sqrt(C) * O.rfft2(O.white_noise(11, (1, P, Nx, Ny), np.float64)))
    phi = O.irfft2(np.sqrt(Cphi) * O.rfft2(O.white_noise(2, (1, 1, Nx, Ny), np.float64)), Ny)
    return proj, f, g, phi


@pytest.mark.parametrize("Ny,Nx,P", [(64, 128, 2), (128, 128, 1)])
def test_oracle_reproduces_golden_flows(Ny, Nx, P):
    z = np.load(os.path.join(G, f"flow_{Ny}x{Nx}_P{P}.npz"))
    proj, f, g, phi = flow_inputs(Ny, Nx, P)
    L = OLF(proj, phi, 7)
    Lf = L.apply(f)
    check_sample(z, "Lf", Lf, 1e-12)
    check_sample(z, "Linvf", L.inv(f), 1e-12)
    check_sample(z, "Ladjg", L.adj(O.rfft2(g)), 1e-12)
    _, df, dp = L.grad_apply(Lf, O.rfft2(g), alias_quirk=True)
    check_sample(z, "grad_q1.dphi", dp, 1e-11)
    a, b = z["adjoint_identity"]
    assert abs(a - b) < 1e-12 * abs(a)


def test_oracle_reproduces_golden_posterior():
    z = np.load(os.path.join(G, "posterior_P_64x128.npz"))
    s = O.load_sim(3.0, (64, 128), "P", np.float64, beam_fwhm=3.0, pixel_mask=dict(pad_deg=0.4, apod_deg=0.4))
    ds = s["ds"]
    fo, po = ds.mix(s["f"], s["phi"])
    np.testing.assert_allclose(ds.logpdf_mixed(fo, po), z["logpdf_mixed"], rtol=1e-12)
    check_sample(z, "d", s["d"], 1e-12)


def test_headline_goldens_are_the_oracle():
    """Drift guard of tests/golden/headline_*.npz and config4_gibbs_pass.npz (the oracle's answers at 1024² / 2048², committed as sampled data
    because the GPU test run has no time to recompute them: tools/make_headline_golden.py, tools/make_config4_golden.py).  The CPU suite has no
    time for the full evaluations either (3-4 minutes on 8 cores), so it re-derives the INPUT fingerprints of the 1024² QU case and its cheapest
    output -- logpdf(Mixed) at the float32-rounded (f°, ϕ°, d): one inverse flow + the data model -- and requires the others to exist with
    the fields the GPU tests read."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mhg", os.path.join(os.path.dirname(G), "..", "tools", "make_headline_golden.py"))
    mhg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mhg)
    g = np.load(os.path.join(G, "headline_grad_P.npz"))
    so, ods, fo, po = mhg.grad_inputs("P")
    for k, a in (("d", ods.d), ("fo", fo), ("po", po), ("Nphi", ods.Nphi)):
        np.testing.assert_allclose(mhg.fingerprint(a), g["fp_" + k], rtol=1e-9, err_msg=k)
    scalars_close("headline golden: logpdf(Mixed) 1024² QU", ods.logpdf_mixed(fo, po), g["lp_q0"], rtol=1e-12)
    np.testing.assert_allclose(g["lp_q0"], g["lp_q1"], rtol=0)                   # the alias quirk touches the gradient only
    need = {"headline_grad_IP.npz": ["lp_q0", "gf_q1_val", "gp_q0_idx", "fp_fo"], "headline_flow_2048.npz": ["Lf_val", "adj_val", "f0_val", "df_val", "dp_val", "fp_gl"],
            "headline_qe_2048.npz": ["phiqe_val", "ALm_val", "fp_d"], "config4_gibbs_pass.npz": ["f_sample_val", "phio_out_val", "dH", "H0", "cg_res", "fp_wp"]}
    for name, keys in need.items():
        z = np.load(os.path.join(G, name))
        for k in keys:
            assert k in z.files and np.all(np.isfinite(np.asarray(z[k]).view(np.float64) if np.iscomplexobj(z[k]) else z[k])), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("Ny,Nx,P", [(64, 128, 2), (128, 128, 1)])
def test_gpu_reproduces_golden_flows(prec, Ny, Nx, P):
    import torch
    import cmblensing_jl_amd as C
    z = np.load(os.path.join(G, f"flow_{Ny}x{Nx}_P{P}.npz"))
    proj, f, g, phi = flow_inputs(Ny, Nx, P)
    p = C.ProjLambert(Ny, Nx, 2.0, torch.float32 if prec == "f32" else torch.float64)
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    L = C.LenseFlow(p, 7)(F(phi, C.MAP))
    tol, gtol = (5e-5, 1.4e-4) if prec == "f32" else (1e-10, 1e-9)      # fp32 measured: flows 2.5e-6 (forward) / 1.6e-5 (adjoint), δϕ 4.4e-5
    Lf = L * F(f, C.MAP)
    check_sample(z, "Lf", Lf.arr.cpu().numpy(), tol)
    check_sample(z, "Linvf", L.ldiv(F(f, C.MAP)).arr.cpu().numpy(), tol)
    gl = F(O.rfft2(g), C.FOURIER)
    check_sample(z, "Ladjg", (L.adjoint * gl).arr.cpu().numpy(), tol)
    check_sample(z, "Linvadjg", L.adjoint.ldiv(gl).arr.cpu().numpy(), tol)
    for q in (0, 1):
        dp, df, _ = L.gradient(C.FLOW_FWD, Lf, gl, alias_quirk=bool(q))
        check_sample(z, f"grad_q{q}.df", df.arr.cpu().numpy(), tol)
        check_sample(z, f"grad_q{q}.dphi", dp.arr.cpu().numpy(), gtol)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("pol,Nside", [("P", (64, 128)), ("IP", (64, 64))])
def test_gpu_reproduces_golden_posterior(prec, pol, Nside):
    import torch
    import cmblensing_jl_amd as C
    from bench import synthetic_cls
    z = np.load(os.path.join(G, f"posterior_{pol}_{Nside[0]}x{Nside[1]}.npz"))
    s = C.load_sim(3.0, Nside, pol, synthetic_cls(), T=torch.float32 if prec == "f32" else torch.float64, beam_fwhm=3.0,
                   pixel_mask=dict(pad_deg=0.4, apod_deg=0.4))
    ds = s["ds"]
    check_sample(z, "d", s["d"].arr.cpu().numpy(), 2.1e-6 if prec == "f32" else 1e-9)              # measured 7.0e-7
    fo, po = ds.mix(s["f"], s["phi"])
    lp, gf, gp = ds.gradient_logpdf_mixed(fo, po)
    scalars_close("golden logpdf_mixed", lp, z["logpdf_mixed"], rtol=2.3e-7 if prec == "f32" else 1e-9)      # 7.7e-8
    check_sample(z, "grad_fo", gf.arr.cpu().numpy(), 5.5e-5 if prec == "f32" else 1e-8)                       # 1.8e-5
    check_sample(z, "grad_phio", gp.arr.cpu().numpy(), 6.4e-6 if prec == "f32" else 1e-8)                     # 2.1e-6
    fw, hist = ds.argmaxf_logpdf(s["phi"], tol=0.0, nsteps=8)
    scalars_close("golden cg_res", [h[1][0] for h in hist], z["cg_res"], rtol=1e-5 if prec == "f32" else 1e-7)   # 3.4e-6
    check_sample(z, "cg_f", fw.arr.cpu().numpy(), 3.7e-6 if prec == "f32" else 1e-8)                          # 1.3e-6
