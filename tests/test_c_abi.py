"""The boundary from a plain-C host: tests/c_abi/lenseflow.c is compiled with gcc against include/cmblens.h alone (C99, no HIP
headers, no HIP link) and dlopen()s the library -- what Julia's `ccall` does.  CPU: it builds, and without a device every entry
point reports CMBL_ERR_HIP instead of crashing.  GPU: ctx -> set_phi -> apply -> grad agree with the float64 oracle vectors of
tests/golden/cabi_lenseflow.bin (tools/make_cabi_golden.py) to 1e-9; tests/c_abi/posterior.c does the same for the dataset /
posterior entry points (cmbl_dataset_*, cmbl_logpdf_mixed, cmbl_grad_logpdf_mixed)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "lenseflow.c")
GOLD = os.path.join(ROOT, "tests", "golden", "cabi_lenseflow.bin")
SRC_POST = os.path.join(ROOT, "tests", "c_abi", "posterior.c")
GOLD_POST = os.path.join(ROOT, "tests", "golden", "cabi_posterior.bin")


def _build(tmp_path, src=SRC):
    exe = str(tmp_path / (os.path.basename(src)[:-2] + "_c"))
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-ldl", "-lm", "-o", exe], check=True)
    return exe


def _lib():
    import cmblensing_jl_amd as C
    return C.library_path()


def test_c_caller_builds_and_fails_loudly_without_a_device(tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: covered by the gpu test")
    r = subprocess.run([exe, _lib(), GOLD], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and "C_ABI_PASS" not in r.stdout
    r = subprocess.run([_build(tmp_path, SRC_POST), _lib(), GOLD_POST], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and "C_ABI_PASS" not in r.stdout


def test_cabi_golden_is_the_oracle():
    """drift guard: the committed binary equals what the oracle computes today"""
    import struct
    import oracle as O
    from oracle.lenseflow import LenseFlow
    raw = open(GOLD, "rb").read()
    Ny, Nx, P, n, theta = struct.unpack("<iiiid", raw[:24])
    a = np.frombuffer(raw[24:], np.float64)
    nmap, nf = Ny * Nx, (Ny // 2 + 1) * Nx * 2
    phi, f, delta, Lf = np.split(a[:nmap + 2 * P * nmap + P * nf], [nmap, nmap + P * nmap, nmap + P * nmap + P * nf])
    proj = O.Proj(Ny, Nx, theta, np.float64)
    L = LenseFlow(proj, phi.reshape(1, 1, Nx, Ny), n)
    np.testing.assert_allclose(L.apply(f.reshape(1, P, Nx, Ny)).ravel(), Lf, rtol=0, atol=1e-12 * np.abs(Lf).max())


@pytest.mark.gpu
def test_c_caller_matches_oracle_on_the_device(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, _lib(), GOLD], capture_output=True, text=True)
    assert r.returncode == 0 and "C_ABI_PASS" in r.stdout, r.stdout + r.stderr


def test_cabi_posterior_golden_is_the_oracle():
    """drift guard for tests/golden/cabi_posterior.bin (tools/make_cabi_posterior_golden.py): the expected logpdf and gradients are what
    the oracle computes today from the stored f°, ϕ°, d"""
    import struct
    import oracle as O
    raw = open(GOLD_POST, "rb").read()
    Ny, Nx, P, n, theta, logdet_sum = struct.unpack("<iiiidd", raw[:32])
    a = np.frombuffer(raw[32:], np.float64)
    Nyh = Ny // 2 + 1
    npl, nmap = Nyh * Nx, Ny * Nx
    off = (7 * P + 2) * npl + nmap                                     # the ten operators
    d, fo, po, lp, gfo, gpo = np.split(a[off:], np.cumsum([P * npl * 2, P * nmap, npl * 2, 1, P * nmap]))
    s = O.load_sim(theta, (Ny, Nx), "P", np.float64, beam_fwhm=3.0, pixel_mask=dict(pad_deg=0.3, apod_deg=0.3), nsteps=n)
    ds = s["ds"]
    np.testing.assert_allclose(np.stack(ds.Cf.pinv().arrays()).ravel(), a[:P * npl], rtol=1e-13)
    cplx = lambda v, shp: v.view(np.complex128).reshape(shp)
    ds.d = cplx(d, (1, P, Nx, Nyh))
    lp2, gfo2, gpo2 = ds.grad_logpdf_mixed(fo.reshape(1, P, Nx, Ny), cplx(po, (1, 1, Nx, Nyh)))
    np.testing.assert_allclose(lp2, lp, rtol=1e-12)
    np.testing.assert_allclose(gfo2.ravel(), gfo, rtol=0, atol=1e-11 * np.abs(gfo).max())
    np.testing.assert_allclose(gpo2.ravel().view(np.float64), gpo, rtol=0, atol=1e-11 * np.abs(gpo).max())


@pytest.mark.gpu
def test_c_caller_of_the_posterior_matches_oracle_on_the_device(tmp_path):
    """cmbl_logpdf_mixed / cmbl_grad_logpdf_mixed (the bench's step) from plain C: dataset built operator by operator through the ABI"""
    exe = _build(tmp_path, SRC_POST)
    r = subprocess.run([exe, _lib(), GOLD_POST], capture_output=True, text=True)
    assert r.returncode == 0 and "C_ABI_PASS" in r.stdout, r.stdout + r.stderr
