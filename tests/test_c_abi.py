"""The boundary from a plain-C host: tests/c_abi/lenseflow.c is compiled with gcc against include/cmblens.h alone (C99, no HIP
headers, no HIP link) and dlopen()s the library -- what Julia's `ccall` does.  CPU: it builds, and without a device every entry
point reports CMBL_ERR_HIP instead of crashing.  GPU: ctx -> set_phi -> apply -> grad agree with the float64 oracle vectors of
tests/golden/cabi_lenseflow.bin (tools/make_cabi_golden.py) to 1e-9."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "lenseflow.c")
GOLD = os.path.join(ROOT, "tests", "golden", "cabi_lenseflow.bin")


def _build(tmp_path):
    exe = str(tmp_path / "lenseflow_c")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-ldl", "-lm", "-o", exe], check=True)
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
