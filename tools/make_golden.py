#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ from the float64 NumPy oracle.

The reference (Julia) cannot be executed in the build image, so these vectors pin the ORACLE's outputs at the
time the oracle passed the reference's own property tests (tests/test_oracle_*.py); they guard against drift of
the oracle and give the GPU tests fixed targets that travel to the GPU box.  Inputs are regenerated from seeds;
only small outputs (or strided samples + checksums of large ones) are stored.
Run from the repo root:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O                                        # noqa: E402
from oracle.lenseflow import LenseFlow                    # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def sample(a, n=4096):
    """deterministic strided sample + moments of a big array"""
    flat = np.asarray(a).ravel()
    idx = np.linspace(0, flat.size - 1, min(n, flat.size)).astype(np.int64)
    return dict(idx=idx, val=flat[idx], l2=np.sqrt(np.sum(np.abs(flat) ** 2)), s=np.sum(flat))


def pack(prefix, d, out):
    for k, v in d.items():
        out[f"{prefix}.{k}"] = v


def flow_case(Ny, Nx, P, theta=2.0):
    camb = O.load_camb()
    proj = O.Proj(Ny, Nx, theta, np.float64)
    cl = camb["unlensed_total"]
    Cphi = O.cl_to_2d(cl["pp"], proj)
    C = (O.cl_to_2d(cl["TT"], proj)[None] if P == 1 else
         np.stack([O.cl_to_2d(cl["EE"], proj), O.cl_to_2d(cl["BB"], proj) + 0.05 * O.cl_to_2d(cl["EE"], proj)]))
    f = O.from_harm(proj, np.sqrt(C) * O.rfft2(O.white_noise(1, (1, P, Nx, Ny), np.float64)))
    g = O.from_harm(proj, np.sqrt(C) * O.rfft2(O.white_noise(11, (1, P, Nx, Ny), np.float64)))
    phi = O.irfft2(np.sqrt(Cphi) * O.rfft2(O.white_noise(2, (1, 1, Nx, Ny), np.float64)), Ny)
    L = LenseFlow(proj, phi, 7)
    out = {}
    Lf = L.apply(f)
    pack("Lf", sample(Lf), out)
    pack("Linvf", sample(L.inv(f)), out)
    gl = O.rfft2(g)
    pack("Ladjg", sample(L.adj(gl)), out)
    pack("Linvadjg", sample(L.invadj(gl)), out)
    for quirk in (False, True):
        f0, df, dp = L.grad_apply(Lf, gl, alias_quirk=quirk)
        pack(f"grad_q{int(quirk)}.df", sample(df), out)
        pack(f"grad_q{int(quirk)}.dphi", sample(dp), out)
    out["adjoint_identity"] = np.array([O.dot_map(f, L.apply(g))[0], O.dot_fourier(proj, L.adj(O.rfft2(f)), gl)[0]])
    return out


def posterior_case(pol, Nside):
    s = O.load_sim(3.0, Nside, pol, np.float64, beam_fwhm=3.0, pixel_mask=dict(pad_deg=0.4, apod_deg=0.4))
    ds = s["ds"]
    fo, po = ds.mix(s["f"], s["phi"])
    out = {"logpdf": ds.logpdf(s["f"], s["phi"]), "logpdf_mixed": ds.logpdf_mixed(fo, po)}
    lp, gf, gp = ds.grad_logpdf_mixed(fo, po)
    pack("grad_fo", sample(gf), out)
    pack("grad_phio", sample(gp), out)
    fw, hist = ds.argmaxf_logpdf(s["phi"], tol=0.0, nsteps=8)
    out["cg_res"] = np.array([h[1][0] for h in hist])
    pack("cg_f", sample(fw), out)
    pack("d", sample(s["d"]), out)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for (Ny, Nx, P) in ((64, 128, 2), (128, 128, 1)):
        np.savez_compressed(os.path.join(OUT, f"flow_{Ny}x{Nx}_P{P}.npz"), **flow_case(Ny, Nx, P))
    for pol, Nside in (("P", (64, 128)), ("IP", (64, 64))):
        np.savez_compressed(os.path.join(OUT, f"posterior_{pol}_{Nside[0]}x{Nside[1]}.npz"), **posterior_case(pol, Nside))
    print(sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
