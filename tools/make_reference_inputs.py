#!/usr/bin/env python3
"""Inputs for julia/make_reference_fixtures.jl: tests/golden/ref_inputs/*.npy (plain NumPy files, one array each, float64 /
complex128, C order with the reference's axes reversed -- NumPy (B, P, Nx, Ny) is Julia's column-major (Ny, Nx, P, B), the same
bytes).  They are the seeded inputs of the existing golden cases (tests/test_golden.py) written out, so that the Julia package
itself can be run on them:

    flow_*      64x128 QU, θpix = 2′, LenseFlow n = 7: ϕ (map), f (QU map), g (QU map; the adjoint / cotangent input)
    post_P_*    64x128 QU, θpix = 3′, beam 3′, LowPass(3000), border mask: mask (map), f (EB Fourier), ϕ (Fourier), d (EB Fourier)
    post_IP_*   64x64 T+QU, same recipe: mask, f (IEB Fourier), ϕ, d

Outputs of the reference for these inputs go to tests/golden/ref_outputs/ (written by the Julia script); tests/
test_reference_fixtures.py compares the oracle and the HIP engine with them when they exist.
Run from the repo root:  python tools/make_reference_inputs.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O                                        # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_inputs")
FLOW = dict(Ny=64, Nx=128, P=2, theta=2.0, nsteps=7)
POST = {"P": dict(Nside=(64, 128), theta=3.0, beam_fwhm=3.0, pixel_mask=dict(pad_deg=0.4, apod_deg=0.4)),
        "IP": dict(Nside=(64, 64), theta=3.0, beam_fwhm=3.0, pixel_mask=dict(pad_deg=0.4, apod_deg=0.4))}


def flow_inputs():
    """the inputs of tests/test_golden.py::flow_inputs(64, 128, 2)"""
    Ny, Nx, P = FLOW["Ny"], FLOW["Nx"], FLOW["P"]
    proj = O.Proj(Ny, Nx, FLOW["theta"], np.float64)
    cl = O.load_camb()["unlensed_total"]
    Cphi = O.cl_to_2d(cl["pp"], proj)
    C = np.stack([O.cl_to_2d(cl["EE"], proj), O.cl_to_2d(cl["BB"], proj) + 0.05 * O.cl_to_2d(cl["EE"], proj)])
    f = O.from_harm(proj, np.sqrt(C) * O.rfft2(O.white_noise(1, (1, P, Nx, Ny), np.float64)))
    g = O.from_harm(proj, np.sqrt(C) * O.rfft2(O.white_noise(11, (1, P, Nx, Ny), np.float64)))
    phi = O.irfft2(np.sqrt(Cphi) * O.rfft2(O.white_noise(2, (1, 1, Nx, Ny), np.float64)), Ny)
    return dict(flow_phi=phi, flow_f=f, flow_g=g)


def posterior_inputs(pol):
    c = POST[pol]
    s = O.load_sim(c["theta"], c["Nside"], pol, np.float64, beam_fwhm=c["beam_fwhm"], pixel_mask=c["pixel_mask"])
    return {f"post_{pol}_mask": s["ds"].Mpix, f"post_{pol}_f": s["f"], f"post_{pol}_phi": s["phi"], f"post_{pol}_d": s["d"]}


def all_inputs():
    out = flow_inputs()
    for pol in POST:
        out.update(posterior_inputs(pol))
    return {k: np.ascontiguousarray(v, dtype=np.complex128 if np.iscomplexobj(v) else np.float64) for k, v in out.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    for k, v in all_inputs().items():
        np.save(os.path.join(OUT, k + ".npy"), v)
        print(f"{k:14s} {v.dtype} {v.shape}")


if __name__ == "__main__":
    main()
