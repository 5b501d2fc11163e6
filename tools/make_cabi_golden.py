#!/usr/bin/env python3
"""tests/golden/cabi_lenseflow.bin: inputs and float64-oracle outputs for the plain-C caller tests/c_abi/lenseflow.c.
Layout (little endian): int32 Ny, Nx, P, nsteps; float64 theta_pix; then float64 arrays in the C ABI's layouts:
phi map (Ny*Nx), f map (P*Ny*Nx), delta Fourier (P*Nyh*Nx complex), L*f map, dphi Fourier (Nyh*Nx complex), df Fourier."""
import os, struct, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
from oracle.lenseflow import LenseFlow

Ny, Nx, P, n, theta = 32, 64, 2, 7, 2.0
proj = O.Proj(Ny, Nx, theta, np.float64)
cl = O.load_camb()["unlensed_total"]
C = np.stack([O.cl_to_2d(cl["EE"], proj), O.cl_to_2d(cl["BB"], proj) + 0.05 * O.cl_to_2d(cl["EE"], proj)])
f = O.from_harm(proj, np.sqrt(C) * O.rfft2(O.white_noise(1, (1, P, Nx, Ny), np.float64)))
g = O.from_harm(proj, np.sqrt(C) * O.rfft2(O.white_noise(11, (1, P, Nx, Ny), np.float64)))
phi = O.irfft2(np.sqrt(O.cl_to_2d(cl["pp"], proj)) * O.rfft2(O.white_noise(2, (1, 1, Nx, Ny), np.float64)), Ny)
L = LenseFlow(proj, phi, n)
Lf = L.apply(f)
delta = O.rfft2(g)
_, df, dphi = L.grad_apply(Lf, delta, alias_quirk=False)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cabi_lenseflow.bin")
with open(out, "wb") as fh:
    fh.write(struct.pack("<iiiid", Ny, Nx, P, n, theta))
    for a in (phi, f, delta, Lf, dphi, df):
        a = np.ascontiguousarray(a)
        fh.write((a.view(np.float64) if np.iscomplexobj(a) else a.astype(np.float64)).tobytes())
print(out, os.path.getsize(out), "bytes")
