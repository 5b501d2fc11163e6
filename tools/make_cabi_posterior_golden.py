#!/usr/bin/env python3
"""tests/golden/cabi_posterior.bin: operators, data, inputs and float64-oracle outputs for the plain-C caller
tests/c_abi/posterior.c (cmbl_dataset_* -> cmbl_logpdf_mixed / cmbl_grad_logpdf_mixed, the call bench.py times).
Layout (little endian): int32 Ny, Nx, P, nsteps; float64 theta_pix, logdet_sum; then float64 arrays in the C ABI's layouts:
  7 operators of P real (Nyh x Nx) planes each, in CMBL_OP_* order (CF_INV, CN_INV, B, MF, D, D_INV, PRECOND_INV),
  CPHI_INV, G_INV (one plane each), MPIX (Ny x Nx map), d (P complex half-planes, harmonic basis), f° (P maps), ϕ° (1 complex
  half-plane); expected: logpdf (1 value), ∇f° (P maps), ∇ϕ° (1 complex half-plane)."""
import os, struct, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O

Nside, theta, n = (32, 64), 3.0, 7
s = O.load_sim(theta, Nside, "P", np.float64, beam_fwhm=3.0, pixel_mask=dict(pad_deg=0.3, apod_deg=0.3), nsteps=n)
ds, proj = s["ds"], s["proj"]
P = ds.P
fo, po = ds.mix(s["f"], s["phi"])
lp, gfo, gpo = ds.grad_logpdf_mixed(fo, po, alias_quirk=False)
assert abs(lp[0] - ds.logpdf_mixed(fo, po)[0]) < 1e-9 * abs(lp[0])
planes = lambda op: np.stack(op.arrays())
ops = [planes(ds.Cf.pinv()), planes(ds.Cn.pinv()), planes(ds.B), planes(ds.Mf), planes(ds.D), planes(ds.D.pinv()), planes(ds.precond_f().pinv()),
       O.pinv(ds.Cphi)[None], O.pinv(ds.G)[None], ds.Mpix]
logdet_sum = float(ds.Cf.logdet(proj)[0] + ds.Cn.logdet(proj)[0] + O.logdet_fourier(proj, ds.Cphi[None, None])[0])
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cabi_posterior.bin")
with open(out, "wb") as fh:
    fh.write(struct.pack("<iiiidd", Nside[0], Nside[1], P, n, theta, logdet_sum))
    for a in ops + [ds.d, fo, po, lp, gfo, gpo]:
        a = np.ascontiguousarray(a)
        fh.write((a.view(np.float64) if np.iscomplexobj(a) else a.astype(np.float64)).tobytes())
print(out, os.path.getsize(out), "bytes; logpdf", lp)
