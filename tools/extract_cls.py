#!/usr/bin/env python3
"""Decode the reference's cached CAMB spectra into a small .npz data fixture.

Source (data file, read-only, only in the build container):
    /root/reference/dat/default_camb_Cls.jld2
written by dat/compute_default_camb_Cls.jl:1-2 from `camb(ℓmax=16000)`
(src/cls.jl:135-200): five groups {unlensed_scalar, lensed_scalar, tensor,
unlensed_total, total} x {TT,EE,BB,TE} + one shared ϕϕ, each a Float64 vector
on ℓ = 2 … 15999 (src/cls.jl:180), units μK² (ϕϕ dimensionless).

JLD2 is HDF5-like; rather than parse it we scan for the 21 zlib streams that
hold the 15998-element Float64 vectors (order verified below through the
identities  unlensed_total = unlensed_scalar + tensor,  total = lensed_scalar
+ tensor,  unlensed_scalar.BB == 0).

Output: tests/golden/camb_cls.npz  (ℓ ≤ LMAX_KEEP, float64).
This script is NOT run on the GPU box; the .npz travels instead.
"""
import sys
import zlib
import numpy as np

SRC = "/root/reference/dat/default_camb_Cls.jld2"
DST = "tests/golden/camb_cls.npz"
NL = 15998
LMAX_KEEP = 12000      # 2048² @ 1′ would need ≈15274; 2′ pixels need 7638


def scan_streams(buf):
    out, i = [], 0
    while True:
        j = buf.find(b"\x78", i)
        if j < 0:
            break
        if buf[j + 1] in (0x01, 0x5E, 0x9C, 0xDA):
            try:
                d = zlib.decompressobj()
                raw = d.decompress(buf[j:])
                if len(raw) == NL * 8:
                    out.append(np.frombuffer(raw, dtype="<f8").copy())
                    i = j + (len(buf) - j - len(d.unused_data))
                    continue
            except zlib.error:
                pass
        i = j + 1
    return out


def main():
    buf = open(SRC, "rb").read()
    s = scan_streams(buf)
    assert len(s) == 21, len(s)
    names = (["unlensed_scalar_" + k for k in ("TT", "EE", "BB", "TE")] + ["phiphi"]
             + [g + "_" + k for g in ("lensed_scalar", "tensor", "unlensed_total", "total")
                for k in ("TT", "EE", "BB", "TE")])
    cl = dict(zip(names, s))
    # verify the assumed order through the reference's own identities (src/cls.jl:192-193)
    assert np.all(cl["unlensed_scalar_BB"] == 0)
    for k in ("TT", "EE", "BB", "TE"):
        # holds only where CAMB itself computed the spectra (ℓ' = 2…4999, src/cls.jl:161,181);
        # above that each spectrum is extrapolated independently (src/cls.jl:100-111)
        m = slice(0, 4998)
        sc = np.abs(cl["unlensed_total_" + k][m]).max()
        np.testing.assert_allclose(cl["unlensed_total_" + k][m],
                                   (cl["unlensed_scalar_" + k] + cl["tensor_" + k])[m], rtol=1e-6, atol=1e-9 * sc)
        np.testing.assert_allclose(cl["total_" + k][m],
                                   (cl["lensed_scalar_" + k] + cl["tensor_" + k])[m], rtol=1e-6, atol=1e-9 * sc)
    ell = np.arange(2, 2 + NL)
    assert cl["phiphi"][0] > 0 and np.all(np.diff(cl["phiphi"][:3000] * ell[:3000] ** 4.0) != 0)
    keep = ell <= LMAX_KEEP
    out = {"ell": ell[keep].astype(np.int32)}
    for k in ("unlensed_scalar", "lensed_scalar", "tensor", "unlensed_total", "total"):
        for x in ("TT", "EE", "BB", "TE"):
            out[f"{k}_{x}"] = cl[f"{k}_{x}"][keep]
    out["phiphi"] = cl["phiphi"][keep]
    np.savez_compressed(DST, **out)
    print("wrote", DST, {k: v.shape for k, v in out.items()})
    print("TT[ℓ=2,100,1000]·ℓ(ℓ+1)/2π =",
          [float(cl["total_TT"][l - 2] * l * (l + 1) / 2 / np.pi) for l in (2, 100, 1000)])


if __name__ == "__main__":
    sys.exit(main())
