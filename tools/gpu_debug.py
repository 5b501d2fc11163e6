"""Quick device-vs-oracle debug run (not a test): python tools/gpu_debug.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle as O
from oracle.lenseflow import LenseFlow as OLF
import cmblensing_jl_amd as C

def rel(a, b): return float(np.linalg.norm((np.asarray(a) - b).ravel()) / np.linalg.norm(np.asarray(b).ravel()))
camb = O.load_camb()
for tT, nT in ((torch.float64, np.float64), (torch.float32, np.float32)):
    for (Ny, Nx) in ((64, 128), (128, 64), (256, 256), (32, 32)):
        p = C.ProjLambert(Ny, Nx, 2.0, tT)
        rng = np.random.default_rng(0)
        m = rng.standard_normal((2, 2, Nx, Ny)).astype(nT)
        fl = p.rfft(p.tensor(m)); ref = O.rfft2(m.astype(float))
        print(nT.__name__, Ny, Nx, "rfft", rel(fl.cpu().numpy(), ref), "irfft", rel(p.irfft(fl).cpu().numpy(), m))
        junk = rng.standard_normal(ref.shape) + 1j * rng.standard_normal(ref.shape)
        print("   non-hermitian irfft", rel(p.irfft(p.tensor(junk)).cpu().numpy(), O.irfft2(junk, Ny)))
        op = O.Proj(Ny, Nx, 2.0, np.float64)
        Cphi = O.cl_to_2d(camb["unlensed_total"]["pp"], op)
        phi = O.irfft2(np.sqrt(Cphi) * O.rfft2(O.white_noise(2, (1, 1, Nx, Ny), float)), Ny).astype(nT).astype(float)
        f = m.astype(float)
        OL = OLF(op, phi, 7)
        L = C.LenseFlow(p, 7)(C.Field(p, p.tensor(phi), C.MAP))
        t = time.time(); out = (L * C.Field(p, p.tensor(f), C.MAP)).arr.cpu().numpy(); dt = time.time() - t
        print("   L*f", rel(out, OL.apply(f)), "%.3fs" % dt)
        print("   L\\f", rel(L.ldiv(C.Field(p, p.tensor(f), C.MAP)).arr.cpu().numpy(), OL.inv(f)))
        gl = O.rfft2(f)
        print("   L'*g", rel((L.adjoint * C.Field(p, p.tensor(gl), C.FOURIER)).arr.cpu().numpy(), OL.adj(gl)))
        fe = OL.apply(f)
        for q in (False, True):
            f0, df, dp = OL.grad_apply(fe, gl, alias_quirk=q)
            gdp, gdf, gf0 = L.gradient(C.FLOW_FWD, C.Field(p, p.tensor(fe), C.MAP), C.Field(p, p.tensor(gl), C.FOURIER), alias_quirk=q)
            print("   grad quirk", q, "f0", rel(gf0.arr.cpu().numpy(), f0), "df", rel(gdf.arr.cpu().numpy(), df), "dphi", rel(gdp.arr.cpu().numpy(), dp))
