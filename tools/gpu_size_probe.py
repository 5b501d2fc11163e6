"""L*f, L'g, gradient against the float64 oracle at a list of shapes: python tools/gpu_size_probe.py f32 32x4096 32x2048 64x1024 ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle as O
from oracle.lenseflow import LenseFlow as OLF
from test_gpu_parity import sims, DT, rel, _pkg
C = _pkg()
camb = O.load_camb()
prec = sys.argv[1]
tT, nT = DT[prec]
for shp in sys.argv[2:]:
    Ny, Nx = map(int, shp.split("x"))
    P = 2
    oproj, simf, simp = sims(camb, Ny, Nx, P, 1)
    f, g, phi = simf(1).astype(nT), simf(5).astype(nT), simp(2, 1).astype(nT)
    OL = OLF(oproj, phi.astype(float), 7)
    p = C.ProjLambert(Ny, Nx, 2.0, tT)
    L = C.LenseFlow(p, 7)(C.Field(p, p.tensor(phi), C.MAP))
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    e1 = rel((L * F(f, C.MAP)).arr.cpu().numpy(), OL.apply(f.astype(float)))
    e2 = rel((L.adjoint * F(g, C.MAP).to(C.FOURIER)).arr.cpu().numpy(), OL.adj(O.rfft2(g.astype(float))))
    e0 = rel(p.rfft(p.tensor(f)).cpu().numpy(), O.rfft2(f.astype(float)))
    print("%s %5dx%-5d rfft %.2e  L*f %.2e  L'g %.2e" % (prec, Ny, Nx, e0, e1, e2), flush=True)
