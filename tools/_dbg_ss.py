import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import cmblensing_jl_amd as C
from test_gpu_fullsize import _fields
for P in (2, 3):
    proj = C.ProjLambert(1024, 1024, 2.0, torch.float32, 0)
    f, g, phi, _ = _fields(C, proj, P, B=1)
    L = C.LenseFlow(proj, 7); L(phi)
    gl = g.to(C.FOURIER)
    def run():
        a = L * f; b = L.ldiv(f); c = L.adjoint * gl
        dphi, df, fs = L.gradient(C.FLOW_FWD, a, gl)
        torch.cuda.synchronize()
        return [x.arr.clone() for x in (a, b, c, dphi, df, fs)]
    for occ in (0, 3):
        proj.set_option("occupancy_tiles", occ)
        proj.set_option("slice_streams", 4); rs = [run() for _ in range(3)]
        proj.set_option("slice_streams", 1); r1 = run()
        for i, r in enumerate(rs):
            print("P", P, "occ", occ, "rep", i, [("same" if torch.equal(x, y) else "%.2e" % float((x - y).abs().max() / y.abs().max())) for x, y in zip(r, r1)])
