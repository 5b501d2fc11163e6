"""One-off sweep of the compile-time-plan list: random (Ny, Nx) pairs of CMBL_CT_LIST, P in 1..3, B in 1..2, both precisions --
   compile-time against run-time plans (to rounding) and tiled against [ky][x] hand-off arrays (bit for bit) on L*f, L'g and the delta-flow gradient.
   python tools/gpu_ct_sweep.py [npairs=24] [seed=0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
CT = (96, 160, 192, 320, 360, 384, 480, 576, 640, 720, 768, 960, 1000, 1152, 1280, 1536, 1920, 2304, 2560, 3072)
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(npairs):
    while True:
        Ny, Nx = (int(v) for v in rng.choice(CT, 2))
        if Ny * Nx <= 1536 * 1536: break
    P, B = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    for T, nT, cT, tol in ((torch.float32, np.float32, np.complex64, 1e-3), (torch.float64, np.float64, np.complex128, 1e-11)):
        pol = {1: "I", 2: "P", 3: "IP"}[P]
        sim = C.load_sim(2.0, (Ny, Nx), pol, synthetic_cls(), T=T, pixel_mask=dict(pad_deg=0.2, apod_deg=0.2), nsteps=7, Nbatch=B, seeds=(it + 1, it + 2, it + 3), Nphi="flat")   # physical spectra: white
        ds, p, f, phi = sim["ds"], sim["proj"], sim["f"], sim["phi"]                 # noise at the pixel scale makes the flow ill-conditioned (errors x 1e6)
        fm = f.to(C.MAP); gf = fm.to(C.FOURIER)
        res = {}
        for name, opts in (("rt", dict(gen_ct=0)), ("ct", dict(gen_ct=1, gen_tiled=0)), ("tiled", dict(gen_ct=1, gen_tiled=7))):
            for k, v in opts.items(): p.set_option(k, v)
            L = ds.L(phi)
            ft = L * fm
            dphi, df, f0 = L.gradient(C.FLOW_FWD, ft, gf)
            res[name] = [ft.arr.clone(), (L.adjoint * gf).arr.clone(), dphi.arr.clone(), df.arr.clone()]
        rel = lambda a, b: float((a - b).abs().pow(2).sum().sqrt() / b.abs().pow(2).sum().sqrt())
        e = max(rel(a, b) for a, b in zip(res["ct"], res["rt"]))
        same = all(torch.equal(a, b) for a, b in zip(res["tiled"], res["ct"]))
        ok = e < tol and same
        bad += not ok
        print(f"{Ny:5d} x {Nx:5d} P={P} B={B} {str(T)[6:]:8s} ct vs rt {e:.2e}  tiled == [ky][x]: {same}  {'ok' if ok else 'FAILED'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
