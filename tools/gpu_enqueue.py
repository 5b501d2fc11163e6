import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
s = C.load_sim(2.0, 1024, "P", synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), Nphi="flat")
ds, f, phi = s["ds"], s["f"], s["phi"]
fm = f.to(C.MAP); L = ds.L(phi); gl = fm.to(C.FOURIER); ft = L * fm
for name, fn in (("L*f", lambda: L * fm), ("L'g", lambda: L.adjoint * gl), ("gradL", lambda: L.gradient(C.FLOW_FWD, ft, gl))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    te = tt = 0
    for _ in range(10):
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        te += t1 - t0; tt += t2 - t0
    print(f"{name}: enqueue {te/10*1e3:.3f} ms, total {tt/10*1e3:.3f} ms")
