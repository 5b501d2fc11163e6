"""Per-kernel-class time of the Wiener-filter CG at 1024² (library HIP-event timers): python tools/gpu_prof_cg.py [pol]"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
pol = sys.argv[1] if len(sys.argv) > 1 else "P"
s = C.load_sim(2.0, 1024, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), Nphi="flat")
ds, p, phi = s["ds"], s["proj"], s["phi"]
ds.argmaxf_logpdf(phi, tol=0.0, nsteps=5)
torch.cuda.synchronize(); t = time.time()
nit = 40
ds.argmaxf_logpdf(phi, tol=0.0, nsteps=nit)
torch.cuda.synchronize(); dt = time.time() - t
print("CG: %.3f ms / iteration" % (dt / nit * 1e3))
lib = p.lib
lib.cmbl_prof_enable(p._h, 1); lib.cmbl_prof_reset(p._h)
ds.argmaxf_logpdf(phi, tol=0.0, nsteps=nit)
torch.cuda.synchronize()
n = lib.cmbl_prof_count()
tot = 0
rows = []
for i in range(n):
    ms, cnt = ctypes.c_double(), ctypes.c_long()
    lib.cmbl_prof_get(p._h, i, ctypes.byref(ms), ctypes.byref(cnt))
    if cnt.value:
        rows.append((lib.cmbl_prof_name(i).decode(), ms.value / nit, cnt.value / nit)); tot += ms.value / nit
for name, ms, cnt in sorted(rows, key=lambda r: -r[1]):
    print(f"  {name:14s} {ms*1e3:8.1f} us/iteration  {cnt:5.1f} launches  {ms/cnt*1e3:6.1f} us each")
print("  sum %.3f ms" % tot)
