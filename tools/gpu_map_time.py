"""Wall time of MAP_joint steps and a quadratic estimate at 1024² vs their kernel-level content (host overhead check)"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
pol = sys.argv[1] if len(sys.argv) > 1 else "P"
t0 = time.time()
s = C.load_sim(2.0, 1024, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
torch.cuda.synchronize(); print("load_sim (incl. QE for Nphi) %.2f s" % (time.time() - t0))
ds, p = s["ds"], s["proj"]
phi0 = C.Field(p, torch.zeros_like(s["phi"].arr), C.FOURIER)
st = C.MAP_joint_step(ds, phi0, cg_nsteps=50, cg_tol=0.0)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.time()
st = C.MAP_joint_step(ds, st["phi"], fstart=st["f"], cg_nsteps=50, cg_tol=0.0)
torch.cuda.synchronize(); dt = time.time() - t0; pr.disable()
print("MAP_joint_step (50 CG its, %d line-search evals): %.1f ms" % (st["linesearch_evals"], dt * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
t0 = time.time(); q = C.quadratic_estimate(ds, "EB"); torch.cuda.synchronize(); print("QE(EB) %.1f ms" % ((time.time() - t0) * 1e3))
