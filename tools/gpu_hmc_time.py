"""Wall time of one HMC step (25 leapfrog steps, src/sampling.jl:405-418) and one Gibbs step at 1024² vs 26 gradient evaluations"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
pol = sys.argv[1] if len(sys.argv) > 1 else "P"
s = C.load_sim(2.0, 1024, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
ds, p = s["ds"], s["proj"]
fo, po = ds.mix(s["f"], s["phi"])
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
g = t(lambda: ds.gradient_logpdf_mixed(fo, po), 10)
wp = p.randn([1], 0, 1)
h = t(lambda: C.hmc_step(ds, fo, po, wp, np.array([0.0]), N=25, eps=0.01))
print(f"∇lnP {g:.2f} ms; hmc_step {h:.1f} ms = {h / g:.1f} gradient evaluations (25 leapfrog steps need 26 + 2 logpdf)")
P = ds.P
wf, wn = p.randn([2], 0, P), p.randn([3], 0, P)
gs = t(lambda: C.gibbs_step(ds, s["phi"], wf, wn, wp, np.array([0.0])), 2)
print(f"gibbs_step {gs:.0f} ms")
