"""Quick timing probe (not the bench): python tools/gpu_time.py [N] [pol]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pol = sys.argv[2] if len(sys.argv) > 2 else "P"
T = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
s = C.load_sim(2.0, N, pol, synthetic_cls(), T=T, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
ds, p, f, phi = s["ds"], s["proj"], s["f"], s["phi"]
fm = f.to(C.MAP)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
L = ds.L(phi)
print("N", N, pol, T)
print("set_phi  %.3f ms" % timeit(lambda: (L.invalidate(), L(phi))))
print("L*f      %.3f ms" % timeit(lambda: L * fm))
print("L\\f      %.3f ms" % timeit(lambda: L.ldiv(fm)))
gl = fm.to(C.FOURIER)
print("L'*g     %.3f ms" % timeit(lambda: L.adjoint * gl))
ft = L * fm
print("(∇L)†    %.3f ms" % timeit(lambda: L.gradient(C.FLOW_FWD, ft, gl)))
fo, po = ds.mix(f, phi)
print("lnP      %.3f ms" % timeit(lambda: ds.logpdf_mixed(fo, po)))
print("∇lnP     %.3f ms" % timeit(lambda: ds.gradient_logpdf_mixed(fo, po)))
t = time.time(); fw, h = ds.argmaxf_logpdf(phi); torch.cuda.synchronize(); dt = time.time() - t
print("wiener CG: %d its, %.1f ms/it" % (len(h), dt / len(h) * 1e3))
