"""Time ∇lnP / (∇L)† for each experiment build, interleaved: [N=1024 DTYPE=f32 NRK=7 POL=P] python tools/gpu_variants.py lib1.so lib2.so ..."""
import os, subprocess, sys
code = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
s = C.load_sim(2.0, int(os.environ.get("N", 1024)), os.environ.get("POL", "P"), synthetic_cls(), T=torch.float64 if os.environ.get("DTYPE") == "f64" else torch.float32,
               pixel_mask=dict(pad_deg=1.0, apod_deg=1.0) if int(os.environ.get("N", 1024)) >= 256 else dict(pad_deg=0.2, apod_deg=0.2), nsteps=int(os.environ.get("NRK", 7)))
ds, f, phi = s["ds"], s["f"], s["phi"]
fm = f.to(C.MAP); L = ds.L(phi); gl = fm.to(C.FOURIER); ft = L * fm
fo, po = ds.mix(f, phi)
def timeit(fn, n=int(os.environ.get("NT", 20))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
r = [timeit(lambda: L * fm), timeit(lambda: L.adjoint * gl), timeit(lambda: L.gradient(C.FLOW_FWD, ft, gl)), timeit(lambda: ds.gradient_logpdf_mixed(fo, po))]
r2 = [timeit(lambda: L.gradient(C.FLOW_FWD, ft, gl)), timeit(lambda: ds.gradient_logpdf_mixed(fo, po))]
print("L*f %.3f  L'g %.3f  gradL %.3f/%.3f  gradlnP %.3f/%.3f ms" % (r[0], r[1], r[2], r2[0], r[3], r2[1]))
'''
import re
rounds = int(os.environ.get("ROUNDS", "3"))
best = {}
for r in range(rounds):                      # interleave the variants: clocks drift by a few percent between processes
    for lib in sys.argv[1:]:
        env = dict(os.environ, CMBL_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = (out.stdout.strip().splitlines() or [out.stderr[-400:]])[-1]
        nums = [float(x) for x in re.findall(r"[0-9]+\.[0-9]+", line)]
        best[lib] = nums if lib not in best else [min(a, b) for a, b in zip(best[lib], nums)]
        print(r, os.path.basename(lib), line, flush=True)
for lib, n in best.items():
    print("MIN", os.path.basename(lib), "L*f %.3f  L'g %.3f  gradL %.3f/%.3f  gradlnP %.3f/%.3f" % tuple(n), flush=True)
