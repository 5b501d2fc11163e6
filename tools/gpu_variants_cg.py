"""A/B of experiment builds on the driver-level hot loop: ms per Wiener-CG iteration (QU and T+QU, 1024² fp32), L*f, L'g, ∇lnP.
   python tools/gpu_variants_cg.py lib1.so lib2.so ...      (ROUNDS=3: variants interleaved, minimum over rounds)"""
import os, subprocess, sys, re
code = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
out = []
for pol in ("P", "IP"):
    s = C.load_sim(2.0, 1024, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
    ds, f, phi = s["ds"], s["f"], s["phi"]
    fm = f.to(C.MAP); L = ds.L(phi); gl = fm.to(C.FOURIER)
    fo, po = ds.mix(f, phi)
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
    ds.argmaxf_logpdf(phi, tol=0.0, nsteps=10)
    cg = min(timeit(lambda: ds.argmaxf_logpdf(phi, tol=0.0, nsteps=40), n=2) / 40 for _ in range(2))
    out += [cg, timeit(lambda: L * fm), timeit(lambda: L.adjoint * gl), timeit(lambda: ds.gradient_logpdf_mixed(fo, po))]
print("QU: cg %.4f L*f %.4f L'g %.4f gradlnP %.4f | IQU: cg %.4f L*f %.4f L'g %.4f gradlnP %.4f" % tuple(out))
'''
rounds = int(os.environ.get("ROUNDS", "3"))
best = {}
for r in range(rounds):
    for lib in sys.argv[1:]:
        env = dict(os.environ, CMBL_LIB=os.path.abspath(lib))
        o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = (o.stdout.strip().splitlines() or [o.stderr[-600:]])[-1]
        nums = [float(x) for x in re.findall(r"[0-9]+\.[0-9]+", line)]
        best[lib] = nums if lib not in best else [min(a, b) for a, b in zip(best[lib], nums)]
        print(r, os.path.basename(lib), line, flush=True)
for lib, n in best.items():
    print("MIN %-16s" % os.path.basename(lib), "QU: cg %.4f L*f %.4f L'g %.4f gradlnP %.4f | IQU: cg %.4f L*f %.4f L'g %.4f gradlnP %.4f" % tuple(n), flush=True)
