"""Slice streams (one launch chain per pol slice) on / off below the built-in size threshold: python tools/gpu_streams_probe.py [N ...]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
for N in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
    for pol in ("P", "IP"):
        s = C.load_sim(2.0, N, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
        ds, proj, f, phi = s["ds"], s["proj"], s["f"], s["phi"]
        fo, po = ds.mix(f, phi)
        def timeit(fn, n=30):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t = time.time()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
        out = []
        for minpix in (1 << 30, 0, 1 << 30, 0):
            proj.set_option("slice_streams_min_pix", minpix)
            g = min(timeit(lambda: ds.gradient_logpdf_mixed(fo, po)) for _ in range(3))
            ds.argmaxf_logpdf(phi, tol=0.0, nsteps=5)
            cg = min(timeit(lambda: ds.argmaxf_logpdf(phi, tol=0.0, nsteps=40), n=2) / 40 for _ in range(3))
            out.append("%s: gradlnP %.3f cg %.4f" % ("streams" if minpix == 0 else "single ", g, cg))
        print(N, pol, " | ".join(out))
