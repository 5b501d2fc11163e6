#!/usr/bin/env python3
"""tests/golden/config2_cg.json: the float64 oracle's Wiener-filter CG on BASELINE config 2 (512² QU, θpix 2′, 1° mask, tol 1e-1):
iteration count and the first residuals.  The oracle needs minutes for this solve (≈110 flows at 512²), too long for the GPU test
run, so its answer is committed as data; tests/test_gpu_configs.py compares the device solve with it and runs the oracle itself only
for the two flows.   python tools/make_config2_golden.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O

t0 = time.time()
so = O.load_sim(2.0, (512, 512), "P", np.float64, beam_fwhm=0.0, pixel_mask=dict(pad_deg=0.4, apod_deg=0.4))
fw, hist = so["ds"].argmaxf_logpdf(so["phi"], tol=1e-1, nsteps=500)
out = dict(config="512x512 QU, theta_pix 2, beam 0, mask pad 0.4 apod 0.4 deg, tol 1e-1, float64 oracle", ncg=len(hist),
           res=[float(h[1][0]) for h in hist], f_l2=float(np.sqrt(np.sum(np.abs(fw) ** 2))), seconds=time.time() - t0)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config2_cg.json")
json.dump(out, open(path, "w"), indent=1)
print(path, out["ncg"], "iterations", f"{out['seconds']:.0f} s")
