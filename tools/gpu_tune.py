"""Sweep the tile knobs (CMBL_TUNE_C / NT / RX) on the bench workload: python tools/gpu_tune.py [N] [pol]"""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pol = sys.argv[2] if len(sys.argv) > 2 else "P"
s = C.load_sim(2.0, N, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
ds, p, f, phi = s["ds"], s["proj"], s["f"], s["phi"]
fm = f.to(C.MAP); gl = fm.to(C.FOURIER)
L = ds.L(phi); ft = L * fm
fo, po = ds.mix(f, phi)
ref = None
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3
for Cc, NT, RX in itertools.product((4, 8, 16), (256, 512, 1024), (1, 2, 4)):
    os.environ.update(CMBL_TUNE_C=str(Cc), CMBL_TUNE_NT=str(NT), CMBL_TUNE_RX=str(RX))
    try:
        t1 = timeit(lambda: L * fm); t2 = timeit(lambda: L.adjoint * gl)
        t3 = timeit(lambda: L.gradient(C.FLOW_FWD, ft, gl)); t4 = timeit(lambda: ds.gradient_logpdf_mixed(fo, po), 3)
        lp = ds.gradient_logpdf_mixed(fo, po)[0][0]
        if ref is None: ref = lp
        print(f"C={Cc:2d} NT={NT:4d} RX={RX}  L*f {t1:6.3f}  L'g {t2:6.3f}  gradL {t3:6.3f}  gradlnP {t4:7.3f} ms   lp-ref {lp-ref:+.3e}", flush=True)
    except Exception as e:
        print(f"C={Cc} NT={NT} RX={RX} failed: {str(e)[:100]}", flush=True)
