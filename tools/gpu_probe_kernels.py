"""Wall clock (timed mode) of ∇lnP and a CG iteration at N² QU / T+QU plus the per-launch kernel times (one launch over all pol
slices) -- the A/B probe for experiment builds: CMBL_LIB=... python tools/gpu_probe_kernels.py [N=1024] [f32|f64] [pols=P,IP]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
DT = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else torch.float32
for pol in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("P", "IP")):
    s = C.load_sim(2.0, N, pol, synthetic_cls(), T=DT, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
    ds, proj, f, phi = s["ds"], s["proj"], s["f"], s["phi"]
    fo, po = ds.mix(f, phi)
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
    g = min(timeit(lambda: ds.gradient_logpdf_mixed(fo, po)) for _ in range(3))
    ds.argmaxf_logpdf(phi, tol=0.0, nsteps=10)
    cg = min(timeit(lambda: ds.argmaxf_logpdf(phi, tol=0.0, nsteps=40), n=2) / 40 for _ in range(3))
    old = proj.set_option("slice_streams", 1)
    proj.prof_reset(); proj.prof_enable(True)
    for _ in range(5): ds.gradient_logpdf_mixed(fo, po)
    ds.argmaxf_logpdf(phi, tol=0.0, nsteps=20)
    proj.prof_enable(False)
    proj.set_option("slice_streams", old)
    tab = proj.prof_table()
    lp, gf, gp = ds.gradient_logpdf_mixed(fo, po)
    print(pol, "gradlnP %.3f ms  cg %.4f ms | lp %.6f |" % (g, cg, lp[0]),
          {k: round(v[0] / v[1] * 1e3, 2) for k, v in tab.items() if k in ("delta_rows", "adj_x", "x_grad", "delta_cols", "flow_y_fwd", "adj_y", "dphi_reduce")})
