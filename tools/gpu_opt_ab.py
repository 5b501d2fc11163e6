"""A/B of one context option on one workload, interleaved, with a bit-identity check of the results:
   python tools/gpu_opt_ab.py <option> <v0,v1,..> [N] [pol] [f32|f64] [nrk]      (CMBL_LIB=... for a dev build)
e.g. python tools/gpu_opt_ab.py col_pipeline 0,1 2048 P f64 10"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls

opt, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
N = (tuple(int(v) for v in sys.argv[3].split("x")) if "x" in sys.argv[3] else int(sys.argv[3])) if len(sys.argv) > 3 else 1024     # N or NyxNx
pol = sys.argv[4] if len(sys.argv) > 4 else "P"
T = torch.float64 if (len(sys.argv) > 5 and sys.argv[5] == "f64") else torch.float32
nrk = int(sys.argv[6]) if len(sys.argv) > 6 else 7
NB = int(os.environ.get("NBATCH", 1))                                    # batch slots (NBATCH=64: small maps filling the chip)
s = C.load_sim(2.0, N, pol, synthetic_cls(), T=T, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0) if np.min(N) >= 256 else dict(pad_deg=0.2, apod_deg=0.2), nsteps=nrk, Nbatch=NB)
ds, p, f, phi = s["ds"], s["proj"], s["f"], s["phi"]
fm = f.to(C.MAP); gl = fm.to(C.FOURIER)
fo, po = ds.mix(f, phi)
nt = int(os.environ.get("NT", 10))

def timeit(fn, n=nt):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def run(v):
    p.set_option(opt, v)
    L = ds.L(phi)
    ft = L * fm
    out = dict(Lf=ft.arr.clone(), Ltg=(L.adjoint * gl).arr.clone())
    dphi, df, _ = L.gradient(C.FLOW_FWD, ft, gl)
    out["dphi"], out["df"] = dphi.arr.clone(), df.arr.clone()
    lp, gf, gp = ds.gradient_logpdf_mixed(fo, po)
    out["gf"], out["gp"], out["lp"] = gf.arr.clone(), gp.arr.clone(), torch.tensor(np.asarray(lp))
    t = [timeit(lambda: L * fm), timeit(lambda: L.adjoint * gl), timeit(lambda: L.gradient(C.FLOW_FWD, ft, gl)), timeit(lambda: ds.gradient_logpdf_mixed(fo, po))]
    return out, t

print(f"N {N} pol {pol} {T} nrk {nrk} option {opt}")
ref, best = None, {}
for r in range(int(os.environ.get("ROUNDS", 3))):
    for v in vals:
        out, t = run(v)
        if ref is None: ref = out
        same = {k: bool(torch.equal(out[k], ref[k])) for k in ref}
        diff = {k: float((out[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ref if not same[k]}
        best[v] = t if v not in best else [min(a, b) for a, b in zip(best[v], t)]
        print(f"round {r} {opt}={v}: L*f {t[0]:.3f}  L'g {t[1]:.3f}  gradL {t[2]:.3f}  gradlnP {t[3]:.3f} ms   bit-identical to first: {all(same.values())} {diff}", flush=True)
for v, t in best.items():
    print(f"MIN {opt}={v}: L*f {t[0]:.3f}  L'g {t[1]:.3f}  gradL {t[2]:.3f}  gradlnP {t[3]:.3f} ms")
