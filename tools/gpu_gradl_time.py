"""Wall time of one flow operator alone, per RK stage (for rocprofv3 --kernel-trace --stats runs and timing-only probe builds, whose results
must not reach the reductions of a posterior gradient):
   CMBL_LIB=... python tools/gpu_gradl_time.py [N=768] [pol=P] [op=gradL|Lf|Ltg] [option=value ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
P = {"I": 1, "P": 2, "IP": 3}[sys.argv[2] if len(sys.argv) > 2 else "P"]
op = sys.argv[3] if len(sys.argv) > 3 else "gradL"
p = C.ProjLambert(N, N, 2.0, torch.float32)
for kv in sys.argv[4:]:
    k, v = kv.split("="); p.set_option(k, int(v))
rng = np.random.default_rng(1)
F = lambda a, b: C.Field(p, p.tensor(a), b)
phi = F(1e-5 * rng.standard_normal((1, 1, N, N)), C.MAP)
f = F(rng.standard_normal((1, P, N, N)), C.MAP)
L = C.LenseFlow(p, 7)(phi)
g = f.to(C.FOURIER)
fn = {"gradL": lambda: L.gradient(C.FLOW_FWD, f, g), "Lf": lambda: L * f, "Ltg": lambda: L.adjoint * g}[op]
best = 1e9
for r in range(5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 20 * 1e3)
print(f"{op} {N} P={P} {' '.join(sys.argv[4:])}: {best:.4f} ms  = {best * 1e3 / 28:.2f} us per stage")
