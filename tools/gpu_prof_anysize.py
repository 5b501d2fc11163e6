"""Per-kernel-class time of one ∇lnP evaluation on the any-size path: python tools/gpu_prof_anysize.py [N] [pol]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pol = sys.argv[2] if len(sys.argv) > 2 else "P"
s = C.load_sim(2.0, N, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
ds, p = s["ds"], s["proj"]
fo, po = ds.mix(s["f"], s["phi"])
for _ in range(2):
    ds.gradient_logpdf_mixed(fo, po)
p.prof_reset(); p.prof_enable(True)
n = 5
for _ in range(n):
    ds.gradient_logpdf_mixed(fo, po)
p.prof_enable(False)
tab = p.prof_table()
tot = sum(ms for ms, _ in tab.values())
print(f"N={N} {pol}: {tot / n:.3f} ms of kernel time per evaluation")
for k, (ms, nl) in sorted(tab.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:20s} {ms / n:8.3f} ms  {nl / n:7.1f} launches  {1e3 * ms / nl:8.2f} us each")
