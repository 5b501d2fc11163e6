"""N evaluations of grad logpdf(Mixed) for a profiler run: python tools/gpu_step_loop.py [N] [pol] [f32|f64] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmblensing_jl_amd as C
from bench import synthetic_cls

N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
pol = sys.argv[2] if len(sys.argv) > 2 else "P"
T = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
s = C.load_sim(2.0, N, pol, synthetic_cls(), T=T, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
ds, f, phi = s["ds"], s["f"], s["phi"]
fo, po = ds.mix(f, phi)
for _ in range(reps):
    ds.gradient_logpdf_mixed(fo, po)
torch.cuda.synchronize()
