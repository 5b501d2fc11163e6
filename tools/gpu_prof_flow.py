"""Small fixed workload for rocprofv3 counter passes: a few L*f, L'g and gradient flows at the bench size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
s = C.load_sim(2.0, N, "P", synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
ds, f, phi = s["ds"], s["f"], s["phi"]
fm = f.to(C.MAP); gl = fm.to(C.FOURIER); L = ds.L(phi)
for _ in range(2):
    ft = L * fm
    g = L.adjoint * gl
    L.gradient(C.FLOW_FWD, ft, gl)
torch.cuda.synchronize()
