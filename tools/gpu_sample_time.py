"""sample_joint at 1024²: wall time per Gibbs step and where the host spends it.  python tools/gpu_sample_time.py [pol] [nchains]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
pol = sys.argv[1] if len(sys.argv) > 1 else "P"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s = C.load_sim(2.0, 1024, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), Nbatch=B, rng="device")
ds = s["ds"]
C.sample_joint(ds, 1, chain_ids=tuple(range(B)), rng="device", phi_start=s["phi"])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.time()
out = C.sample_joint(ds, 3, chain_ids=tuple(range(B)), rng="device", phi_start=s["phi"])
torch.cuda.synchronize(); dt = time.time() - t0; pr.disable()
print(f"{pol} B={B}: {dt / 3 * 1e3:.0f} ms per Gibbs step ({B * 3 / dt:.2f} chain-steps/s), ncg {out['ncg'][:, 0]}, accept {out['accept'].mean():.2f}")
pstats.Stats(pr).sort_stats("tottime").print_stats(7)
