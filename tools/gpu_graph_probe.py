"""Would replaying a captured flow beat enqueueing its launches one by one?  Eager vs torch.cuda.CUDAGraph replay of L*f, L'g and
(∇L)† with the context on a non-default stream: python tools/gpu_graph_probe.py [N ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024]
s = torch.cuda.Stream()
for N in sizes:
    with torch.cuda.stream(s):
        sim = C.load_sim(2.0 if N > 256 else 3.0, N, "P", synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=0.5, apod_deg=0.5))
        ds, f, phi = sim["ds"], sim["f"], sim["phi"]
        fm = f.to(C.MAP); L = ds.L(phi); gl = fm.to(C.FOURIER); ft = L * fm
        ops = {"L*f": lambda: L * fm, "L'g": lambda: L.adjoint * gl, "gradL": lambda: L.gradient(C.FLOW_FWD, ft, gl)}
        for name, fn in ops.items():
            for _ in range(3): fn()
            s.synchronize(); t = time.time()
            for _ in range(20): fn()
            s.synchronize(); eager = (time.time() - t) / 20 * 1e3
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    out = fn()
                for _ in range(3): g.replay()
                s.synchronize(); t = time.time()
                for _ in range(20): g.replay()
                s.synchronize(); rep = (time.time() - t) / 20 * 1e3
                print("N %4d %-6s eager %.3f ms   graph replay %.3f ms" % (N, name, eager, rep), flush=True)
            except Exception as e:
                print("N %4d %-6s eager %.3f ms   capture failed: %s" % (N, name, eager, str(e)[:200]), flush=True)
                torch.cuda.synchronize()
