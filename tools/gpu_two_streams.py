"""Does splitting the pol slices of one flow over two streams (two half-size dependent chains interleaving on the GPU) beat one
launch chain over both slices?  python tools/gpu_two_streams.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
N = 1024
rng = np.random.default_rng(0)
phi_np = 1e-4 * rng.standard_normal((1, 1, N, N))
f_np = rng.standard_normal((1, 2, N, N))
def make(stream, P, sl):
    with torch.cuda.stream(stream):
        p = C.ProjLambert(N, N, 2.0, torch.float32, 0)
        L = C.LenseFlow(p, 7)
        phi = C.Field(p, p.tensor(phi_np), C.MAP)
        f = C.Field(p, p.tensor(f_np[:, sl]), C.MAP)
        L(phi)
    return p, L, f
s0 = torch.cuda.current_stream()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
p2, L2, f2 = make(s0, 2, slice(0, 2))
pa, La, fa = make(sa, 1, slice(0, 1))
pb, Lb, fb = make(sb, 1, slice(1, 2))
torch.cuda.synchronize()
def one():
    return L2 * f2
def two():
    with torch.cuda.stream(sa): a = La * fa
    with torch.cuda.stream(sb): b = Lb * fb
    return a, b
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
print("one chain over both pols: %.3f ms" % timeit(one))
print("two chains, one pol each, two streams: %.3f ms" % timeit(two))
with torch.cuda.stream(sa): ta = timeit(lambda: La * fa)
print("single pol alone: %.3f ms" % ta)
a, b = two(); torch.cuda.synchronize(); r = one(); torch.cuda.synchronize()
print("same result:", float((r.arr[:, 0] - a.arr[:, 0]).abs().max()), float((r.arr[:, 1] - b.arr[:, 0]).abs().max()))
