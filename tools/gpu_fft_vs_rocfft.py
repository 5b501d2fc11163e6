"""Basis transform against the vendor library: torch.fft (hipFFT -> rocFFT) vs cmbl_rfft / cmbl_irfft on the same
(B, P, Nx, Ny) maps.  python tools/gpu_fft_vs_rocfft.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e6

for N, P, T in ((1024, 2, torch.float32), (1024, 3, torch.float32), (2048, 2, torch.float64), (512, 2, torch.float32)):
    p = C.ProjLambert(N, N, 2.0, T, 0)
    x = torch.randn(1, P, N, N, dtype=T, device="cuda")
    fl = p.rfft(x)
    ref = torch.fft.rfft2(x)
    err = float((fl - ref).abs().max() / ref.abs().max())
    t_r = timeit(lambda: torch.fft.rfft2(x)); t_c = timeit(lambda: p.rfft(x))
    t_ri = timeit(lambda: torch.fft.irfft2(ref, s=(N, N))); t_ci = timeit(lambda: p.irfft(fl))
    by = x.numel() * x.element_size() * 2 / 1e3      # read map + write half-plane, in KB -> GB/s = KB/us * 1e-3... (KB/us = GB/s)
    print(f"N={N} P={P} {str(T)[6:]}: rfft2 rocFFT {t_r:7.1f} us ({by / t_r:6.0f} GB/s)  ours {t_c:7.1f} us ({by / t_c:6.0f} GB/s) | "
          f"irfft2 rocFFT {t_ri:7.1f} us  ours {t_ci:7.1f} us | max rel diff {err:.1e}")
