"""Phase timestamps inside the forward flow's row kernel k_x_fft<2> (debug build: tools/devbuild.sh stx -DCMBL_STAMPS -DCMBL_STAMPS_X):
   CMBL_LIB=cmblensing.jl_amd/_dev/lib_stx.so python tools/gpu_stamps_x.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CMBL_SLICE_STREAMS", "1")
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
s = C.load_sim(2.0, 1024, "P", synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), Nphi="flat")
ds, f, phi = s["ds"], s["f"], s["phi"]
fm = f.to(C.MAP); L = ds.L(phi)
for _ in range(3):
    out = L * fm
torch.cuda.synchronize()
lib = C.load_library()
nb = int(os.environ.get("NB", 2 * ((513 + 3) // 4)))            # row groups of the last launch (RPW = 4)
buf = (ctypes.c_ulonglong * (nb * 16))()
lib.cmbl_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cmbl_debug_stamps(buf, nb * 16) == 0
st = np.array(buf[:], dtype=np.uint64).reshape(nb, 16).astype(np.int64)
seq = [0, 6, 1, 2, 5, 3, 4]
names = ["loads issued+committed (wave 0)", "barrier after load", "forward FFT", "inverse FFT (wave 0)", "barrier after inverse", "store"]
for (a, b), n in zip(zip(seq[:-1], seq[1:]), names):
    d = st[:, b] - st[:, a]
    print(f"{n:34s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):7.0f} p50 {np.percentile(d, 50):7.0f} p90 {np.percentile(d, 90):7.0f} max {d.max():7d} cycles")
tot = st[:, 4] - st[:, 0]
print("total mean %.0f p50 %.0f p90 %.0f max %d" % (tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), tot.max()))
w = st[:, 14:16]
t0 = w[:, 0].min()
print("wall clock: block starts after first start: p50 %.2f us p99 %.2f us max %.2f us; block durations mean %.2f us p90 %.2f max %.2f us; launch span %.2f us"
      % (np.percentile(w[:, 0] - t0, 50) / 100, np.percentile(w[:, 0] - t0, 99) / 100, (w[:, 0] - t0).max() / 100,
         (w[:, 1] - w[:, 0]).mean() / 100, np.percentile(w[:, 1] - w[:, 0], 90) / 100, (w[:, 1] - w[:, 0]).max() / 100, (w[:, 1].max() - t0) / 100))
print("shader clock estimate: %.2f GHz" % (tot.mean() / ((w[:, 1] - w[:, 0]).mean() * 10)))

if os.environ.get("WAVES"):                                        # -DCMBL_STAMPS_WAVES build: spread over the wavefronts of a workgroup
    nw = 8
    big = (ctypes.c_ulonglong * (4096 * 16 + nb * 16 * 2))()
    assert lib.cmbl_debug_stamps(big, len(big)) == 0
    wv = np.array(big[4096 * 16:], dtype=np.uint64).reshape(nb, 16, 2).astype(np.int64)[:, :nw, :]
    wv = wv[: 2 * (513 // 4)]                                      # full row groups only
    d = wv[:, :, 1] - wv[:, :, 0]
    print("transform chain per wave [cycles]: mean over workgroups by wave index:", " ".join("%d" % x for x in d.mean(axis=0)))
    print("   within a workgroup: fastest wave mean %.0f, slowest wave mean %.0f, wave 0 mean %.0f" % (d.min(axis=1).mean(), d.max(axis=1).mean(), d[:, 0].mean()))
    st0 = wv[:, :, 0] - wv[:, :, 0].min(axis=1, keepdims=True)
    print("   chain start after the first wave's start, by wave index:", " ".join("%d" % x for x in st0.mean(axis=0)))
    en = wv[:, :, 1] - wv[:, :, 0].min(axis=1, keepdims=True)
    print("   chain end after the first wave's start, by wave index:", " ".join("%d" % x for x in en.mean(axis=0)))
