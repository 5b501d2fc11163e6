"""Phase timestamps inside k_delta_cols (debug build: tools/devbuild.sh stamps -DCMBL_STAMPS):
   CMBL_SLICE_STREAMS=1 CMBL_LIB=cmblensing.jl_amd/_dev/lib_stamps.so python tools/gpu_stamps.py
(slice streams off: concurrent half-grid launches would write the same stamp slots)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CMBL_SLICE_STREAMS", "1")
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
N = int(os.environ.get("N", 1024))                          # N=2048 DT=f64 NRK=10 NB=2048: BASELINE config 5
T = torch.float64 if os.environ.get("DT", "f32") == "f64" else torch.float32
s = C.load_sim(2.0, N, "P", synthetic_cls(), T=T, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), Nphi="flat", nsteps=int(os.environ.get("NRK", 7)))
ds, f, phi = s["ds"], s["f"], s["phi"]
fm = f.to(C.MAP); L = ds.L(phi); gl = fm.to(C.FOURIER); ft = L * fm
for _ in range(3):
    L.gradient(C.FLOW_FWD, ft, gl)
torch.cuda.synchronize()
lib = C.load_library()
nb = int(os.environ.get('NB', 512))
buf = (ctypes.c_ulonglong * (nb * 16))()
lib.cmbl_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cmbl_debug_stamps(buf, nb * 16) == 0
st = np.array(buf[:], dtype=np.uint64).reshape(nb, 16).astype(np.int64)
rel = st[:, :13] - st[:, :1]
names = ["start", "loads committed (pair tile)", "N-pt inverse + read (dx,dy)", "barrier", "commit H tile", "M-pt inverse + read (L df)",
         "pointwise + RK + stores", "-", "N-pt write + forward", "pair split + Wx,Wy stores", "-", "M-pt write + forward", "Anext store"]
prev = 0
for i in range(13):
    m = rel[:, i]
    if (m < 0).any():
        continue
    print(f"{i:2d} {names[i]:34s} t = {m.mean():8.0f}  (+{m.mean() - prev:7.0f})   min {m.min():7d} max {m.max():7d}")
    prev = m.mean()
w = st[:, 14:16]
t0 = w[:, 0].min()
print("wall clock (10 ns ticks): block starts after first start: p50 %.2f us p99 %.2f us max %.2f us; block durations mean %.2f us max %.2f us; launch span %.2f us"
      % (np.percentile(w[:, 0] - t0, 50) / 100, np.percentile(w[:, 0] - t0, 99) / 100, (w[:, 0] - t0).max() / 100,
         (w[:, 1] - w[:, 0]).mean() / 100, (w[:, 1] - w[:, 0]).max() / 100, (w[:, 1].max() - t0) / 100))

# where the slow workgroups are: duration by XCD (linear workgroup id % 8), by slice, by position in the grid
dur = (w[:, 1] - w[:, 0]) / 100.0
gx = nb // 2
lin = np.arange(nb)
print("duration [us] by XCD (linear id % 8):   " + " ".join("%.2f" % dur[lin % 8 == k].mean() for k in range(8)))
print("duration [us] by slice (blockIdx.y):    " + " ".join("%.2f" % dur[lin // gx == k].mean() for k in range(2)))
print("duration [us] by blockIdx.x octile:     " + " ".join("%.2f" % dur[(lin % gx) * 8 // gx == k].mean() for k in range(8)))
srt = np.argsort(-dur)[:12]
print("slowest: " + ", ".join("wg %d (x %d, y %d) %.2f" % (i, i % gx, i // gx, dur[i]) for i in srt))
print("first-commit wait [cycles] of the 32 slowest vs all: %.0f vs %.0f; transform phases (stamps 1->9): %.0f vs %.0f"
      % (rel[np.argsort(-dur)[:32], 1].mean(), rel[:, 1].mean(), (rel[np.argsort(-dur)[:32], 9] - rel[np.argsort(-dur)[:32], 1]).mean(), (rel[:, 9] - rel[:, 1]).mean()))
