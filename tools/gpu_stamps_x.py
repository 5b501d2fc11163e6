"""Phase timestamps inside the forward flow's row kernel k_x_fft<2> (debug build: tools/devbuild.sh stx -DCMBL_STAMPS -DCMBL_STAMPS_X):
   CMBL_LIB=cmblensing.jl_amd/_dev/lib_stx.so python tools/gpu_stamps_x.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
nb = 1026
buf = (ctypes.c_ulonglong * (nb * 16))()
lib.cmbl_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cmbl_debug_stamps(buf, nb * 16) == 0
st = np.array(buf[:], dtype=np.uint64).reshape(nb, 16).astype(np.int64)
d = np.diff(st[:, :5], axis=1)
for i, n in enumerate(["load row + twiddles + sync", "forward FFT", "inverse FFT (with i*lx)", "store"]):
    print(f"{n:30s} mean {d[:, i].mean():8.0f}  min {d[:, i].min():6d}  max {d[:, i].max():6d} cycles")
print("total", (st[:, 4] - st[:, 0]).mean())
# per-XCD launch span: blocks with the same (id % 8) share a clock; keep the blocks of the last launch only
for x in range(2):
    sel = st[x::8]
    med = np.median(sel[:, 0])
    sel = sel[np.abs(sel[:, 0] - med) < 200000]
    print("XCD", x, len(sel), "blocks; first start -> last end:", sel[:, 4].max() - sel[:, 0].min(), "cycles; start spread", sel[:, 0].max() - sel[:, 0].min(),
          "; starts sorted (first 6, last 6):", np.sort(sel[:, 0] - sel[:, 0].min())[[0, 1, 2, 3, 4, 5, -6, -5, -4, -3, -2, -1]])
