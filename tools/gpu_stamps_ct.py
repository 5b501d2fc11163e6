"""Phase timestamps inside k_ct_dft (debug build: tools/devbuild.sh stampsct -DCMBL_STAMPS -DCMBL_STAMPS_ROWS -DCMBL_STAMPS_CT):
   CMBL_CT_STAMP_KIND=<kind, +8 for the d/dx pass> CMBL_STAMPS_TU=cty_f32_a [OP=Lf] [NB=blocks] CMBL_LIB=cmblensing.jl_amd/_dev/lib_stampsct.so python tools/gpu_stamps_ct.py [N]
kinds: 0 complex, 1 real, 2 real pair, 3 c2r, 4 pair c2r, 5 / 6 / 7 real with the stage's pointwise work in the fetch"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
s = C.load_sim(2.0, N, "P", synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), Nphi="flat", nsteps=7)
ds, f, phi = s["ds"], s["f"], s["phi"]
fm = f.to(C.MAP); L = ds.L(phi); gl = fm.to(C.FOURIER); ft = L * fm
for _ in range(3):
    if os.environ.get("OP", "gradL") == "Lf": L * fm                      # OP=Lf: the forward flow (its d/dx pass is a k_ct_dftx launch: kind 8)
    else: L.gradient(C.FLOW_FWD, ft, gl)
torch.cuda.synchronize()
lib = C.load_library()
nb = int(os.environ.get("NB", 96))
buf = (ctypes.c_ulonglong * (nb * 16))()
lib.cmbl_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cmbl_debug_stamps(buf, nb * 16) == 0
st = np.array(buf[:], dtype=np.uint64).reshape(nb, 16).astype(np.int64)
rel = st[:, :7] - st[:, :1]
names = ["start", "fetch: loads, values, LDS stores", "barrier", "transform", "(multiply +) second transform", "barrier", "stores"]
prev = 0
for i in range(7):
    m = rel[:, i]
    print(f"{i} {names[i]:36s} t = {m.mean():8.0f}  (+{m.mean() - prev:7.0f})   min {m.min():7d} max {m.max():7d}")
    prev = m.mean()
w = st[:, 14:16]
t0 = w[:, 0].min()
print("wall clock: block starts after the first p50 %.2f us max %.2f us; block durations mean %.2f us max %.2f us; span %.2f us"
      % (np.percentile(w[:, 0] - t0, 50) / 100, (w[:, 0] - t0).max() / 100, (w[:, 1] - w[:, 0]).mean() / 100, (w[:, 1] - w[:, 0]).max() / 100, (w[:, 1].max() - t0) / 100))
