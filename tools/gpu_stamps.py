"""Phase timestamps inside k_delta_y (debug build: tools/devbuild.sh stamps -DCMBL_STAMPS):
   CMBL_LIB=cmblensing.jl_amd/_dev/lib_stamps.so python tools/gpu_stamps.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
s = C.load_sim(2.0, 1024, "P", synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
ds, f, phi = s["ds"], s["f"], s["phi"]
fm = f.to(C.MAP); L = ds.L(phi); gl = fm.to(C.FOURIER); ft = L * fm
for _ in range(3):
    L.gradient(C.FLOW_FWD, ft, gl)
torch.cuda.synchronize()
lib = C.load_library()
nb = 512
buf = (ctypes.c_ulonglong * (nb * 16))()
lib.cmbl_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cmbl_debug_stamps(buf, nb * 16) == 0
st = np.array(buf[:], dtype=np.uint64).reshape(nb, 16).astype(np.int64)
swp = os.environ.get("SWAP", "0") == "1"
for sel, lab in ((slice(0, None, 2), "even tiles (order A)"), (slice(1, None, 2), "odd tiles")) if swp else ((slice(None), "all"),):
    t = st[sel]
    rel = t[:, :13] - t[:, :1]
    print(lab, "stamp times since kernel start (mean cycles):", " ".join(f"{i}:{rel[:, i].mean():.0f}" for i in range(13)))
