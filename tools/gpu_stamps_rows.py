"""Launch timeline of k_delta_rows (debug build: tools/devbuild.sh strows -DCMBL_STAMPS -DCMBL_STAMPS_ROWS):
   CMBL_SLICE_STREAMS=1 CMBL_LIB=cmblensing.jl_amd/_dev/lib_strows.so python tools/gpu_stamps_rows.py
Per workgroup: start and end on the chip-wide 100 MHz clock -- shows whether all workgroups of the launch are resident at once
(adjoint part + d/dx part: 2 x 258 workgroups at 1024^2 QU against 2 per CU x 256 CUs)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CMBL_SLICE_STREAMS", "1")
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls
s = C.load_sim(2.0, 1024, "P", synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), Nphi="flat")
ds, f, phi = s["ds"], s["f"], s["phi"]
fm = f.to(C.MAP); L = ds.L(phi); gl = fm.to(C.FOURIER); ft = L * fm
for _ in range(3):
    L.gradient(C.FLOW_FWD, ft, gl)
torch.cuda.synchronize()
lib = C.load_library()
nb = int(os.environ.get("NB", 4 * ((513 + 3) // 4)))
buf = (ctypes.c_ulonglong * (nb * 16))()
lib.cmbl_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cmbl_debug_stamps(buf, nb * 16) == 0
st = np.array(buf[:], dtype=np.uint64).reshape(nb, 16).astype(np.int64)
w = st[:, 14:16]
t0 = w[:, 0].min()
start, dur = (w[:, 0] - t0) / 100.0, (w[:, 1] - w[:, 0]) / 100.0
half = nb // 2
print("workgroups %d (adjoint part %d, d/dx part %d); launch span %.2f us" % (nb, half, nb - half, (w[:, 1].max() - t0) / 100.0))
print("start after the first start [us]: p50 %.2f p90 %.2f p99 %.2f max %.2f; workgroups starting later than 2 us: %d"
      % (np.percentile(start, 50), np.percentile(start, 90), np.percentile(start, 99), start.max(), int((start > 2).sum())))
for name, sl in (("adjoint part", slice(0, half)), ("d/dx part", slice(half, nb))):
    print("%-13s duration [us]: mean %.2f p50 %.2f p90 %.2f max %.2f" % (name, dur[sl].mean(), np.percentile(dur[sl], 50), np.percentile(dur[sl], 90), dur[sl].max()))
late = np.argsort(-start)[:8]
print("latest starters: " + ", ".join("wg %d start %.2f dur %.2f" % (i, start[i], dur[i]) for i in late))
