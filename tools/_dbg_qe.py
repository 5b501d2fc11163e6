import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle as O
from test_gpu_parity import _dataset_pair
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
C, so, sd = _dataset_pair("f64", "P", (N, N), theta=2.0, mask=False, beam=1.0)
ods, ds, p = so["ds"], sd["ds"], sd["proj"]
ds.set_data(C.Field(p, p.tensor(so["d"]), C.HARMONIC))
planes = lambda op: {k: op.d[i] for i, k in enumerate(["E", "B"])}
TF = {k: planes(ods.Mf)[k] * planes(ods.B)[k] for k in ("E", "B")}
dd = {k: so["d"][:, i:i + 1] for i, k in enumerate(("E", "B"))}
pq, AL, Nphi = O.quadratic_estimate(so["proj"], "EB", dd, dd, planes(ods.Cf), planes(ods.Cftilde), planes(ods.Cn), ods.Cphi, TF)
got = C.quadratic_estimate_native(ds, "EB")
lm = so["proj"].lmag
e = np.abs(got["AL"] - AL) / np.maximum(np.abs(AL), 1e-300)
idx = np.argsort(e.ravel())[::-1][:8]
for i in idx:
    print("l %.1f  got %.6e want %.6e relerr %.2e  Cphi %.3e" % (lm.ravel()[i], got["AL"].ravel()[i], AL.ravel()[i], e.ravel()[i], ods.Cphi.ravel()[i]))
for lmax in (1000, 3000, 5000, 6000, 1e9):
    m = (ods.Cphi > 0) & (lm < lmax)
    print("l <", lmax, "max rel", e[m].max(), "rel L2", np.linalg.norm((got["AL"] - AL)[m]) / np.linalg.norm(AL[m]))
print("phiqe rel L2", np.linalg.norm(got["phiqe"].arr.cpu().numpy() - pq) / np.linalg.norm(pq))
