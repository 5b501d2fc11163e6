export CMBL_PARITY_LOG=$PWD/gpurun_out/r05_parity.log
rm -f $CMBL_PARITY_LOG
python -m pytest tests -m gpu -q > gpurun_out/r05_gputest_1.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r05_gputest_1.log
python - <<'PY'
import time, torch, numpy as np
import cmblensing_jl_amd as C
from bench import synthetic_cls
s = C.load_sim(2.0, 2048, "P", synthetic_cls(), T=torch.float64, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), nsteps=10)
ds = s["ds"]
for name, fn in (("native", C.quadratic_estimate_native), ("python", C.quadratic_estimate)):
    fn(ds, "EB"); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); fn(ds, "EB"); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("QE EB 2048^2 fp64", name, ["%.1f" % t for t in ts])
PY
