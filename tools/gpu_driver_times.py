"""Driver-level wall times at 1024² fp32 (QU and T+QU): hmc_step, MAP_joint step, quadratic_estimate -- the Python drivers and the
library's own loop bodies (cmbl_hmc_step, cmbl_map_joint_step, cmbl_quadratic_estimate) side by side, against the ∇lnP evaluations
they consist of.   python tools/gpu_driver_times.py > profiles/rNN_driver_times.txt"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls

def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3

for pol in ("P", "IP"):
    s = C.load_sim(2.0, 1024, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
    ds, p = s["ds"], s["proj"]
    fo, po = ds.mix(s["f"], s["phi"])
    g = t(lambda: ds.gradient_logpdf_mixed(fo, po), 10)
    l = t(lambda: ds.logpdf_mixed(fo, po), 10)
    wp = p.randn([1], 0, 1)
    h_py = t(lambda: C.hmc_step(ds, fo, po, wp, np.array([0.0]), N=25, eps=0.01))
    h_c = t(lambda: C.hmc_step_native(ds, fo, po, white_p=wp, log_u=np.array([0.0]), N=25, eps=0.01))
    print(f"{pol}: ∇lnP {g:.2f} ms, lnP {l:.2f} ms; hmc_step (25 leapfrog steps = 26 ∇lnP + 2 lnP = {26 * g + 2 * l:.1f} ms of evaluations): "
          f"Python driver {h_py:.1f} ms, cmbl_hmc_step {h_c:.1f} ms")
    phi0 = C.Field(p, torch.zeros_like(s["phi"].arr), C.FOURIER)
    kw = dict(cg_nsteps=50, cg_tol=0.0)
    st = C.MAP_joint_step(ds, phi0, **kw)
    m_py = t(lambda: C.MAP_joint_step(ds, phi0, **kw), 2)
    m_c = t(lambda: C.MAP_joint_step_native(ds, phi0, **kw), 2)
    cg = t(lambda: ds.argmaxf_logpdf(phi0, tol=0.0, nsteps=50), 2)
    print(f"{pol}: MAP_joint step (50 CG iterations = {cg:.1f} ms, 1 ∇lnP, {st['linesearch_evals']} line-search lnP): Python driver {m_py:.1f} ms, cmbl_map_joint_step {m_c:.1f} ms")
    which = "EB"
    q_py = t(lambda: C.quadratic_estimate(ds, which), 2)
    q_c = t(lambda: C.quadratic_estimate_native(ds, which), 2)
    print(f"{pol}: quadratic_estimate({which}): Python driver {q_py:.1f} ms, cmbl_quadratic_estimate {q_c:.1f} ms")
