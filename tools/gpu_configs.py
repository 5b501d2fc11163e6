"""Measure the BASELINE.json configurations and the reference's own six-row benchmark table (test/runbenchmarks.jl:114-121,
N=256, fp32, θpix=3, spin-0 / spin-2) on the device.  Prints a markdown table.   python tools/gpu_configs.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmblensing_jl_amd as C
from bench import synthetic_cls, survey_bytes as algorithmic_bytes      # SURVEY §8(d) pass structure of the REFERENCE: a speed-up figure, not a bandwidth

cls = synthetic_cls()

def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3

def six_rows(N, pol, T, theta, nsteps=7, mask=None):
    s = C.load_sim(theta, N, pol, cls, T=T, pixel_mask=mask, nsteps=nsteps)
    ds, f, phi = s["ds"], s["f"], s["phi"]
    fm = f.to(C.MAP); gl = fm.to(C.FOURIER); L = ds.L(phi); ft = L * fm
    fo, po = ds.mix(f, phi)
    norm_grad = lambda: L.gradient(C.FLOW_FWD, L * fm, gl, basis_df=C.FOURIER)      # gradient(ϕ->norm(L(ϕ)*f)) = forward + δ-flow
    return dict(cache=timeit(lambda: (L.invalidate(), L(phi))), L=timeit(lambda: L * fm), Ladj=timeit(lambda: L.adjoint * gl),
                gradL=timeit(norm_grad), lnP=timeit(lambda: ds.logpdf_mixed(fo, po)), gradlnP=timeit(lambda: ds.gradient_logpdf_mixed(fo, po), 3)), s

print("### reference table (test/runbenchmarks.jl:132-145), N=256 fp32 θpix=3′ — reference CPU ms vs MI355X ms")
ref = {"I": (25, 13, 13, 85, 65, 240), "P": (25, 30, 30, 140, 110, 380)}
print("| op | spin-0 ref | spin-0 MI355X | spin-2 ref | spin-2 MI355X |\n|---|---|---|---|---|")
rows = {}
for pol in ("I", "P"):
    rows[pol], _ = six_rows(256, pol, torch.float32, 3.0)
for i, k in enumerate(("cache", "L", "Ladj", "gradL", "lnP", "gradlnP")):
    print(f"| {k} | {ref['I'][i]} | {rows['I'][k]:.3f} | {ref['P'][i]} | {rows['P'][k]:.3f} |")

print("\n### BASELINE.json configs")
# config 2: 512² QU fp32: fwd + adjoint + one Wiener CG
r, s = six_rows(512, "P", torch.float32, 2.0, mask=dict(pad_deg=1.0, apod_deg=1.0))
t = time.time(); fw, h = s["ds"].argmaxf_logpdf(s["phi"]); torch.cuda.synchronize(); dt = time.time() - t
ab = algorithmic_bytes(512, 2, 1, 1, 7, 4)
print(f"config 2 (512² QU fp32): L*f {r['L']:.3f} ms ({ab['lenseflow']/r['L']/1e6:.0f} GB/s alg.), L'g {r['Ladj']:.3f} ms, Wiener CG {len(h)} its in {dt*1e3:.1f} ms ({dt/len(h)*1e3:.2f} ms/it), ∇lnP {r['gradlnP']:.3f} ms")
# config 3: 1024² IQU fp32 MAP_joint gradient step
s3 = C.load_sim(2.0, 1024, "IP", cls, T=torch.float32, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0))
ds3 = s3["ds"]
ds3.host["Nphi"] = C.quadratic_estimate(ds3, "EB")["Nphi"] / 2
fo, po = ds3.mix(s3["f"], s3["phi"])
t_g = timeit(lambda: ds3.gradient_logpdf_mixed(fo, po), 10, 3)
ab = algorithmic_bytes(1024, 3, 1, 1, 7, 4)
t = time.time(); st = C.MAP_joint_step(ds3, C.Field(s3["proj"], torch.zeros_like(s3["phi"].arr), C.FOURIER), cg_nsteps=100); torch.cuda.synchronize(); dt = time.time() - t
print(f"config 3 (1024² IQU fp32): ∇lnP {t_g:.3f} ms ({ab['grad_lnP']/t_g/1e6:.0f} GB/s alg.); one MAP_joint step (CG {len(st['cg_hist'])} its, line search {st['linesearch_evals']} evals): {dt*1e3:.0f} ms")
# config 5: 2048² QU fp64 n=10 L*f + quadratic_estimate(EB)
s5 = C.load_sim(2.0, 2048, "P", cls, T=torch.float64, nsteps=10)
ds5 = s5["ds"]; L5 = ds5.L(s5["phi"]); fm5 = s5["f"].to(C.MAP)
t_l = timeit(lambda: L5 * fm5, 3)
ab = algorithmic_bytes(2048, 2, 1, 1, 10, 8)
t = time.time(); qe = C.quadratic_estimate(ds5, "EB"); torch.cuda.synchronize(); dt = time.time() - t
print(f"config 5 (2048² QU fp64 n=10): L*f {t_l:.3f} ms ({ab['lenseflow']/t_l/1e6:.0f} GB/s alg., {ab['lenseflow']/1e9:.1f} GB); quadratic_estimate(EB) {dt*1e3:.0f} ms")
