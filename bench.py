#!/usr/bin/env python3
"""Headline benchmark: LenseFlow + ∇logP steps/sec on a 1024² QU flat-sky map (BASELINE.json `metric`).

One "step" = one evaluation of ∇_(f°,ϕ°) logpdf(Mixed(ds)) (the reference's "∇lnP" row,
test/runbenchmarks.jl:120): precompute(ϕ) + 1 inverse flow + 1 forward flow + 2 δ-flows +
the Fourier-diagonal / mask / reduction work, LenseFlow n = 7 RK4 steps, fp32, inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nside 1024] [--pol P] [--nbatch B] [--config {2,3,5}] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--config C` runs BASELINE.json's configuration C (2: 512² QU fp32; 3: 1024² T+QU fp32; 5: 2048² QU fp64 n=10) with the same step and
adds that configuration's own operations (L*f, L'g, Wiener CG / one MAP_joint step / quadratic_estimate) under `extras`.

roofline: `frac` divides SURVEY.md §8(d)'s ALGORITHMIC bytes of the dominant kernel (its share of the reference's pass structure) by
the kernel's mean launch time and the 8 TB/s peak; `frac_traffic` divides the MEASURED HBM bytes of the same launch (rocprofv3 PMC
passes, profiles/r02_traffic_*.json) instead -- the utilisation figure -- and `frac_traffic_vs_6300` uses the 6.3 TB/s a streaming copy
achieves on this part.  `per_kernel` carries the same three numbers for every flow kernel, `whole_step` for the whole step.

N > 1: one process per GPU, every rank runs its own independent posterior chain state (weak scaling, no
data-path collective); RCCL (`nccl` backend) only gathers the per-chain scalars, as SURVEY.md §8(e) prescribes.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synthetic_cls():
    """Spectra of the reference's own data fixture (decoded dat/default_camb_Cls.jld2 -> tests/golden/camb_cls.npz)."""
    import cmblensing_jl_amd as C
    z = np.load(os.path.join(ROOT, "tests", "golden", "camb_cls.npz"))
    ell = z["ell"]
    out = {}
    for g in ("unlensed_scalar", "tensor", "total"):
        out[g] = {k: C.Cls(ell, z[f"{g}_{k}"]) for k in ("TT", "EE", "BB", "TE")}
        out[g]["pp"] = C.Cls(ell, z["phiphi"])
    return out


def algorithmic_bytes(N, P, B, Bphi, n, s):
    """SURVEY.md §8(d) formulas (unit = one map-pass = N²·s bytes)."""
    mp = N * N * s
    lf = 4 * n * (15 * P * B + 2 * Bphi) * mp
    delta = 4 * n * (30 * P * B + 30 * B + 7 * Bphi) * mp
    pre = (18 + 10 * (2 * n + 1)) * Bphi * mp
    grad = pre + 2 * lf + 2 * delta + 60 * P * B * mp
    return dict(map_pass=mp, lenseflow=lf, delta_flow=delta, precompute=pre, grad_lnP=grad)


# share of SURVEY §8(d)'s per-stage map-passes carried by each of our kernels, per (pol,batch) slice
KERNEL_SHARE = {
    "flow_y_fwd": lambda P, B, Bphi: 10.5 * P * B + 2 * Bphi,      # y-halves of rfft/2 irfft, i·ly multiply, velocity ⊕ RK
    "x_grad": lambda P, B, Bphi: 4.5 * P * B,                      # x-halves of rfft/irfft(∂x), i·lx multiply
    "adj_y": lambda P, B, Bphi: 7.5 * P * B + 2 * Bphi,
    "adj_x": lambda P, B, Bphi: 7.5 * P * B,
    # δ-flow stage = two launches: columns (f part 10.5 + δf part 7.5 per slice) and rows (δf row pass 7.5 + next stage's d/dx pass
    # 4.5 per slice).  The δϕ update is a quadrature over the stages and is formed once per δ-flow (dphi_reduce + 5 rffts): the
    # ≈30·B + 5·Bϕ map-passes per stage that SURVEY §8(d) counts for it are work this design does not do.
    "delta_cols": lambda P, B, Bphi: 18.0 * P * B + 2 * Bphi,
    "delta_rows": lambda P, B, Bphi: 12.0 * P * B,
}


def measured_traffic(N, P, B, dtype):
    """HBM bytes from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate runs, gfx950 correction
    applied: profiles/r02_traffic_*.json, tools/make_traffic_json.py): ({kernel class: bytes per launch}, bytes per step) or
    ({}, None) when no profile matches this workload.  It cannot be measured from inside this process."""
    path = os.path.join(ROOT, "profiles", f"r02_traffic_{N}{'IQU'[3 - P:] if P > 1 else 'I'}_{dtype}.json")
    try:
        z = json.load(open(path))
        w = z["workload"]
        if (w["nside"], w["npol"], w["nbatch"], w["dtype"]) != (N, P, B, dtype):
            return {}, None
        return {k: v["traffic_bytes_per_launch"] for k, v in z["by_class"].items()}, z.get("total_bytes_per_step")
    except (OSError, KeyError, ValueError):
        return {}, None


def cpu_baseline(N, pol, nsteps, npT=np.float32):
    """The NumPy oracle (kind 'port': the Julia reference cannot run here) timed on the host cores: one ∇lnP
    evaluation of the same workload (bounded sample)."""
    import oracle as O
    t0 = time.time()
    so = O.load_sim(2.0, N, pol, npT, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), nsteps=nsteps)
    ds = so["ds"]
    fo, po = ds.mix(so["f"], so["phi"])
    t_setup = time.time() - t0
    t0 = time.time()
    ds._L = None
    lp, gf, gp = ds.grad_logpdf_mixed(fo, po)
    dt = time.time() - t0
    return dict(value=1.0 / dt, unit="steps/s", cores=int(os.environ.get("CMBL_ORACLE_FFT_WORKERS", os.cpu_count() or 1)),
                kind="port", sample=f"1 ∇logpdf(Mixed) evaluation, {N}² {pol} {np.dtype(npT).name}, n={nsteps}, NumPy/SciPy-pocketfft oracle "
                f"({dt:.2f} s; setup {t_setup:.1f} s not counted)")


CONFIGS = {2: dict(nside=512, pol="P", dtype="f32", nrk=7), 3: dict(nside=1024, pol="IP", dtype="f32", nrk=7),
           5: dict(nside=2048, pol="P", dtype="f64", nrk=10)}


def config_extras(C, torch, cfg, sim, timeit):
    """the operations BASELINE.json names for configuration `cfg`, timed outside the headline region"""
    ds, f, phi = sim["ds"], sim["f"], sim["phi"]
    fm = f.to(C.MAP)
    L = ds.L(phi)
    gl = fm.to(C.FOURIER)
    ex = {"L*f_ms": timeit(lambda: L * fm), "L'g_ms": timeit(lambda: L.adjoint * gl)}
    if cfg == 2:
        t0 = time.perf_counter(); fw, h = ds.argmaxf_logpdf(phi); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ex.update(wiener_cg_iterations=len(h), wiener_cg_ms=dt * 1e3, wiener_cg_ms_per_iteration=dt * 1e3 / len(h))
    if cfg == 3:
        p0 = C.Field(sim["proj"], torch.zeros_like(phi.arr), C.FOURIER)
        C.MAP_joint_step(ds, p0, cg_nsteps=100)
        t0 = time.perf_counter(); st = C.MAP_joint_step(ds, p0, cg_nsteps=100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ex.update(map_joint_step_ms=dt * 1e3, map_joint_cg_iterations=len(st["cg_hist"]), map_joint_linesearch_evals=st["linesearch_evals"],
                  map_joint_note="one MAP_joint step from ϕ = 0: Wiener CG capped at 100 iterations + ∇logpdf(Mixed) + Brent line search")
    if cfg == 5:
        C.quadratic_estimate(ds, "EB")
        t0 = time.perf_counter(); C.quadratic_estimate(ds, "EB"); torch.cuda.synchronize()
        ex["quadratic_estimate_EB_ms"] = (time.perf_counter() - t0) * 1e3
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nside", type=int, default=1024)
    ap.add_argument("--pol", default="P", choices=["I", "P", "IP"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--nrk", type=int, default=7, help="LenseFlow RK4 steps")
    ap.add_argument("--nbatch", type=int, default=1, help="chains per GPU (batch dim 4)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 5], help="BASELINE.json configuration (0 = the headline workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo lets the N>1 logic be exercised on a box with fewer GPUs than ranks")
    args = ap.parse_args()
    if args.config:
        for k, v in CONFIGS[args.config].items():
            setattr(args, k, v)

    import torch
    import cmblensing_jl_amd as C

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "gloo":
            local = local % max(torch.cuda.device_count(), 1)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    cdev = "cpu" if args.dist_backend == "gloo" else "cuda"          # where the (tiny) collective payloads live
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")

    N, pol, B, nrk = args.nside, args.pol, args.nbatch, args.nrk
    P = {"I": 1, "P": 2, "IP": 3}[pol]
    tT, npT, sz = (torch.float32, np.float32, 4) if args.dtype == "f32" else (torch.float64, np.float64, 8)
    # every rank = an independent chain: different simulation seeds per rank (seed = base + chain id)
    seeds = (1 + 1000 * rank, 2 + 1000 * rank, 3 + 1000 * rank)
    sim = C.load_sim(2.0, N, pol, synthetic_cls(), T=tT, device=local, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0),
                     nsteps=nrk, Nbatch=B, seeds=seeds)
    ds, proj = sim["ds"], sim["proj"]
    fo, po = ds.mix(sim["f"], sim["phi"])

    def step():
        return ds.gradient_logpdf_mixed(fo, po)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        lp, gf, gp = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lp, gf, gp = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the trivial result gather: per-chain logpdf scalars to every rank over RCCL
        mine = torch.tensor(lp, device=cdev, dtype=torch.float64)
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        lps = torch.cat(allp).cpu().numpy()
    else:
        lps = np.asarray(lp)
    assert np.all(np.isfinite(lps)), lps
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    out = {
        "metric": f"LenseFlow+∇logP steps/sec (∇logpdf(Mixed) evaluations/s, LenseFlow n={nrk}, whole job)",
        "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{N}² flat-sky {pol} (npol={P}), θpix=2′, ∇logpdf(Mixed(ds)) step = precompute + L\\f° + L·f + 2 δ-flows "
                               f"+ diag/mask/reductions; 1° apodised border mask, LowPass(3000), 3 μK′ noise"
                               + (f" [BASELINE.json configs[{args.config - 1}]]" if args.config else ""),
                   "nside": N, "npol": P, "chains_per_gpu": B, "rk4_steps": nrk, "parallelism": f"{world} independent chains (no data-path collective)"},
        "logpdf": [float(x) for x in lps],
    }

    if rank == 0 and not args.no_roofline:
        # per-launch HIP events on the library's stream over a re-run of (at most 20 of) the same steps.  The timed region above
        # runs each pol slice as its own launch chain on its own stream (concurrent half-size launches have no individual
        # bandwidth), so this leg switches that off: one launch over all slices, the same kernels.
        os.environ["CMBL_SLICE_STREAMS"] = "1"
        nprof = min(args.steps, 20)
        proj.prof_reset(); proj.prof_enable(True)
        for _ in range(nprof):
            step()
        proj.prof_enable(False)
        os.environ.pop("CMBL_SLICE_STREAMS")
        tab = proj.prof_table()
        tot = sum(v[0] for v in tab.values())
        ab = algorithmic_bytes(N, P, B, B, nrk, sz)
        traf, traf_step = measured_traffic(N, P, B, args.dtype)
        per = {}
        for k, (ms, nl) in sorted(tab.items(), key=lambda kv: -kv[1][0]):
            t_us = ms / nl * 1e3
            e = {"ms_per_step": ms / nprof, "avg_launch_us": t_us, "launches_per_step": nl / nprof}
            if k in KERNEL_SHARE:
                e["algorithmic_bytes_per_launch"] = KERNEL_SHARE[k](P, B, B) * ab["map_pass"]
                e["frac"] = e["algorithmic_bytes_per_launch"] / (t_us * 1e-6) / 8e12
            if k in traf:
                e["traffic_bytes_per_launch"] = traf[k]
                e["frac_traffic"] = traf[k] / (t_us * 1e-6) / 8e12
                e["frac_traffic_vs_6300"] = traf[k] / (t_us * 1e-6) / 6.3e12
            per[k] = e
        dom = max((k for k in per if k in KERNEL_SHARE), key=lambda k: per[k]["ms_per_step"])
        d = per[dom]
        whole = {"survey_algorithmic_GB": ab["grad_lnP"] / 1e9,
                 "survey_equivalent_GB_per_s": ab["grad_lnP"] / 1e9 / (ms_per_step * 1e-3),
                 "survey_equivalent_frac": ab["grad_lnP"] / 1e9 / (ms_per_step * 1e-3) / 8000.0,
                 "note": "survey_equivalent_* divides SURVEY §8(d)'s pass structure of the REFERENCE (31.5 GB at 1024² QU) by our step time: "
                         "a speed-up figure, not a bandwidth; traffic_* are the measured HBM bytes of our launches"}
        if traf_step:
            whole.update(traffic_GB_per_step=traf_step / 1e9, traffic_GB_per_s=traf_step / 1e9 / (ms_per_step * 1e-3),
                         frac_traffic=traf_step / (ms_per_step * 1e-3) / 8e12, frac_traffic_vs_6300=traf_step / (ms_per_step * 1e-3) / 6.3e12)
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": d["algorithmic_bytes_per_launch"] / (d["avg_launch_us"] * 1e-6) / 1e9,
                           "peak": 8000.0, "unit": "GB/s", "frac": d["frac"], "traffic": d.get("traffic_bytes_per_launch"),
                           "frac_traffic": d.get("frac_traffic"), "frac_traffic_vs_6300": d.get("frac_traffic_vs_6300"),
                           "avg_launch_us": d["avg_launch_us"], "launches_per_step": d["launches_per_step"],
                           "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"], "kernel_time_share": d["ms_per_step"] * nprof / tot,
                           "note": "per-kernel figures from a re-run with CMBL_SLICE_STREAMS=1 (one launch over all pol slices); "
                                   "value / ms_per_step / whole_step are the timed region with one launch chain per pol slice; "
                                   "`frac` = SURVEY-algorithmic bytes, `frac_traffic` = measured HBM bytes (profiles/r02_traffic_*.json)",
                           "per_kernel": per, "whole_step": whole}
    if rank == 0 and args.config:
        def timeit(fn, n=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        out["extras"] = config_extras(C, torch, args.config, sim, timeit)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, pol, nrk, npT)
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
