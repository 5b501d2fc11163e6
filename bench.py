#!/usr/bin/env python3
"""Headline benchmark: LenseFlow + ∇logP steps/sec on a 1024² QU flat-sky map (BASELINE.json `metric`).

One "step" = one evaluation of ∇_(f°,ϕ°) logpdf(Mixed(ds)) (the reference's "∇lnP" row,
test/runbenchmarks.jl:120): precompute(ϕ) + 1 inverse flow + 1 forward flow + 2 δ-flows +
the Fourier-diagonal / mask / reduction work, LenseFlow n = 7 RK4 steps, fp32, inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nside 1024] [--pol P] [--nbatch B] [--config {2,3,5}] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--config C` runs BASELINE.json's configuration C (2: 512² QU fp32; 3: 1024² T+QU fp32; 5: 2048² QU fp64 n=10) with the same step and
adds that configuration's own operations (L*f, L'g, Wiener CG / one MAP_joint step / quadratic_estimate) under `extras`.
`--only cg` (profiling aid, not a headline): the step is ONE Wiener-filter CG iteration (src/numerical_algorithms.jl:73-134).

roofline: every fraction in the line is a bandwidth.
  * `frac` / `achieved`: the COMPULSORY bytes of the dominant kernel's launch -- every array the launch must read or write once,
    derived per kernel in `compulsory_bytes()` below and in DESIGN.md §5 -- divided by the launch's mean duration (the kernel's own
    start/stop timestamps, recorded by the library with hipExtLaunchKernel events on the stream it launches on) and by 8 TB/s.
  * `traffic`: bytes the L2 exchanged with the fabric for the same launch, from separate rocprofv3 `--pmc FETCH_SIZE` / `--pmc
    WRITE_SIZE` passes (profiles/rNN_traffic_*.json of the newest round, calibrated on known-size copies: profiles/rNN_counter_calibration.json);
    `traffic_over_compulsory` ≈ 1 means no wasted re-reads.  The counters include Infinity-Cache hits.
  * `survey_equivalent_*` (under `whole_step` only): SURVEY.md §8(d)'s pass structure of the REFERENCE divided by our step time -- a
    speed-up figure that may exceed the peak because the fused kernels do not move those bytes; never used for `frac`.

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
PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
STREAM_GBS = 6300.0        # what a streaming copy achieves on the part (same guide)


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


def survey_bytes(N, P, B, Bphi, n, s):
    """SURVEY.md §8(d) formulas for the REFERENCE's pass structure (unit = one map-pass = N²·s bytes)."""
    mp = N * N * s
    lf = 4 * n * (15 * P * B + 2 * Bphi) * mp
    delta = 4 * n * (30 * P * B + 30 * B + 7 * Bphi) * mp
    pre = (18 + 10 * (2 * n + 1)) * Bphi * mp
    grad = pre + 2 * lf + 2 * delta + 60 * P * B * mp
    return dict(map_pass=mp, lenseflow=lf, delta_flow=delta, precompute=pre, grad_lnP=grad)


def compulsory_bytes(kernel, Ny, Nx, P, B, Bphi, n, s):
    """Bytes one launch of `kernel` must move once (mean over the 4n launches of a flow), for the layouts of DESIGN.md §2:
    map = Ny·Nx·s (real, [x][y]); mixed = mixed_rows(Ny/2+1)·Nx·2s (y-transformed, rows padded to a multiple of 4); F = (Ny/2+1)·Nx·2s.
    S = P·B slices.  Stage-dependent terms: the RK accumulator is not read in stage 1 of 4 (¾), stage 4 writes the state instead of
    the accumulator (same size), the last launch of a flow writes no next-stage input (1 − 1/4n).  p(t) (two maps per ϕ slot) is
    counted once per ϕ slot: the pol slices of a slot read the same lines within one launch (L2 hits; confirmed by the counters).
    Returns None for kernels without a model."""
    nyh = Ny // 2 + 1
    mp, mx, F = Ny * Nx * s, ((nyh + 3) & ~3) * Nx * 2 * s, nyh * Nx * 2 * s
    S, nst = P * B, 4 * n
    nl = 1.0 - 1.0 / nst
    adj_x = S * (2 * mx + F + 0.75 * F + F + nl * mx)                    # Wx, Wy; Y0, acc; acc|Y0; Hnext
    table = {
        "x_grad": S * 2 * mx,                                           # A -> Gx
        "flow_y_fwd": S * (2 * mx + mp + 0.75 * mp + mp + nl * mx) + Bphi * 2 * mp,      # Gx, A; y0, acc; acc|y0; Anext; p(t)
        "adj_y": S * 3 * mx + Bphi * 2 * mp,                            # H; Wx, Wy; p(t)
        "adj_x": adj_x,
        "delta_cols": S * (3 * mx + mp + 0.75 * mp + mp + 2 * mp + 2 * mx + nl * mx) + Bphi * 2 * mp,   # Gx, A, H; y0, acc; acc|y0; w1, w2; Wx, Wy; Anext; p(t)
        "delta_rows": adj_x + S * nl * 2 * mx,                          # the δf row pass + ∂x of the next stage's f (not in the last launch)
        "dphi_reduce": Bphi * 5 * mp + nst * 2 * S * mp + 5 * B * mp,   # five ϕ maps; per-stage products; five reduced maps
        "y_r2c": None, "y_c2r": None,
    }
    return table.get(kernel)


def measured_traffic(N, P, B, dtype, n, unit="grad"):
    """L2<->fabric bytes from the committed rocprofv3 PMC passes (profiles/rNN_traffic_*.json, newest round; tools/run_traffic.sh): ({kernel class:
    bytes per launch}, bytes per step, file name) or ({}, None, None) when no profile matches this workload.  Counters cannot be
    read from inside this process."""
    import glob
    pol = {1: "I", 2: "QU", 3: "IQU"}[P]
    tail = f"_traffic_{'cg_' if unit == 'cg' else ''}{N}{pol}_{dtype}{'_B%d' % B if B > 1 else ''}.json"
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]" + tail)))      # the newest round's profile of this workload
    if not found:
        return {}, None, None
    path, name = found[-1], os.path.basename(found[-1])
    try:
        z = json.load(open(path))
        w = z["workload"]
        if (w["nside"], w["npol"], w["nbatch"], w["dtype"], w.get("nrk", n)) != (N, P, B, dtype, n):
            return {}, None, None
        return {k: v["traffic_bytes_per_launch"] for k, v in z["by_class"].items()}, z.get("total_bytes_per_step"), "profiles/" + name
    except (OSError, KeyError, ValueError):
        return {}, None, None


PROF_KERNEL = {"delta_cols": "k_delta_cols<", "delta_rows": "k_delta_rows<", "flow_y_fwd": "k_flow_y_fwd<", "adj_y": "k_adj_y<", "adj_x": "k_adj_x<",
               "dphi_reduce": "k_dphi_reduce<"}


def profiler_mean_us(kernel, N, P, B, dtype):
    """Mean duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this workload (profiles/), or None.  The
    profiler's kernel times run 3-7 % above the un-profiled ones; reported next to the live figure, never instead of it."""
    import csv
    pol = {1: "I", 2: "QU", 3: "IQU"}[P]
    tag = f"{N}{pol}_{dtype}{'_B%d' % B if B > 1 else ''}"
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_kernel_stats_{tag}_50steps.csv")))[-1:] \
        + sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_kernel_stats_{tag}.csv")))[-1:]
    for path in cands:
        name = os.path.basename(path)
        if kernel in PROF_KERNEL and os.path.isfile(path):
            for r in csv.DictReader(open(path)):
                if PROF_KERNEL[kernel] in r["Name"]:
                    return float(r["AverageNs"]) / 1e3, "profiles/" + name
    return None, None


def cpu_baseline(N, pol, nsteps, npT=np.float32, dev=None, dev_exact=None):
    """The NumPy oracle (kind 'port': the Julia reference cannot run here) on the host cores: one ∇lnP evaluation of the same
    workload (bounded sample), timed in the workload's precision.  With `dev` (the device's own inputs and outputs of the timed step,
    as host arrays) the FLOAT64 oracle is also evaluated on exactly those fp32/fp64-rounded inputs and the HIP results are compared with
    it: `parity_at_config` -- the parity statement at the headline size, where the GPU tests' oracle comparisons used to stop at 512².
    `dev_exact`: the same for the step run with the REFERENCE's arithmetic (aliased δϕ velocity, src/lenseflow.jl:198-200; plain
    working-precision sums, src/util.jl:288-316), against the oracle run the same way.
    Returns (cpu_baseline, parity_at_config | None, parity of dev_exact | None)."""
    import oracle as O
    rel = lambda a, b: float(np.linalg.norm((np.asarray(a) - b).ravel()) / np.linalg.norm(np.asarray(b).ravel()))
    pm = dict(pad_deg=1.0, apod_deg=1.0)
    sims = {}

    def one(T, inputs=None):
        t0 = time.time()
        if T not in sims:
            sims[T] = O.load_sim(2.0, N, pol, T, pixel_mask=pm, nsteps=nsteps)
        so = sims[T]
        ds = so["ds"]
        if inputs is None:
            fo, po = ds.mix(so["f"], so["phi"])
            quirk = False
        else:
            ds.d = inputs["d"].astype(np.complex128)
            fo, po, quirk = inputs["fo"].astype(np.float64), inputs["po"].astype(np.complex128), inputs["alias_quirk"]
        t_setup = time.time() - t0
        t0 = time.time()
        ds._L = None
        res = ds.grad_logpdf_mixed(fo, po, alias_quirk=quirk)
        return res, time.time() - t0, t_setup

    def parity_of(dv, lp, gf, gp):
        return {"logpdf_rel": float(np.max(np.abs((np.asarray(dv["lp"]) - lp) / lp))), "gf_rel_l2": rel(dv["gf"], gf), "gphi_rel_l2": rel(dv["gp"], gp),
                "logpdf_hip": [float(x) for x in np.atleast_1d(dv["lp"])], "logpdf_oracle": [float(x) for x in np.atleast_1d(lp)],
                "alias_quirk": bool(dv["alias_quirk"]), "sum_mode": dv["sum_mode"],
                "oracle": "float64 NumPy/SciPy oracle (oracle/dataset.py grad_logpdf_mixed) on the device's own rounded f°, ϕ°, d; operators "
                          "rebuilt from the same spectra / seeds in float64",
                "tolerance": {"f32": "3 x the errors measured at this size, tests/test_gpu_headline_parity.py: logpdf 5e-8, ∇f° 6e-6 (QU) / 1.2e-4 (T+QU), "
                                     "∇ϕ° 7.5e-7 / 9.5e-6", "f64": "1e-10 / 1e-9 / 1e-9"}[
                    "f32" if dv["fo"].dtype == np.float32 else "f64"]}

    parity = parity_exact = None
    if dev is not None and npT == np.float64:
        (lp, gf, gp), dt, t_setup = one(np.float64, dev)               # one evaluation serves both legs
    else:
        _, dt, t_setup = one(npT)
        if dev is not None:
            (lp, gf, gp), dt64, _ = one(np.float64, dev)
    if dev is not None:
        parity = parity_of(dev, lp, gf, gp)
    if dev_exact is not None:
        (lpe, gfe, gpe), _, _ = one(np.float64, dev_exact)
        parity_exact = parity_of(dev_exact, lpe, gfe, gpe)
        parity_exact["note"] = ("the step run with the reference's arithmetic as written (alias_quirk = true: src/lenseflow.jl:198-200; sums in the "
                                "working precision: src/util.jl:288-316) against the float64 oracle with the same aliasing; logpdf carries the "
                                "rounding of a working-precision sum over all modes")
    base = dict(value=1.0 / dt, unit="steps/s", cores=int(os.environ.get("CMBL_ORACLE_FFT_WORKERS", os.cpu_count() or 1)),
                kind="port", sample=f"1 ∇logpdf(Mixed) evaluation, {N}² {pol} {np.dtype(npT).name}, n={nsteps}, NumPy/SciPy-pocketfft oracle "
                f"({dt:.2f} s; setup {t_setup:.1f} s not counted)")
    return base, parity, parity_exact


CONFIGS = {2: dict(nside=512, pol="P", dtype="f32", nrk=7), 3: dict(nside=1024, pol="IP", dtype="f32", nrk=7),
           5: dict(nside=2048, pol="P", dtype="f64", nrk=10)}


def kernel_table(proj, run, nunits, N, P, B, nrk, sz, traf):
    """Per-kernel-class timings of `run()` (the library's per-launch events: each kernel's own start/stop timestamps) with the
    compulsory bytes of the classes that have a model and the measured traffic where a profile exists."""
    proj.prof_reset(); proj.prof_enable(True)
    run()
    proj.prof_enable(False)
    tab = proj.prof_table()
    per = {}
    for k, (ms, nl) in sorted(tab.items(), key=lambda kv: -kv[1][0]):
        t_us = ms / nl * 1e3
        e = {"ms_per_unit": ms / nunits, "avg_launch_us": t_us, "launches_per_unit": nl / nunits}
        cb = compulsory_bytes(k, N, N, P, B, B, nrk, sz)
        if cb:
            e["compulsory_bytes_per_launch"] = cb
            e["achieved_GBps"] = cb / (t_us * 1e-6) / 1e9
            e["frac"] = e["achieved_GBps"] / PEAK_GBS
        if k in traf:
            e["traffic_bytes_per_launch"] = traf[k]
            e["frac_traffic"] = traf[k] / (t_us * 1e-6) / 1e9 / PEAK_GBS
            if cb:
                e["traffic_over_compulsory"] = traf[k] / cb
        per[k] = e
    return per


def cg_block(C, torch, sim, nit=40):
    """One Wiener-filter CG iteration (src/numerical_algorithms.jl:73-134 driving src/maximization.jl:17-42), per kernel class."""
    ds, phi, proj = sim["ds"], sim["phi"], sim["proj"]
    ds.argmaxf_logpdf(phi, tol=0.0, nsteps=4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ds.argmaxf_logpdf(phi, tol=0.0, nsteps=nit)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    N, P = proj.Nx, ds.P
    sz = 4 if proj.T == torch.float32 else 8
    dtype = "f32" if sz == 4 else "f64"
    traf, traf_unit, src = measured_traffic(N, P, 1, dtype, ds.L.nsteps, unit="cg")
    old = proj.set_option("slice_streams", 1)
    per = kernel_table(proj, lambda: ds.argmaxf_logpdf(phi, tol=0.0, nsteps=nit), nit, N, P, 1, ds.L.nsteps, sz, traf)
    proj.set_option("slice_streams", old)
    flow = ("x_grad", "flow_y_fwd", "adj_y", "adj_x")
    out = {"ms_per_iteration": dt / nit * 1e3, "iterations_timed": nit,
           "launches_per_iteration": sum(v["launches_per_unit"] for v in per.values()),
           "non_flow_launches_per_iteration": sum(v["launches_per_unit"] for k, v in per.items() if k not in flow),
           "per_kernel": per,
           "note": "ms_per_iteration: wall clock of a fixed-length run (tol = 0) / iterations, set-up launches (b = L'B'M'Cn^-1 d, first "
                   "residual) included; per_kernel from a second run with per-launch events, one launch over all pol slices"}
    if traf_unit:
        out.update(traffic_GB_per_iteration=traf_unit / 1e9, frac_traffic=traf_unit / (dt / nit) / 1e9 / PEAK_GBS, traffic_source=src)
    return out


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
        ex["cg_iteration"] = cg_block(C, torch, sim)
    if cfg == 3:
        p0 = C.Field(sim["proj"], torch.zeros_like(phi.arr), C.FOURIER)
        C.MAP_joint_step(ds, p0, cg_nsteps=100)
        t0 = time.perf_counter(); st = C.MAP_joint_step(ds, p0, cg_nsteps=100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ex.update(map_joint_step_ms=dt * 1e3, map_joint_cg_iterations=len(st["cg_hist"]), map_joint_linesearch_evals=st["linesearch_evals"],
                  map_joint_note="one MAP_joint step from ϕ = 0: Wiener CG capped at 100 iterations + ∇logpdf(Mixed) + Brent line search")
        ex["cg_iteration"] = cg_block(C, torch, sim)
    if cfg == 5:
        # quadratic_estimate(:EB) (src/quadratic_estimate.jl:29-200): the library's own loop body (cmbl_quadratic_estimate: control flow on
        # the host inside the library, every field operation a launch) is what a non-Python host calls; the Python driver of the same
        # estimator (host-side NumPy algebra between the launches) is kept beside it
        for name, fn in (("quadratic_estimate_EB_native_ms", C.quadratic_estimate_native), ("quadratic_estimate_EB_ms", C.quadratic_estimate)):
            fn(ds, "EB"); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); fn(ds, "EB"); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            ex[name] = min(ts)
        ex["quadratic_estimate_note"] = ("best of 3; quadratic_estimate_EB_native_ms = cmbl_quadratic_estimate (C ABI, planes resident on the device), quadratic_estimate_EB_ms = "
                                         "drivers.quadratic_estimate (the Python driver: the key rounds 2-4 reported; round 5 reported the native call under it)")
    return ex


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch_command(args, argv, env):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): the command that starts the N ranks, one
    process per GPU, on this node -- or None when this process is already a rank (or N = 1).  The reference needs no external
    launcher either: `sample_joint` / `MAP_marg` `pmap` over the workers the session holds, one GPU each
    (src/sampling.jl:266,292, src/util_parallel.jl:73-102)."""
    if args.gpus <= 1 or "WORLD_SIZE" in env or "RANK" in env:
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def launch_env(env, nranks, ncpu=None):
    """environment of the N ranks a self-launch starts: the host cores are SHARED by the ranks during set-up (`load_sim`: NumPy / pocketfft,
    BLAS) -- every rank defaulting to all of them would oversubscribe the host N-fold.  Values the caller set are kept."""
    ncpu = ncpu or os.cpu_count() or 1
    per = str(max(1, ncpu // max(1, nranks)))
    out = dict(env, HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "CMBL_ORACLE_FFT_WORKERS"):
        out.setdefault(k, per)
    return out


def assign_devices(world, ndev, backend):
    """rank -> device index: one GPU per rank, each GPU used once (src/util_parallel.jl:73-102 assigns a unique GPU per worker and
    errors otherwise).  Only the `gloo` backend -- a test aid for boxes with fewer GPUs than ranks -- may share devices."""
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if backend != "gloo" and ndev < world:
        raise SystemExit(f"--gpus {world} but only {ndev} GPU(s) visible: one rank per GPU (use --dist-backend gloo to share devices in tests)")
    return [r % ndev for r in range(world)]


def dry_run(args, torch, dist, rank, world, local, seeds):
    """`--dry-run-ranks N`: everything of the N-GPU run except the kernels -- N processes, rendezvous on 127.0.0.1, one (mocked) device
    per rank, seeds = base + 1000 rank, W + K stand-in steps between the same barriers, MAX over ranks, the per-chain gather and the
    collective report -- over gloo on the CPU, so that the launch path of `python bench.py --gpus 8` is exercised where no 8-GPU node
    exists (tests/test_bench_launch.py).  The reference's counterpart: one worker per GPU, src/util_parallel.jl:73-102."""
    def step():
        time.sleep(1e-3)
        return [float(seeds[0])]
    def barrier():
        if dist is not None:
            dist.barrier()
    for _ in range(args.warmup):
        step()
    barrier(); t0 = time.perf_counter()
    for _ in range(args.steps):
        lp = step()
    barrier(); dt = time.perf_counter() - t0
    coll = None
    lps = lp
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor(lp, dtype=torch.float64)
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        lps = [float(x) for x in torch.cat(allp)]
        me = {"rank": rank, "device": local, "name": "mock device (dry run)", "pci_bus_id": "mock:%02x" % local, "uuid": "mock-%d" % local,
              "pid": os.getpid(), "seeds": list(seeds)}
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        coll = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": everyone}
    if rank == 0:
        print(json.dumps({"metric": "dry run of the N-rank launch path (NOT a measurement)", "dry_run": True, "value": world * args.steps / dt, "unit": "stand-in steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "scaling": "weak",
                          "config": {"workload": "stand-in step (1 ms sleep); mocked devices", "parallelism": f"{world} independent chains (no data-path collective)"},
                          "logpdf": lps, "collective": coll}, ensure_ascii=False))
    if dist is not None:
        dist.destroy_process_group()


def clock_ramp(step, sync, block=5, tol=0.01, max_blocks=60):
    """Untimed spin before the warm-up steps: blocks of `block` steps until two consecutive blocks agree to `tol` (the chip's clocks
    ramp over the first ~0.1-0.3 s of work: a cold K = 20 run read 8 % low in round 3).  Returns (steps spun, last block ms/step)."""
    prev, n = None, 0
    for _ in range(max_blocks):
        sync(); t0 = time.perf_counter()
        for _ in range(block):
            step()
        sync(); cur = (time.perf_counter() - t0) / block
        n += block
        if prev is not None and abs(cur - prev) <= tol * prev:
            break
        prev = cur
    return n, cur * 1e3


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
    ap.add_argument("--only", default="", choices=["", "cg"], help="profiling aid: the step is one Wiener-CG iteration instead of ∇lnP")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo lets the N>1 logic be exercised on a box with fewer GPUs than ranks")
    ap.add_argument("--no-ramp", action="store_true", help="skip the untimed clock-ramp spin before the warm-up steps")
    ap.add_argument("--dry-run-ranks", type=int, default=0, metavar="N",
                    help="dress rehearsal of the N-GPU run on a box without the GPUs: the self-launch, rank -> device assignment (N mocked devices), "
                         "per-rank seeds, barriers, MAX-over-ranks timing, the result gather and the collective report all run (gloo, CPU); the "
                         "step itself is a stand-in.  The line carries \"dry_run\": true and is never a measurement")
    args = ap.parse_args()
    dry = args.dry_run_ranks > 0
    if dry:
        args.gpus, args.dist_backend = args.dry_run_ranks, "gloo"
        args.no_roofline = args.no_extras = args.no_cpu_baseline = args.no_ramp = True
    if args.config:
        for k, v in CONFIGS[args.config].items():
            setattr(args, k, v)
    cmd = self_launch_command(args, sys.argv[1:], os.environ)
    if cmd is not None:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU), like the reference's own
        # `pmap` over workers (src/sampling.jl:266,292), and rank 0's JSON line passes through on stdout
        import subprocess
        env = launch_env(os.environ, args.gpus)
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    if not dry:
        import cmblensing_jl_amd as C

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    rccl = None
    ndev = args.dry_run_ranks if dry else (torch.cuda.device_count() if torch.cuda.is_available() else 0)
    local = assign_devices(world, ndev, "nccl" if dry else args.dist_backend)[local]        # the rehearsal keeps the one-GPU-per-rank rule
    if world > 1 or os.environ.get("CMBL_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    cdev = "cpu" if args.dist_backend == "gloo" else "cuda"          # where the (tiny) collective payloads live
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    N, pol, B, nrk = args.nside, args.pol, args.nbatch, args.nrk
    P = {"I": 1, "P": 2, "IP": 3}[pol]
    tT, npT, sz = (torch.float32, np.float32, 4) if args.dtype == "f32" else (torch.float64, np.float64, 8)
    # every rank = an independent chain: different simulation seeds per rank (seed = base + chain id)
    seeds = (1 + 1000 * rank, 2 + 1000 * rank, 3 + 1000 * rank)
    if dry:
        return dry_run(args, torch, dist, rank, world, local, seeds)
    sim = C.load_sim(2.0, N, pol, synthetic_cls(), T=tT, device=local, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0),
                     nsteps=nrk, Nbatch=B, seeds=seeds)
    ds, proj = sim["ds"], sim["proj"]
    fo, po = ds.mix(sim["f"], sim["phi"])

    if args.only == "cg":
        # one unit = one CG iteration: a fixed-length solve (tol = 0) of steps + warmup iterations in total
        def run_all():
            return ds.argmaxf_logpdf(sim["phi"], tol=0.0, nsteps=args.steps + args.warmup)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run_all()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(json.dumps({"metric": "Wiener-filter CG iterations/s (profiling aid)", "value": (args.steps + args.warmup) / dt, "unit": "iterations/s",
                          "ms_per_step": dt / (args.steps + args.warmup) * 1e3, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "config": {"workload": f"{N}² {pol} Wiener CG iteration"}}, ensure_ascii=False))
        return

    def step():
        return ds.gradient_logpdf_mixed(fo, po)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ramp_steps, ramp_ms = (0, None) if args.no_ramp else clock_ramp(step, torch.cuda.synchronize)
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
        # who took part, as the collective library sees it: rank, device index and PCI bus id of every rank
        prop = torch.cuda.get_device_properties(local)
        me = {"rank": rank, "device": local, "name": prop.name, "pci_bus_id": getattr(prop, "pci_bus_id", None), "uuid": str(getattr(prop, "uuid", ""))}
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": everyone}
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
                   "nside": N, "npol": P, "chains_per_gpu": B, "rk4_steps": nrk, "parallelism": f"{world} independent chains (no data-path collective)",
                   # the arithmetic that was timed: CMBL_REFERENCE_EXACT=1 switches both to the reference's
                   "alias_quirk": bool(ds.alias_quirk), "sum_accuracy_mode": "working" if C.reference_exact() else "float64",
                   "reference_exact": bool(C.reference_exact()),
                   "arithmetic_note": "default of the Python host: consistent δϕ velocity + float64 accumulation (DESIGN.md §3 Q1); the reference's own "
                                      "arithmetic is timed and compared under extras.reference_exact"},
        "clock_ramp": {"untimed_steps": ramp_steps, "last_block_ms_per_step": ramp_ms,
                       "note": "untimed spin before the W warm-up steps, until two consecutive 5-step blocks agree to 1 %"},
        "logpdf": [float(x) for x in lps],
    }
    if rccl is not None:
        out["collective"] = rccl

    if rank == 0 and world == 1 and not args.no_roofline:
        # (N > 1: no rank does extra work after the timed region, so none idles at destroy_process_group while rank 0 profiles)
        # per-launch timestamps over a re-run of (at most 20 of) the same steps.  The timed region above runs each pol slice as its
        # own launch chain on its own stream (concurrent half-size launches have no individual bandwidth), so this leg switches that
        # off: one launch over all slices, the same kernels.
        old_ss = proj.set_option("slice_streams", 1)
        nprof = min(args.steps, 20)
        traf, traf_step, traf_src = measured_traffic(N, P, B, args.dtype, nrk)
        per = kernel_table(proj, lambda: [step() for _ in range(nprof)], nprof, N, P, B, nrk, sz, traf)
        proj.set_option("slice_streams", old_ss)
        tot = sum(v["ms_per_unit"] for v in per.values())
        dom = max((k for k in per if "frac" in per[k]), key=lambda k: per[k]["ms_per_unit"])
        d = per[dom]
        sb = survey_bytes(N, P, B, B, nrk, sz)
        modelled = sum(v["compulsory_bytes_per_launch"] * v["launches_per_unit"] for v in per.values() if "frac" in v)
        whole = {"launches_per_step": sum(v["launches_per_unit"] for v in per.values()),
                 "non_flow_launches_per_step": sum(v["launches_per_unit"] for k, v in per.items() if "frac" not in v),
                 "compulsory_GB_per_step_flow_kernels": modelled / 1e9,
                 "flow_kernels_GBps": modelled / 1e9 / (ms_per_step * 1e-3),
                 "survey_equivalent_GB": sb["grad_lnP"] / 1e9,
                 "survey_equivalent_GBps": sb["grad_lnP"] / 1e9 / (ms_per_step * 1e-3),
                 "note": "survey_equivalent_* = SURVEY §8(d)'s pass structure of the REFERENCE divided by our step time: a speed-up figure, "
                         "not a bandwidth (the fused kernels do not move those bytes).  traffic_* = measured L2<->fabric bytes of all launches of a step."}
        if traf_step:
            whole.update(traffic_GB_per_step=traf_step / 1e9, traffic_GBps=traf_step / 1e9 / (ms_per_step * 1e-3),
                         frac_traffic=traf_step / (ms_per_step * 1e-3) / 1e9 / PEAK_GBS,
                         frac_traffic_vs_streaming=traf_step / (ms_per_step * 1e-3) / 1e9 / STREAM_GBS)
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": d["achieved_GBps"], "peak": PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
                           "traffic": d.get("traffic_bytes_per_launch"), "traffic_over_compulsory": d.get("traffic_over_compulsory"),
                           "traffic_source": traf_src, "frac_vs_streaming_copy": d["achieved_GBps"] / STREAM_GBS,
                           "avg_launch_us": d["avg_launch_us"], "launches_per_step": d["launches_per_unit"],
                           "compulsory_bytes_per_launch": d["compulsory_bytes_per_launch"], "kernel_time_share": d["ms_per_unit"] / tot,
                           "note": "achieved = compulsory bytes of one launch (bench.py compulsory_bytes, DESIGN.md §5) / mean kernel duration "
                                   "(the kernel's own start/stop timestamps, hipExtLaunchKernel events; one launch over all pol slices, "
                                   "option slice_streams = 1); value / ms_per_step / whole_step are the timed region with one launch chain per pol "
                                   "slice.  traffic = measured L2<->fabric bytes per launch (rocprofv3 PMC, Infinity-Cache hits included)",
                           "per_kernel": per, "whole_step": whole}
        pus, psrc = profiler_mean_us(dom, N, P, B, args.dtype)
        if pus:
            out["roofline"]["rocprofv3"] = {"avg_launch_us": pus, "frac": d["compulsory_bytes_per_launch"] / (pus * 1e-6) / 1e9 / PEAK_GBS, "source": psrc,
                                            "note": "the same kernel's mean in the committed rocprofv3 kernel-trace summary of this workload; the profiler's "
                                                    "kernel times are 3-7 % above the un-profiled ones (DESIGN.md §5)"}
    if rank == 0 and not args.no_extras and world == 1:
        def timeit(fn, n=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        if args.config:
            out["extras"] = config_extras(C, torch, args.config, sim, timeit)
        elif B == 1 and not args.no_roofline:
            # the driver-level hot loop: one Wiener-filter CG iteration on this workload and on its T+QU sibling
            ex = {"cg_iteration": {f"{N}_{pol}": cg_block(C, torch, sim)}}
            if pol == "P" and args.dtype == "f32":
                sim3 = C.load_sim(2.0, N, "IP", synthetic_cls(), T=tT, device=local, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), nsteps=nrk)
                ex["cg_iteration"][f"{N}_IP"] = cg_block(C, torch, sim3)
                fo3, po3 = sim3["ds"].mix(sim3["f"], sim3["phi"])
                ex["grad_lnP_ms_IP"] = timeit(lambda: sim3["ds"].gradient_logpdf_mixed(fo3, po3), n=20)
                ex["grad_lnP_IP_note"] = f"{N}² T+QU (BASELINE configs[2] workload, the north_star target): ∇logpdf(Mixed) step, mean of 20"
                del sim3, fo3, po3
                # the same step with 8 independent chains per GPU as batch slots (north_star: chains / batch slots fill the chip): launches are
                # several residency rounds long and the phases of a launch overlap -- what the part delivers when it is not latency-bound
                sim8 = C.load_sim(2.0, N, pol, synthetic_cls(), T=tT, device=local, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), nsteps=nrk, Nbatch=8)
                fo8, po8 = sim8["ds"].mix(sim8["f"], sim8["phi"])
                ms8 = timeit(lambda: sim8["ds"].gradient_logpdf_mixed(fo8, po8), n=10)
                ex["eight_chains_per_gpu"] = {"ms_per_call": ms8, "evaluations_per_s": 8e3 / ms8,
                                              "note": f"{N}² {pol}, Nbatch = 8 in one call (bench.py --nbatch 8 is the full line of this workload)"}
                # ... and the Wiener-CG iteration in that mode (what `sample_joint` with several chains per GPU spends its time in)
                nit8 = 20
                sim8["ds"].argmaxf_logpdf(sim8["phi"], tol=0.0, nsteps=3)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sim8["ds"].argmaxf_logpdf(sim8["phi"], tol=0.0, nsteps=nit8)
                torch.cuda.synchronize(); ms_it8 = (time.perf_counter() - t0) / nit8 * 1e3
                cg1 = ex["cg_iteration"][f"{N}_{pol}"]
                ex["cg_iteration"][f"{N}_{pol}_B8"] = {"ms_per_iteration": ms_it8, "ms_per_iteration_per_chain": ms_it8 / 8, "iterations_timed": nit8}
                if "traffic_GB_per_iteration" in cg1:
                    ex["cg_iteration"][f"{N}_{pol}_B8"].update(
                        frac_traffic=8 * cg1["traffic_GB_per_iteration"] / ms_it8 / 1e-3 / PEAK_GBS,
                        traffic_note="8 x the counter traffic of the B = 1 iteration (no counter pass exists for the B = 8 iteration itself)")
                del sim8, fo8, po8
                # small maps (the reference's own test sizes): the one-launch flows of csrc/kernels_small.hpp against the two-launches-per-stage path
                sm = {}
                for nb in (1, 64):
                    sims = C.load_sim(2.0, 64, pol, synthetic_cls(), T=tT, device=local, pixel_mask=dict(pad_deg=0.2, apod_deg=0.2), nsteps=nrk, Nbatch=nb)
                    Ls, fms = sims["ds"].L(sims["phi"]), sims["f"].to(C.MAP)
                    gls = fms.to(C.FOURIER)
                    for v, name in ((0, "staged"), (1, "one_launch")):
                        sims["proj"].set_option("small_flow", v)
                        sm[f"B{nb}_{name}"] = {"L*f_ms": timeit(lambda: Ls * fms, n=20), "L'g_ms": timeit(lambda: Ls.adjoint * gls, n=20)}
                    del sims, Ls, fms, gls
                sm["note"] = "64² QU fp32: L*f / L'g with option small_flow = 0 (two launches per RK stage) and 1 (default: one launch per flow), B = 1 and B = 64"
                ex["small_maps_64"] = sm
                # a survey patch side that is not a power of two (3 * 2^k): the any-size path with its compile-time-plan transforms
                # (csrc/kernels_ct.hpp), and the run-time-planned transforms of rounds 2-4 for comparison (option gen_ct)
                Na = 768
                sima = C.load_sim(2.0, Na, pol, synthetic_cls(), T=tT, device=local, pixel_mask=dict(pad_deg=1.0, apod_deg=1.0), nsteps=nrk)
                foa, poa = sima["ds"].mix(sima["f"], sima["phi"])
                ms_ct = timeit(lambda: sima["ds"].gradient_logpdf_mixed(foa, poa), n=10)
                sima["proj"].set_option("gen_ct", 0)
                ms_rt = timeit(lambda: sima["ds"].gradient_logpdf_mixed(foa, poa), n=5)
                sima["proj"].set_option("gen_ct", 1)
                ex["any_size_768"] = {"ms_per_step": ms_ct, "ms_per_step_run_time_plans": ms_rt,
                                      "note": f"{Na}² {pol} fp32, the same ∇logpdf(Mixed) step through the any-size path (DESIGN.md §4.3, profiles/r06_anysize_times.txt)"}
                del sima, foa, poa
            out["extras"] = ex
    dev_exact = None
    if rank == 0 and world == 1 and B == 1 and not args.no_extras and not args.no_roofline and not C.reference_exact():
        # The same step with the REFERENCE's arithmetic as written: the aliased δϕ velocity (src/lenseflow.jl:198-200 + src/field_vectors.jl:48-49)
        # and plain sums in the working precision (src/util.jl:288-316).  The default line above times the consistent form with float64
        # accumulation; this is the mode a Julia caller of julia/CMBLensingHIPExt.jl gets by default.
        host = lambda t: t.detach().cpu().numpy()
        old_q = ds.alias_quirk
        ds.alias_quirk = True
        proj.set_sum_accuracy_mode("working")
        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nex = min(args.steps, 50)
            for _ in range(nex):
                lpe, gfe, gpe = step()
            torch.cuda.synchronize()
            ms_exact = (time.perf_counter() - t0) / nex * 1e3
            dev_exact = dict(fo=host(fo.arr), po=host(po.arr), d=host(sim["d"].arr), lp=np.asarray(lpe), gf=host(gfe.arr), gp=host(gpe.arr),
                             alias_quirk=True, sum_mode="working")
        finally:
            ds.alias_quirk = old_q
            proj.set_sum_accuracy_mode("float64")
        out.setdefault("extras", {})["reference_exact"] = {
            "ms_per_step": ms_exact, "steps_per_s": 1e3 / ms_exact, "steps_timed": nex, "alias_quirk": True, "sum_accuracy_mode": "working",
            "note": "the headline workload with the reference's arithmetic as written (CMBL_REFERENCE_EXACT=1 makes it the default of a process; "
                    "it is the default of the Julia glue julia/CMBLensingHIPExt.jl); same launches, the switch sits in k_dphi_reduce and in "
                    "the accumulators of the reductions"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        dev = None
        if B == 1:
            lp, gf, gp = step()
            host = lambda t: t.detach().cpu().numpy()
            dev = dict(fo=host(fo.arr), po=host(po.arr), d=host(sim["d"].arr), lp=np.asarray(lp), gf=host(gf.arr), gp=host(gp.arr),
                       alias_quirk=bool(ds.alias_quirk), sum_mode="working" if C.reference_exact() else "float64")
        out["cpu_baseline"], par, par_exact = cpu_baseline(N, pol, nrk, npT, dev, dev_exact)
        if par is not None:
            out["parity_at_config"] = par
        if par_exact is not None:
            out["extras"]["reference_exact"]["parity_at_config"] = par_exact
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
