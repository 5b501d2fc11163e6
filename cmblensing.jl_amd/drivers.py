"""Host-side drivers on top of the C-ABI operators: quadratic_estimate, MAP_joint, the HMC / Gibbs passes of sample_joint.

These mirror the reference's Julia drivers (src/quadratic_estimate.jl, src/maximization.jl:116-233, src/sampling.jl:14-46,
388-464): control flow on the host, every field operation a library call.
"""
import itertools
import os

import numpy as np
import torch

from .engine import Field, MAP, FOURIER, HARMONIC


# ---------------------------------------------------------------------------------------------------------------------
def brent_minimize(f, lo, hi, abs_tol=1e-4, rel_tol=None, max_iter=1000):
    """Brent's bounded 1-D minimiser (golden section + successive parabolic interpolation), the algorithm behind
    `Optim.optimize(f, lo, hi, Brent(); abs_tol)` used at src/maximization.jl:194-199.  Returns (xmin, fmin, nfev)."""
    rel_tol = np.sqrt(np.finfo(float).eps) if rel_tol is None else rel_tol
    golden = (3 - np.sqrt(5)) / 2
    x = w = v = lo + golden * (hi - lo)
    fx = fw = fv = f(x)
    nfev, step, old_step = 1, 0.0, 0.0
    for _ in range(max_iter):
        mid = (lo + hi) / 2
        tol = rel_tol * abs(x) + abs_tol
        if abs(x - mid) <= 2 * tol - (hi - lo) / 2:
            break
        p = q = 0.0
        if abs(old_step) > tol:
            r = (x - w) * (fx - fv)
            q = (x - v) * (fx - fw)
            p = (x - v) * q - (x - w) * r
            q = 2 * (q - r)
            if q > 0:
                p = -p
            else:
                q = -q
        if abs(p) < abs(q * old_step / 2) and p > q * (lo - x) and p < q * (hi - x):
            old_step, step = step, p / q
            xt = x + step
            if (xt - lo) < 2 * tol or (hi - xt) < 2 * tol:
                step = tol if x < mid else -tol
        else:
            old_step = (hi - x) if x < mid else (lo - x)
            step = golden * old_step
        u = x + (step if abs(step) >= tol else (tol if step > 0 else -tol))
        fu = f(u)
        nfev += 1
        if fu <= fx:
            if u < x:
                hi = x
            else:
                lo = x
            v, fv, w, fw, x, fx = w, fw, x, fx, u, fu
        else:
            if u < x:
                lo = u
            else:
                hi = u
            if fu <= fw or w == x:
                v, fv, w, fw = w, fw, u, fu
            elif fu <= fv or v == x or v == w:
                v, fv = u, fu
    return x, fx, nfev


# ---------------------------------------------------------------------------------------------------------------------
def _eps3(a, b):
    return 1 if (a, b) == (1, 2) else (-1 if (a, b) == (2, 1) else 0)


def _inds(D):
    return list(itertools.product((1, 2), repeat=D))


class _Legs:
    """memoised QE_leg evaluations (src/quadratic_estimate.jl:83-91): key = (field id, n, p1, p2)"""

    def __init__(self, proj):
        self.proj, self.cache, self.keep = proj, {}, []

    def __call__(self, C, *inds):
        n = sum(1 for x in inds if isinstance(x, int))
        first = [x[0] if isinstance(x, tuple) else x for x in inds]
        key = (id(C), n, first.count(1), first.count(2))
        if key not in self.cache:
            self.keep.append(C)
            self.cache[key] = self.proj.qe_leg(C, n, first.count(1), first.count(2))
        return self.cache[key]


def quadratic_estimate(ds, which=None, wiener_filtered=True, AL=None):
    """`quadratic_estimate(ds, which)` (src/quadratic_estimate.jl:29-47) with unlensed weights, using the Fourier-diagonal
    approximations M̂·B̂, Cn̂ exactly like the reference.  Returns dict(phiqe [Field FOURIER], AL, Nphi) (planes as numpy)."""
    proj, h = ds.proj, ds.host
    which = which or ("TT" if ds.P == 1 else "EB")
    off = {1: {"T": 0}, 2: {"E": 0, "B": 1}, 3: {"T": 0, "E": 3, "B": 4}}[ds.P]          # plane index (BlockDiagIEB: TT,TE,ET,EE,BB)
    dof = {1: {"T": 0}, 2: {"E": 0, "B": 1}, 3: {"T": 0, "E": 1, "B": 2}}[ds.P]          # data component index
    plane = lambda op, k: np.asarray(op.p[off[k]], float)
    Bsz = ds.d.arr.shape[0]

    def fld(a, batch=False):                          # numpy plane or (B,1,..) array -> complex device tensor (B or 1,1,Nx,Nyh)
        a = np.asarray(a)
        a = a[None, None] if a.ndim == 2 else a
        return proj.tensor(a.astype(np.complex128))

    def filt(k, extra=1.0):                           # extra * (Σtot \ (TF * d[k]))   host-side diagonal algebra, device data
        TFk = plane(h["Mf"], k) * plane(h["B"], k)
        S = TFk ** 2 * plane(h["Cftilde"], k) + plane(h["Cn"], k)
        with np.errstate(divide="ignore", invalid="ignore"):
            w = TFk / S * extra
        w[~np.isfinite(w)] = 0
        dk = ds.d.arr[:, dof[k]:dof[k] + 1].contiguous()
        return proj.diag_apply(w[None], dk, FOURIER, FOURIER)           # DiagOp * field on the device (cmbl_diag_apply)

    with np.errstate(divide="ignore", invalid="ignore"):
        W = {}
        for k in off:
            TFk = plane(h["Mf"], k) * plane(h["B"], k)
            S = TFk ** 2 * plane(h["Cftilde"], k) + plane(h["Cn"], k)
            iS = 1 / S
            iS[~np.isfinite(iS)] = 0
            C = plane(h["Cf"], k)
            W[k] = (fld(TFk ** 2 * iS), fld(TFk ** 2 * C * iS), fld(TFk ** 2 * C ** 2 * iS))   # orders 0, 1, 2 in C
    L = _Legs(proj)
    mul = lambda a, b, s=1.0, out=None: proj.map_fma(a, b, s, out)

    if which == "TT":
        a, b = filt("T"), filt("T", plane(h["Cf"], "T"))
        un = None
        for i in (1, 2):
            t = proj.fourier_lmul(mul(L(a), L(b, (i,))), int(i == 1), int(i == 2))
            un = proj.axpby(-1.0, t, basis=FOURIER) if un is None else proj.axpby(1.0, un, -1.0, t, basis=FOURIER)
        w0, w1, w2 = W["T"]
        def A(i, j):
            acc = mul(L(w2, (i,), (j,)), L(w0))
            return mul(L(w1, (i,)), L(w1, (j,)), 1.0, acc)
    elif which == "EE":
        a1, a2 = filt("E", plane(h["Cf"], "E")), filt("E")
        un = None
        for i in (1, 2):
            acc = None
            for (j, k) in _inds(2):
                acc = mul(L(a1, (i,), j, k), L(a2, j, k), -2.0, acc)
            acc = mul(L(a1, (i,)), L(a2), 1.0, acc)
            t = proj.fourier_lmul(acc, int(i == 1), int(i == 2))
            un = t if un is None else proj.axpby(1.0, un, 1.0, t, basis=FOURIER)
        w0, w1, w2 = W["E"]
        def A(i, j):
            acc = None
            for (k, l, m, n, p, q) in _inds(6):
                e = _eps3(m, p) * _eps3(n, q)
                if e:
                    acc = mul(L(w2, (i,), (j,), k, l, m, n), L(w0, k, l, p, q), -4.0 * e, acc)
                    acc = mul(L(w1, (i,), k, l, m, n), L(w1, (j,), k, l, p, q), -4.0 * e, acc)
            acc = mul(L(w2, (i,), (j,)), L(w0), 1.0, acc)
            return mul(L(w1, (i,)), L(w1, (j,)), 1.0, acc)
    elif which == "EB":
        e1, b2 = filt("E"), filt("B")
        ce1, cb2 = filt("E", plane(h["Cf"], "E")), filt("B", plane(h["Cf"], "B"))
        un = None
        for i in (1, 2):
            acc = None
            for (j, k, l) in _inds(3):
                e = _eps3(k, l)
                if e:
                    acc = mul(L(ce1, (i,), j, k), L(b2, j, l), 2.0 * e, acc)
                    acc = mul(L(e1, j, k), L(cb2, (i,), j, l), -2.0 * e, acc)
            t = proj.fourier_lmul(acc, int(i == 1), int(i == 2))
            un = t if un is None else proj.axpby(1.0, un, 1.0, t, basis=FOURIER)
        (E0, E1, E2), (B0, B1, B2) = W["E"], W["B"]
        def A(i, j):
            acc = None
            for (k, l, m, n, p, q) in _inds(6):
                e = _eps3(m, p) * _eps3(n, q)
                if e:
                    acc = mul(L(E2, (i,), (j,), k, l, m, n), L(B0, k, l, p, q), 4.0 * e, acc)
                    acc = mul(L(E1, (i,), k, l, m, n), L(B1, (j,), k, l, p, q), -8.0 * e, acc)
                    acc = mul(L(E0, k, l, m, n), L(B2, (i,), (j,), k, l, p, q), 4.0 * e, acc)
            return acc
    else:
        raise ValueError(f"which={which!r} not implemented")          # src/quadratic_estimate.jl:41

    if AL is None:
        tot = None
        for (i, j) in _inds(2):
            t = proj.fourier_lmul(A(i, j), int(i == 1) + int(j == 1), int(i == 2) + int(j == 2), take_abs=True)
            tot = t if tot is None else proj.axpby(1.0, tot, 1.0, t, basis=FOURIER)
        tot = tot[0, 0].real.double().cpu().numpy()
        with np.errstate(divide="ignore"):
            AL = 1 / tot
        AL[~np.isfinite(AL)] = 0
    Cphi = np.asarray(h["Cphi"], float)
    wf = AL.copy()
    if wiener_filtered:
        with np.errstate(divide="ignore", invalid="ignore"):
            g = Cphi / (Cphi + AL)
        g[~np.isfinite(g)] = 0
        wf = g * AL
    phiqe = proj.diag_apply(wf[None], un, FOURIER, FOURIER)
    return dict(phiqe=Field(proj, phiqe, FOURIER), AL=AL, Nphi=AL.copy())


def quadratic_estimate_native(ds, which=None, wiener_filtered=True, AL=None):
    """`quadratic_estimate` as ONE library call (`cmbl_quadratic_estimate`): the same legs and sums inside the library.  Returns the same dict.
    The double-precision planes the call takes (Cf, Cf~, Cn, TF = Mf .* B, Cϕ) are kept on the device between calls -- they change only when
    the dataset's operators do (`theta.set_theta` replaces the host arrays, which drops the cached copies) -- so a call moves no plane over PCIe
    but the normalisation it returns."""
    import ctypes
    from .lib import check
    proj, h = ds.proj, ds.host
    which = which or ("TT" if ds.P == 1 else "EB")
    off = {1: {"T": 0}, 2: {"E": 0, "B": 1}, 3: {"T": 0, "E": 3, "B": 4}}[ds.P]
    comps = {"TT": ["T"], "EE": ["E"], "EB": ["E", "B"]}[which]
    # key: the identity of the host operators AND a content fingerprint (a strided sum of ~1000 entries each) -- an in-place edit of a host
    # plane (`h["Cn"].p[...] *= a`) keeps the ids; `ds.set_op` drops the cache as well
    def fp(op):
        a = np.asarray(op.p if hasattr(op, "p") else op).ravel()
        return float(np.abs(a[:: max(1, a.size // 1009)]).sum())
    key = (which,) + tuple((id(h[k]), fp(h[k])) for k in ("Cf", "Cftilde", "Cn", "Mf", "B", "Cphi"))
    cache = ds.__dict__.setdefault("_qe_planes", {})
    if key not in cache:
        cache.clear()
        plane = lambda op, k: np.ascontiguousarray(np.asarray(op.p[off[k]], np.float64))
        pack = lambda f: torch.as_tensor(np.ascontiguousarray(np.stack([f(k) for k in comps])), device=proj.device)
        cache[key] = dict(Cf=pack(lambda k: plane(h["Cf"], k)), Cft=pack(lambda k: plane(h["Cftilde"], k)), Cn=pack(lambda k: plane(h["Cn"], k)),
                          TF=pack(lambda k: plane(h["Mf"], k) * plane(h["B"], k)),
                          Cphi=torch.as_tensor(np.ascontiguousarray(np.asarray(h["Cphi"], np.float64)), device=proj.device),
                          keep=[h[k] for k in ("Cf", "Cftilde", "Cn", "Mf", "B", "Cphi")])       # the ids stay unique while these live
    pl = cache[key]
    B = ds.d.arr.shape[0]
    out = proj.empty(FOURIER, 1, B)
    ALo = torch.empty_like(pl["Cphi"])
    dp = lambda t: None if t is None else ctypes.cast(ctypes.c_void_p(t.data_ptr()), ctypes.POINTER(ctypes.c_double))
    ALi = None if AL is None else torch.as_tensor(np.ascontiguousarray(np.asarray(AL, np.float64)), device=proj.device)
    check(ds.lib.cmbl_quadratic_estimate(ds._h, {"TT": 0, "EE": 1, "EB": 2}[which], dp(pl["Cf"]), dp(pl["Cft"]), dp(pl["Cn"]), dp(pl["TF"]), dp(pl["Cphi"]),
                                         1 if wiener_filtered else 0, dp(ALi), ctypes.c_void_p(out.data_ptr()), dp(ALo), B))
    ALh = ALo.cpu().numpy()
    return dict(phiqe=Field(proj, out, FOURIER), AL=ALh, Nphi=ALh.copy())


# ---------------------------------------------------------------------------------------------------------------------
def MAP_joint_step(ds, phi, fstart=None, alpha_prev=1.0, alpha_tol=1e-4, alpha_max=None, cg_tol=1e-1, cg_nsteps=500, alias_quirk=None):
    """One iteration of the `MAP_joint` loop body (src/maximization.jl:160-206) at fiducial θ with G = I (:146).
    The line search runs in the field precision T like the reference's `optimize(T(0), T(αmax), Brent(); abs_tol=T(αtol))`:
    Brent's relative tolerance is sqrt(eps(T)) and a NaN logpdf is replaced by (α/αmax)·prevfloat(T(Inf)) (:194-199)."""
    proj, h = ds.proj, ds.host
    fin = np.finfo(np.float32 if proj.T == torch.float32 else np.float64)
    Ginv_saved = ds.ops["G_inv"]
    ds.set_op("G_inv", np.ones_like(h["Cphi"])[None])
    try:
        f, hist = ds.argmaxf_logpdf(phi, fstart=fstart, tol=cg_tol, nsteps=cg_nsteps)              # :164-169
        fo, po = ds.mix(f, phi, G=np.ones_like(h["Cphi"]))                                         # :176
        lp0, gfo, gpo = ds.gradient_logpdf_mixed(fo, po, alias_quirk=alias_quirk)                  # :178
        with np.errstate(divide="ignore"):
            Hinv = 1 / (_pinv(h["Cphi"]) + _pinv(h["Nphi"]))                                       # dataset.jl:134-137
        Hinv[~np.isfinite(Hinv)] = 0
        dphi = Field(proj, proj.diag_apply(Hinv[None], gpo.arr, FOURIER, FOURIER), FOURIER)        # :188
        amax = 2 * alpha_prev if alpha_max is None else alpha_max                                  # :193
        def neg(a):
            v = -float(np.sum(ds.logpdf_mixed(fo, proj.axpby(1.0, po, a, dphi))))
            return (a / amax) * float(fin.max) if np.isnan(v) else v                               # :198
        alpha, _, nls = brent_minimize(neg, 0.0, amax, abs_tol=alpha_tol, rel_tol=float(np.sqrt(fin.eps)))   # :194-199
        po2 = proj.axpby(1.0, po, alpha, dphi)                                                     # :201
        lp = ds.logpdf_mixed(fo, po2)                                                              # :205
        f2, phi2 = ds.unmix(fo, po2, G=np.ones_like(h["Cphi"]))                                    # :206
        return dict(f=f, phi=phi2, f_mixed=fo, phi_mixed=po2, grad_phi=gpo, dphi=dphi, alpha=alpha, logpdf=lp, logpdf_before=lp0,
                    cg_hist=hist, linesearch_evals=nls, dphi_norm=float(np.sqrt(np.sum(dphi.dot(dphi)))))
    finally:
        ds.set_op("G_inv", Ginv_saved)


def MAP_joint(ds, nsteps=20, phi_start=None, fstart=None, **kw):
    """`MAP_joint(ds; nsteps, fstart)` (src/maximization.jl:116-233) at the dataset's current θ: returns (f, ϕ, history)."""
    proj = ds.proj
    B = ds.d.arr.shape[0]
    phi = Field(proj, torch.zeros_like(proj.empty(FOURIER, 1, B)), FOURIER) if phi_start is None else phi_start
    f, alpha, hist = fstart, 1.0, []
    for _ in range(nsteps):
        st = MAP_joint_step(ds, phi, fstart=f, alpha_prev=alpha, **kw)
        f, phi, alpha = st["f"], st["phi"], st["alpha"]
        hist.append(dict(logpdf=st["logpdf"], alpha=alpha, ncg=len(st["cg_hist"]), dphi_norm=st["dphi_norm"],
                         linesearch_evals=st["linesearch_evals"]))
    return f, phi, hist


# ---------------------------------------------------------------------------------------------------------------------
def simulate_data(ds, phi, white_f, white_n):
    """`simulate(rng, ds; ϕ).d` (src/dataset.jl:59-66 run as a simulation): f ~ 𝒩(0,Cf), n ~ 𝒩(0,Cn), d = M·B·L(ϕ)·f + n.
    white_*: (B,P,Nx,Ny) unit white-noise maps (device tensors from ProjLambert.randn, or host arrays).  Returns d (HARMONIC)."""
    proj, h = ds.proj, ds.host
    raw = lambda w, op: Field(proj, proj.diag_apply(op.sqrt().p, proj.rfft(proj.tensor(w)), HARMONIC, HARMONIC), HARMONIC)
    return ds.mean(raw(white_f, h["Cf"]), phi) + raw(white_n, h["Cn"])


def MAP_marg(ds, nsteps=10, nsteps_with_meanfield_update=4, alpha=0.2, Nsims=50, sims_per_batch=1, phi_start=None, cg_tol=1e-1,
             cg_nsteps=500, base_seed=0, rng="device", whites=None, dist=None, progress=None, alias_quirk=None):
    """`MAP_marg(ds)` (src/maximization.jl:245-343) at fiducial θ: ϕ ← ϕ + α·Hϕ⁻¹·(g_data − ḡ_sims − Cϕ⁻¹ϕ) with
    g = ∂logpdf/∂ϕ at the Wiener-filtered f, the mean field ḡ averaged over Nsims simulated data sets (re-drawn from the SAME
    random numbers every step, `_rng = copy(rng)` :283) and refreshed during the first `nsteps_with_meanfield_update` steps.
    Sims are independent: `sims_per_batch` of them share a launch as batch slots (1 = the reference's per-sim CG stopping),
    and with `dist` (torch.distributed) sim i lives on rank i mod world, the mean field is one all_reduce (the `pmap` + `mean`
    of :304-311).  Sim i draws from generator base_seed + i (device Philox, or host PCG64, or injected `whites[kind][i]`).
    Returns (ϕ, trace)."""
    from . import rng as R
    from .chains import allreduce_sum
    proj, P, h = ds.proj, ds.P, ds.host
    assert ds.d.arr.shape[0] == 1, "MAP_marg for batched fields not implemented (src/maximization.jl:262)"
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist is not None and dist.is_initialized() else (1, 0)
    mine = [i for i in range(Nsims) if i % world == rank]
    with np.errstate(divide="ignore"):
        Hinv = 1 / (_pinv(h["Cphi"]) + _pinv(h["Nphi"]))                                          # :268
    Hinv[~np.isfinite(Hinv)] = 0
    phi = Field(proj, torch.zeros_like(proj.empty(FOURIER, 1, 1)), FOURIER) if phi_start is None else phi_start
    batches = [mine[i:i + sims_per_batch] for i in range(0, len(mine), sims_per_batch)]
    assert nsteps_with_meanfield_update >= 1, "the mean field must be estimated at least once"
    f_prev_sims, f_prev, gbar, trace = [None] * len(batches), None, None, []

    def draw(kind, ids):
        if whites is not None:
            return np.stack([whites[kind][i] for i in ids])
        if rng == "device":
            return proj.randn([base_seed + i for i in ids], R.stream_id(kind, 0), P)
        return np.stack([np.random.Generator(np.random.PCG64([base_seed + i, kind])).standard_normal((P, proj.Nx, proj.Ny)) for i in ids])

    def gMAP(d, fprev):                                                                           # :287-303
        f_wf, hist = ds.argmaxf_logpdf(phi, d=d, fstart=fprev, tol=cg_tol, nsteps=cg_nsteps)
        return ds.gradientphi_logpdf(f_wf, phi, d=d, alias_quirk=alias_quirk), f_wf, hist

    for step in range(1, nsteps + 1):
        g_data, f_prev, hist = gMAP(ds.d, f_prev)
        ncg = [len(hist)]
        if step <= nsteps_with_meanfield_update:                                                  # :305-316
            tot = torch.zeros_like(proj.empty(FOURIER, 1, 1))
            for k, ids in enumerate(batches):
                d_sim = simulate_data(ds, phi, draw(R.STREAM_F, ids), draw(R.STREAM_N, ids))      # :282-286
                g, f_prev_sims[k], hist = gMAP(d_sim, f_prev_sims[k])
                tot += g.arr.sum(dim=0, keepdim=True)
                ncg.append(len(hist))
            gbar = Field(proj, allreduce_sum(tot, dist) / Nsims, FOURIER)
        # final total posterior gradient, including gradient of the prior (:319).  g_data and every sim gradient carry the −Cϕ⁻¹ϕ of
        # the joint logpdf at the ϕ they were evaluated at, exactly as in the reference (ḡ is frozen after the mean-field steps).
        prior = Field(proj, proj.diag_apply(ds.ops["Cphi_inv"], phi.to(FOURIER).arr, FOURIER, FOURIER), FOURIER)
        g = g_data - gbar - prior
        phi = proj.axpby(1.0, phi.to(FOURIER), alpha, Field(proj, proj.diag_apply(Hinv[None], g.arr, FOURIER, FOURIER), FOURIER))   # :322
        trace.append(dict(step=step, g_norm=float(np.sqrt(np.sum(g.dot(g)))), ncg=ncg, phi=phi))
        if progress:
            progress(step, trace[-1])
    return phi, trace


def _pinv(x):
    with np.errstate(divide="ignore", invalid="ignore"):
        r = 1.0 / np.asarray(x, float)
    r[~np.isfinite(r)] = 0
    return r


# ---------------------------------------------------------------------------------------------------------------------
def mass_matrix_phi(ds):
    """src/sampling.jl:422-425"""
    h = ds.host
    return _pinv(h["G"]) ** 2 * (_pinv(h["Cphi"]) + _pinv(h["Nphi"]))


def symplectic_integrate(proj, x0, p0, Lam, U, dUdx, N=50, eps=0.1):
    """src/sampling.jl:14-46 on FOURIER Fields; Λ a real plane.  Returns (ΔH, x, p)."""
    Li = _pinv(Lam)[None]
    Linv = lambda p: Field(proj, proj.diag_apply(Li, p.arr, FOURIER, FOURIER), FOURIER)
    H = lambda x, p: U(x) - p.dot(Linv(p)) / 2
    x, p = x0, p0
    g = dUdx(x)
    for _ in range(N):
        x1 = proj.axpby(1.0, x, -eps, Linv(proj.axpby(1.0, p, -eps / 2, g)))
        g1 = dUdx(x1)
        p = proj.axpby(1.0, p, -eps / 2, proj.axpby(1.0, g1, 1.0, g))
        x, g = x1, g1
    return H(x, p) - H(x0, p0), x, p


def hmc_step(ds, fo, po, white_p, log_u, N=25, eps=0.01, always_accept=False, alias_quirk=None):
    """`hmc_step` (src/sampling.jl:405-418) over ϕ° with U = logpdf(Mixed(ds)); white_p / log_u are the injected draws."""
    proj = ds.proj
    Lam = mass_matrix_phi(ds)
    p0 = Field(proj, proj.diag_apply(np.sqrt(Lam)[None], proj.rfft(proj.tensor(white_p)), FOURIER, FOURIER), FOURIER)
    U = lambda x: ds.logpdf_mixed(fo, x)
    dU = lambda x: ds.gradient_logpdf_mixed(fo, x, alias_quirk=alias_quirk)[2]
    dH, xt, _ = symplectic_integrate(proj, po, p0, Lam, U, dU, N=N, eps=eps)
    accept = np.logical_or(always_accept, np.asarray(log_u) < dH)                 # a NaN ΔH (diverged trajectory) compares false: rejected
    x = proj.axpby(accept.astype(float), xt, 1.0 - accept.astype(float), po)      # x = accept*xtest + (1-accept)*x   (:415)
    return x, dH, accept


def hmc_step_native(ds, fo, po, white_p=None, log_u=None, seeds=None, step=0, N=25, eps=0.01, always_accept=False, alias_quirk=None):
    """The same update as `hmc_step`, as ONE library call (`cmbl_hmc_step`, include/cmblens.h): what a host that is neither Julia nor
    Python calls.  white_p / log_u None: drawn inside the library from `seeds` (one key per batch slot) and `step`, with the stream ids
    `sample_joint(rng="device")` uses.  Returns (ϕ°, ΔH, accept)."""
    import ctypes
    from .lib import check
    proj = ds.proj
    fo, po = fo.to(MAP), po.to(FOURIER)
    B = fo.arr.shape[0]
    alias_quirk = ds.alias_quirk if alias_quirk is None else alias_quirk
    mass = proj.tensor(mass_matrix_phi(ds)[None, None]).contiguous()
    wp = None if white_p is None else proj.tensor(white_p).contiguous()
    lu = None if log_u is None else (ctypes.c_double * B)(*[float(x) for x in np.atleast_1d(log_u)])
    sd = None if seeds is None else (ctypes.c_uint64 * B)(*[int(s) for s in seeds])
    out = proj.empty(FOURIER, 1, B)
    dH, acc = (ctypes.c_double * B)(), (ctypes.c_int * B)()
    ds.L.invalidate()
    ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    check(ds.lib.cmbl_hmc_step(ds._h, ds.L._h, ptr(fo.arr), ptr(po.arr), ptr(mass), ptr(wp), lu, sd, int(step), int(N), float(eps),
                               1 if always_accept else 0, 1 if alias_quirk else 0, B, ptr(out), dH, acc))
    return Field(proj, out, FOURIER), np.array(dH[:]), np.array(acc[:], bool)


def MAP_joint_step_native(ds, phi, fstart=None, alpha_prev=1.0, alpha_tol=1e-4, alpha_max=None, cg_tol=1e-1, cg_nsteps=500, alias_quirk=None):
    """The loop body of `MAP_joint_step` as ONE library call (`cmbl_map_joint_step`).  Returns dict(f, phi, alpha, logpdf, ncg, linesearch_evals)."""
    import ctypes
    from .lib import check
    proj, h = ds.proj, ds.host
    phi = phi.to(FOURIER)
    B = ds.d.arr.shape[0]
    alias_quirk = ds.alias_quirk if alias_quirk is None else alias_quirk
    with np.errstate(divide="ignore"):
        Hinv = 1 / (_pinv(h["Cphi"]) + _pinv(h["Nphi"]))                                       # dataset.jl:134-137
    Hinv[~np.isfinite(Hinv)] = 0
    hinv = proj.tensor(Hinv[None, None]).contiguous()
    fs = None if fstart is None else fstart.to(HARMONIC).arr.contiguous()
    f_out, phi_out = proj.empty(HARMONIC, ds.P, B), proj.empty(FOURIER, 1, B)
    lp, alpha, ncg, nls = (ctypes.c_double * B)(), ctypes.c_double(0), ctypes.c_int(0), ctypes.c_int(0)
    amax = 2 * alpha_prev if alpha_max is None else alpha_max
    ds.L.invalidate()
    ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    check(ds.lib.cmbl_map_joint_step(ds._h, ds.L._h, ptr(phi.arr), ptr(fs), ptr(hinv), float(amax), float(alpha_tol), float(cg_tol), int(cg_nsteps),
                                     1 if alias_quirk else 0, B, ptr(f_out), ptr(phi_out), lp, ctypes.byref(alpha), ctypes.byref(ncg), ctypes.byref(nls)))
    return dict(f=Field(proj, f_out, HARMONIC), phi=Field(proj, phi_out, FOURIER), alpha=alpha.value, logpdf=np.array(lp[:]) - ds.logdet_mix,
                ncg=ncg.value, linesearch_evals=nls.value)


def sample_f(ds, phi, white_f, white_n, fstart=None, tol=1e-1, nsteps=500):
    """`sample_f` (src/maximization.jl:56-62)"""
    proj, h = ds.proj, ds.host
    raw = lambda w, op: Field(proj, proj.diag_apply(op.sqrt().p, proj.rfft(proj.tensor(w)), HARMONIC, HARMONIC), HARMONIC)
    fs, ns = raw(white_f, h["Cf"]), raw(white_n, h["Cn"])
    dsim = ds.mean(fs, phi) + ns
    df, hist = ds.argmaxf_logpdf(phi, d=ds.d - dsim, fstart=fstart, tol=tol, nsteps=nsteps)
    return fs + df, hist


def gibbs_step(ds, phi, white_f, white_n, white_p, log_u, N=25, eps=0.01, always_accept=False, theta_pass=None, alias_quirk=None):
    """One `sample_joint` step (src/sampling.jl:187-193, 388-464): f | ϕ,θ -> mix -> HMC ϕ° | f°,θ -> [θ | f°,ϕ°] -> unmix -> logpdf.
    `theta_pass(fo, po)` runs the Gibbs θ passes in the mixed space and leaves the dataset at the new θ (theta.py)."""
    f, hist = sample_f(ds, phi, white_f, white_n)
    fo, po = ds.mix(f, phi)
    po2, dH, accept = hmc_step(ds, fo, po, white_p, log_u, N=N, eps=eps, always_accept=always_accept, alias_quirk=alias_quirk)
    if theta_pass is not None:
        theta_pass(fo, po2)
    f2, phi2 = ds.unmix(fo, po2)
    lp = ds.logpdf(f2, phi2)
    return dict(f=f2, phi=phi2, dH=dH, accept=accept, logpdf=lp, cg_hist=hist)


# ---------------------------------------------------------------------------------------------------------------------
def sample_joint(ds, nsamps_per_chain, chain_ids=(0,), base_seed=0, phi_start="prior", N=25, eps=0.01, nburnin_always_accept=10,
                 dist=None, nchains_total=None, progress=None, rng="host", first_step=0, filename=None, nfilewrite=5, nsavemaps=1,
                 resume=None, theta_ranges=None, theta_start="prior", alias_quirk=None):
    """`sample_joint` at fixed θ (src/sampling.jl:180-335): Gibbs loop  f | ϕ  ->  mix  ->  HMC ϕ° | f°  ->  unmix  ->  logpdf.
    The chains owned by this process are the batch slots of `ds` (`ds.d` must have len(chain_ids) slots; the reference runs
    chains under pmap, one worker per GPU, src/sampling.jl:266,292).  Chain c draws from its own generator keyed by
    base_seed + c, so results do not depend on how chains are partitioned over ranks: rng="host" NumPy PCG64 maps uploaded every
    step; rng="device" Philox4x32-10 on the GPU (cmbl_randn; sequence = (draw kind, step index), so a run resumed at
    `first_step` continues the same streams).  With `dist` (torch.distributed) the per-step scalars of all chains are
    all-gathered (RCCL over xGMI on GPUs) -- the only communication.
    `filename` (".zip"): every `nfilewrite` steps the samples since the last write are appended as a new chunk (scalars every
    step, ϕ and f maps every `nsavemaps` steps and at the end of each chunk; src/sampling.jl:226-228,311-320), gathered to rank
    0 which owns the file; `resume=True` continues from the file's last sample up to `nsamps_per_chain` (:247-256), an existing
    file needs an explicit `resume` (:239-241).
    `theta_ranges` = {"Aphi": grid, "r": grid} adds the Gibbs θ passes (`gibbs_sample_slice_θ!`, :427-437) after the HMC pass; the
    dataset's ParamDependentOps are then re-evaluated at the sampled θ for the following passes (single chain per dataset).  The
    sampled θ is part of every saved sample (`theta_<key>`, like `filter_for_saving` keeps θ, :226-228) and a resumed run continues
    from the file's last θ (:247-256).
    Defaults follow the reference (:190-214): `phi_start="prior"` draws ϕ ~ 𝒩(0, Cϕ) (0 / None = zero, or a Field), `theta_start=
    "prior"` draws θ uniformly inside its range (or a dict), `nburnin_always_accept=10`.  Step numbering: the reference stores the
    initial state as step 1 and its first Gibbs pass is step 2 (:268,277); here the first Gibbs pass is saved as step 1, so
    `always_accept = (step < nburnin_always_accept)` (:400) is evaluated with the reference's step = ours + 1.
    Returns dict(logpdf, dH, accept [nsamps, nchains], phi, f [, theta])."""
    from .chains import gather_chain_values, chain_seed
    from . import rng as R
    from . import chainfile as CF
    proj, P = ds.proj, ds.P
    B = len(chain_ids)
    assert ds.d.arr.shape[0] == B, "dataset batch size must equal the number of local chains"
    assert rng in ("host", "device")
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if multi else 0
    assert not multi or nchains_total is not None, "nchains_total (chains over all ranks) is required when running on more than one rank"
    ntot = nchains_total if multi else B          # file / gather order: global chain id across ranks, position in a single process
    fidx = list(chain_ids) if multi else list(range(B))
    gdev = proj.device if multi and dist.get_backend() == "nccl" else "cpu"
    # resume: True / False / None as in the reference (:236-256), or the path of ANOTHER chain file (any readable format, e.g. a
    # `.jld2` the Julia package wrote, which is read-only here) whose last state starts a new file `filename`
    resume_src = resume if isinstance(resume, str) else None
    CF.check_filename(filename, False if resume_src else resume)
    chunk_index, clobber, theta_resume = 1, True, None
    if resume_src is not None or (filename is not None and resume and os.path.isfile(filename)):
        chunk_index, first_step, last = CF.last_state(resume_src or filename)
        clobber = resume_src is not None
        if resume_src is not None:
            chunk_index = 1
        phi_start = Field(proj, proj.tensor(np.stack([last[c]["phi"] for c in fidx])[:, None]), FOURIER)
        theta_resume = {k[6:]: float(v) for k, v in last[fidx[0]].items() if k.startswith("theta_")}
    seeds = [chain_seed(base_seed, c) for c in chain_ids]
    rngs = [np.random.Generator(np.random.PCG64(s if first_step == 0 else [s, first_step])) for s in seeds]
    hist = dict(logpdf=[], dH=[], accept=[], ncg=[])
    chunk = [[] for _ in range(B)]

    def flush():
        nonlocal chunk_index, clobber, chunk
        nsamp = len(chunk[0])
        if nsamp == 0:
            return
        gath = lambda a: gather_chain_values(fidx, a, ntot, dist if multi else None, gdev)
        skeys = ["step", "logpdf", "dH", "accept", "ncg"] + sorted(k for k in chunk[0][0] if k.startswith("theta_"))
        sc = {k: gath(np.array([[s[k] for s in ch] for ch in chunk], float).reshape(B, nsamp)) for k in skeys}
        has = [i for i, s in enumerate(chunk[0]) if "phi" in s]
        mp = {}
        for k in ("phi", "f"):
            a = np.stack([np.stack([np.asarray(ch[i][k], np.complex128) for i in has]) for ch in chunk])      # (B, nmaps, ...)
            g = gath(a.view(np.float64))
            mp[k] = g.view(np.complex128)
        if rank == 0:
            out = []
            for c in range(ntot):
                samples = []
                for i in range(nsamp):
                    smp = {k: sc[k][c, i] for k in sc}
                    smp["step"] = int(smp["step"])
                    if i in has:
                        smp.update({k: mp[k][c, has.index(i)] for k in mp})
                    samples.append(smp)
                out.append(samples)
            CF.write_chunk(filename, chunk_index, out, rundat=dict(nchains=ntot, base_seed=base_seed, N=N, eps=eps, rng=rng, nsavemaps=nsavemaps,
                                                                  nfilewrite=nfilewrite, Ny=proj.Ny, Nx=proj.Nx, theta_pix=proj.theta_pix, npol=P),
                           clobber=clobber)
        chunk_index, clobber, chunk = chunk_index + 1, False, [[] for _ in range(B)]

    theta, theta_hist = None, []
    if theta_ranges:
        from . import theta as TH
        assert B == 1, "θ sampling: one chain per dataset (the operators of a dataset carry a single θ)"
        theta = dict(r=None, Aphi=None)
        if theta_resume:                                            # resumed: the file's last θ (merge!(states, last(chunk)), :252)
            theta.update(theta_resume)
        elif isinstance(theta_start, str):
            assert theta_start == "prior", theta_start              # gibbs_initialize_θ! (:340-354): uniform inside the range
            for j, (key, xs) in enumerate(theta_ranges.items()):
                u0 = R.uniform(seeds[0], R.stream_id(R.STREAM_INIT, 1 + j))[0] if rng == "device" else rngs[0].random()
                theta[key] = float(xs[0] + u0 * (xs[-1] - xs[0]))
        else:
            theta.update(theta_start or {})
        TH.set_theta(ds, **theta)
    # ϕ is initialised AFTER θ like the reference's initializer list (:186-190): the prior draw uses Cϕ(θ) and the host generator
    # hands out its θ draw first
    if isinstance(phi_start, str):
        assert phi_start == "prior", phi_start                      # gibbs_initialize_ϕ! (:363-384): simulate(ds.Cϕ(θ)), after θ (:340-354)
        w0 = proj.randn(seeds, R.stream_id(R.STREAM_INIT, 0), 1) if rng == "device" else np.stack([r.standard_normal((1, proj.Nx, proj.Ny)) for r in rngs])
        phi = Field(proj, proj.diag_apply(np.sqrt(np.asarray(ds.host["Cphi"], float))[None], proj.rfft(proj.tensor(w0)), FOURIER, FOURIER), FOURIER)
    elif phi_start is None or (np.isscalar(phi_start) and phi_start == 0):
        phi = Field(proj, torch.zeros_like(proj.empty(FOURIER, 1, B)), FOURIER)
    else:
        phi = phi_start
    f = None
    for step in range(first_step, nsamps_per_chain) if (filename is not None and resume) else range(first_step, first_step + nsamps_per_chain):
        if rng == "device":
            wf, wn, wp = (proj.randn(seeds, R.stream_id(kind, step), Pp) for kind, Pp in ((R.STREAM_F, P), (R.STREAM_N, P), (R.STREAM_P, 1)))
            logu = np.log(np.array([R.uniform(s, R.stream_id(R.STREAM_U, step))[0] for s in seeds]))
        else:
            draw = lambda Pp: np.stack([r.standard_normal((Pp, proj.Nx, proj.Ny)) for r in rngs])
            wf, wn, wp = draw(P), draw(P), draw(1)
            logu = np.log(np.array([r.random() for r in rngs]))
        tpass = None
        if theta_ranges:
            def tpass(fo_, po_, step=step):
                for j, (key, xs) in enumerate(theta_ranges.items()):
                    uu = R.uniform(seeds[0], R.stream_id(R.STREAM_U + 1 + j, step)) if rng == "device" else [rngs[0].random()]
                    val, _ = TH.gibbs_sample_theta(ds, fo_, po_, theta, key, xs, uu)
                    theta[key] = float(val[0])
                TH.set_theta(ds, **theta)
        st = gibbs_step(ds, phi, wf, wn, wp, logu, N=N, eps=eps, always_accept=(step + 2 < nburnin_always_accept), theta_pass=tpass,
                        alias_quirk=alias_quirk)
        if theta_ranges:
            theta_hist.append(dict(theta))
        phi, f = st["phi"], st["f"]
        hist["logpdf"].append(st["logpdf"]); hist["dH"].append(st["dH"]); hist["accept"].append(st["accept"].astype(float))
        hist["ncg"].append(np.full(B, len(st["cg_hist"]), float))
        if filename is not None:
            endchunk = (step + 1) % nfilewrite == 0
            maps = step == 0 or (step + 1) % nsavemaps == 0 or endchunk
            if maps:
                ph, fh = phi.to(FOURIER).arr.cpu().numpy(), f.to(HARMONIC).arr.cpu().numpy()
            for b in range(B):
                smp = dict(step=step + 1, logpdf=st["logpdf"][b], dH=st["dH"][b], accept=float(st["accept"][b]), ncg=float(len(st["cg_hist"])))
                if theta_ranges:
                    smp.update({"theta_" + k: float(v) for k, v in theta.items() if v is not None})
                if maps:
                    smp.update(phi=ph[b, 0], f=fh[b])
                chunk[b].append(smp)
            if endchunk:
                flush()
        if progress:
            progress(step, st)
    out = {k: (np.stack(v) if v else np.zeros((0, B))) for k, v in hist.items()}                     # (nsamps, B)
    if multi:
        out = {k: gather_chain_values(fidx, v.T, ntot, dist, gdev).T for k, v in out.items()}   # (nsamps, nchains_total)
    out["phi"], out["f"] = phi, f
    if theta_ranges:
        out["theta"] = theta_hist
    return out
