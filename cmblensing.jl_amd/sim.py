"""`load_sim` and the host-side operator algebra that feeds the device dataset.

Mirrors src/dataset.jl:186-338 (load_sim), src/cls.jl:11-35,288-309 (Cℓs, noiseCℓs, beamCℓs),
src/specialops.jl:61-118,232-241 (BlockDiagIEB, BandPass), src/proj_lambert.jl:173-175,361-371
(Cℓ_to_2D / Cℓ_to_Cov) and src/field_vectors.jl:64-84 (2x2 sqrt/pinv with the reference's symmetric-matrix
assumption).  Like the reference's own GPU extension (ext/CMBLensingCUDAExt.jl:46-49) the ℓ-space operator
planes are built once on the host; everything applied per iteration lives on the device.
"""
import numpy as np
import torch

from .engine import ProjLambert, BaseDataSet, Field, MAP, FOURIER, HARMONIC


class Cls:
    """Tabulated spectrum with linear interpolation, NaN outside the table (src/cls.jl:11-29)."""

    def __init__(self, ell, cl):
        ell, cl = np.asarray(ell, float), np.asarray(cl, float)
        ok = ~np.isnan(cl)
        self.ell, self.cl = ell[ok], cl[ok]

    def __call__(self, l):
        l = np.asarray(l, float)
        out = np.interp(l, self.ell, self.cl)
        out[(l < self.ell[0]) | (l > self.ell[-1])] = np.nan
        return out


def _pinv(x):
    with np.errstate(divide="ignore", invalid="ignore"):
        r = 1.0 / x
    r[~np.isfinite(r)] = 0
    return r


def cl_to_2d(cl, proj, units=None, dtype=np.float64):
    """nan2zero(Cℓ(ℓmag)) / units on the half-plane grid [x, ky] (src/proj_lambert.jl:173-175,362-364)."""
    v = cl(proj.lmag)
    v[~np.isfinite(v)] = 0
    return (v / (proj.Opix if units is None else units)).astype(dtype)


def noise_cls(muK_arcmin_T=3.0, lknee=100.0, alphaknee=3.0, lmax=8000):
    """src/cls.jl:288-299 with beamFWHM = 0"""
    ell = np.arange(2, lmax + 1)
    base = np.deg2rad(muK_arcmin_T / 60) ** 2 * (1 + (lknee / ell) ** alphaknee)
    return {"TT": Cls(ell, base), "EE": Cls(ell, 2 * base), "BB": Cls(ell, 2 * base), "TE": Cls(ell, 0 * base)}


def beam_cls(beam_fwhm=0.0, lmax=8000):
    """src/cls.jl:307-309"""
    ell = np.arange(2, lmax + 1, dtype=float)
    return Cls(ell, np.exp(-ell ** 2 * np.deg2rad(beam_fwhm / 60) ** 2 / (8 * np.log(2))))


def lowpass(l, dl=50):
    """LowPass(ℓ; Δℓ) (src/specialops.jl:236-240)"""
    ramp_up = (np.cos(np.linspace(np.pi, 0, dl)) + 1) / 2
    return Cls(np.arange(l + 1), np.r_[np.ones(l - dl + 1), 1 - ramp_up])


def border_mask(proj, pad_deg=1.0, apod_deg=1.0):
    """Deterministic cosine-apodised border mask [x, y] (input data; the reference's make_mask,
    src/masking.jl:1-25, additionally punches random point-source holes)."""
    def prof(n):
        d = (np.minimum(np.arange(n), n - 1 - np.arange(n)) + 0.5) * proj.theta_pix / 60
        t = np.clip((d - pad_deg) / max(apod_deg, 1e-30), 0, 1)
        return (1 - np.cos(np.pi * t)) / 2
    return np.outer(prof(proj.Nx), prof(proj.Ny))


class HarmOp:
    """Host-side real operator, diagonal in ℓ in the harmonic basis: npol diagonal planes, or for npol = 3 the
    BlockDiagIEB planes (TT, TE, ET, EE, BB) (src/specialops.jl:61-118)."""

    def __init__(self, planes):
        self.p = np.asarray(planes, dtype=np.float64)
        self.block = self.p.shape[0] == 5

    @staticmethod
    def from_cls(pol, proj, cls, units=None, te_zero=False):
        get = (lambda k: cls[k]) if isinstance(cls, dict) else (lambda k: cls)
        c = lambda k: cl_to_2d(get(k), proj, units)
        if pol == "I":
            return HarmOp([c("TT")])
        if pol == "P":
            return HarmOp([c("EE"), c("BB")])
        tt = c("TT")
        te = np.zeros_like(tt) if te_zero else c("TE")
        return HarmOp([tt, te, te, c("EE"), c("BB")])

    def __add__(self, o):
        if np.isscalar(o):
            q = self.p.copy()
            q[[0, 3, 4] if self.block else slice(None)] += o
            return HarmOp(q)
        return HarmOp(self.p + o.p)

    def scale(self, s):
        return HarmOp(self.p * s)

    def __matmul__(self, o):
        if not self.block:
            return HarmOp(self.p * o.p)
        a, b, c, d, e = self.p
        A, B, C, D, E = o.p
        return HarmOp([a * A + b * C, a * B + b * D, c * A + d * C, c * B + d * D, e * E])

    def T(self):
        return HarmOp(self.p[[0, 2, 1, 3, 4]]) if self.block else self

    def pinv(self):
        if not self.block:
            return HarmOp(_pinv(self.p))
        a, _, c, d, e = self.p                      # reads A[2,1] for both off-diagonals (src/field_vectors.jl:80-84)
        idet = _pinv(a * d - c * c)
        return HarmOp([d * idet, -(c * idet), -(c * idet), a * idet, _pinv(e)])

    def sqrt(self):
        if not self.block:
            return HarmOp(np.sqrt(self.p))
        a, _, c, d, e = self.p                      # src/field_vectors.jl:68-73
        s = np.sqrt(a * d - c * c)
        t = _pinv(np.sqrt(a + (d + 2 * s)))
        return HarmOp([t * (a + s), t * c, t * c, t * (d + s), np.sqrt(e)])

    def logdet(self, proj):
        """Σ λ·log|d| skipping non-finite (src/proj_lambert.jl:331-336, src/specialops.jl:96)"""
        planes = [self.p[0] * self.p[3] - self.p[2] * self.p[2], self.p[4]] if self.block else list(self.p)
        tot = 0.0
        for d in planes:
            with np.errstate(divide="ignore"):
                v = np.log(np.abs(d)) * proj.lam[None, :]
            tot += v[np.isfinite(v)].sum()
        return tot


def white_noise(seed, shape):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal(shape)


def load_sim(theta_pix, Nside, pol, cls, T=torch.float32, device=0, muK_arcmin_T=3.0, lknee=100.0, alphaknee=3.0,
             beam_fwhm=0.0, pixel_mask=None, bandpass_lmax=3000, nsteps=7, Nbatch=1, seeds=(1, 2, 3),
             Nphi=None, Nphi_fac=2, G=None, rng="host"):
    # Nphi: None / "qe" -> N⁰ of the quadratic estimator like the reference; "flat" -> cheap flat level; or an [x,ky] plane
    """`load_sim` (src/dataset.jl:186-338).  `cls`: dict group -> dict {TT,EE,BB,TE,pp} of Cls for the groups
    'unlensed_scalar', 'tensor', 'total' (e.g. decoded from the reference's dat/default_camb_Cls.jld2).
    Returns dict(f, phi, ftilde, d, ds, proj) with Fields on the device."""
    Ny, Nx = (Nside, Nside) if np.isscalar(Nside) else Nside
    proj = ProjLambert(Ny, Nx, theta_pix, T, device)
    P = {"I": 1, "P": 2, "IP": 3}[pol]
    lmax = proj.lmax
    ncl = noise_cls(muK_arcmin_T, lknee, alphaknee, lmax)
    mk = lambda c, **kw: HarmOp.from_cls(pol, proj, c, **kw)

    Cphi = cl_to_2d(cls["total"]["pp"], proj)                                   # :267
    Cfs, Cten = mk(cls["unlensed_scalar"]), mk(cls["tensor"])                  # :268-269
    Cf = Cfs + Cten                                                             # :273 at r = r₀
    Cn = mk(ncl)                                                                # :271-272
    Mf = mk(lowpass(bandpass_lmax), units=1, te_zero=True)                      # :279
    bcl = beam_cls(beam_fwhm, lmax)
    Bop = mk(Cls(bcl.ell, np.sqrt(bcl.cl)), units=1, te_zero=True)              # :300
    Mpix = border_mask(proj, **pixel_mask) if pixel_mask is not None else None

    qe_nphi = Nphi is None or (isinstance(Nphi, str) and Nphi == "qe")
    if qe_nphi or (isinstance(Nphi, str) and Nphi == "flat"):                   # provisional flat level until the data exist (see below)
        sel = (proj.lmag > 100) & (proj.lmag < 2000)
        Nphi = np.where(Cphi > 0, np.exp(np.mean(np.log(Cphi[sel]))), 0.0)
    Nphi = np.asarray(Nphi, float) / Nphi_fac
    Gp = np.ones_like(Cphi) if G is None else np.asarray(G, float)              # G(θ) ≡ I at fiducial θ (:317-320)
    s2len = np.deg2rad(5 / 60) ** 2
    D = ((Cf + (Cn.scale(2) + s2len)) @ Cf.pinv()).sqrt()                       # :322-328
    precond = Cf.pinv() + (Bop.T() @ Mf.T() @ Cn.pinv() @ Mf @ Bop)             # src/dataset.jl:129-132
    logdet_sum = Cf.logdet(proj) + Cn.logdet(proj) + HarmOp([Cphi]).logdet(proj)
    ops = dict(Cf_inv=Cf.pinv().p, Cn_inv=Cn.pinv().p, B=Bop.p, Mf=Mf.p, D=D.p, D_inv=D.pinv().p,
               precond_inv=precond.pinv().p, Cphi_inv=_pinv(Cphi)[None], G_inv=_pinv(Gp)[None], Mpix=Mpix)
    ds = BaseDataSet(proj, P, ops, logdet_sum=logdet_sum, nsteps=nsteps)
    Cft = mk(cls["total"])                                                     # Cf̃ (:270)
    ds.host = dict(Cf=Cf, Cn=Cn, Cphi=Cphi, Mf=Mf, B=Bop, D=D, Nphi=Nphi, G=Gp, Mpix=Mpix, precond=precond, Cftilde=Cft,
                   Cfs=Cfs, Cten=Cten, r0=0.2, Aphi0=1.0, s2len=s2len)
    if G is not None:
        ds.host["G_user"] = Gp.copy()                                           # theta.set_theta keeps a user-supplied G at fiducial Aϕ           # θ layer (theta.py): r₀ = Cℓ.params.r, Aϕ₀ = 1 (:239,250)

    # simulate: x = sqrt(C)·rfft(white)   (src/specialops.jl:6), white ~ NumPy PCG64(seed) uploaded, or with rng="device"
    # drawn on the GPU (cmbl_randn: Philox4x32-10, batch slot b keyed by seed + 1000003*b)
    def sim(op_planes, seed, Pp):
        if rng == "device":
            w = proj.randn([seed + 1000003 * b for b in range(Nbatch)], 0, Pp)
        else:
            w = proj.tensor(white_noise(seed, (Nbatch, Pp, Nx, Ny)))
        return Field(proj, proj.diag_apply(op_planes, proj.rfft(w), HARMONIC, HARMONIC), HARMONIC)
    f = sim(Cf.sqrt().p, seeds[0], P)
    phi = sim(np.sqrt(Cphi)[None], seeds[1], 1)
    phi = Field(proj, phi.arr, FOURIER)
    n = sim(Cn.sqrt().p, seeds[2], P)
    ftilde = ds.L(phi) * f                                                       # map
    d = ds.mean(f, phi) + n                                                      # M·B·L(ϕ)·f + n  (src/dataset.jl:59-66)
    ds.set_data(d)
    if qe_nphi:                                                                 # ds.Nϕ = quadratic_estimate(ds).Nϕ / Nϕ_fac   (:316)
        from .drivers import quadratic_estimate
        ds.host["Nphi"] = quadratic_estimate(ds)["Nphi"] / Nphi_fac
    return dict(f=f, phi=phi, ftilde=ftilde, d=d, n=n, ds=ds, proj=proj)
