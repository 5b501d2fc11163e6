"""Data model, posterior, its hand-written gradient, the Wiener filter and `load_sim`.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
    src/dataset.jl:37-137,186-338     (BaseDataSet, model, gradientf_logpdf, mix/unmix,
                                       preconditioners, load_sim)
    src/distributions.jl:8-15         (Gaussian logpdf)
    src/specialops.jl:61-118          (BlockDiagIEB)
    src/field_vectors.jl:64-84        (2x2 sqrt / det / pinv incl. quirk Q2)
    src/maximization.jl:17-62         (argmaxf_logpdf, sample_f)
    src/autodiff.jl:105-133           (gradient conventions)
of /root/reference.

Basis names used here: 'map' (I / QU / IQU maps), 'qu' (their rfft2), 'harm' (Fourier /
EBFourier / IEBFourier -- the basis the covariances are diagonal in).
"""
import numpy as np
from .flatsky import (Proj, rfft2, irfft2, cl_to_2d, nan2zero, pinv, qu2eb, eb2qu, dot_fourier,
                      logdet_fourier, noise_cls, beam_cls, lowpass, load_camb, white_noise,
                      border_mask, Cls, ctype)
from .lenseflow import LenseFlow
from .cg import conjugate_gradient

__all__ = ["HarmOp", "DataSet", "load_sim", "to_harm", "from_harm", "harm_to_qu", "qu_to_harm"]


# ---------------------------------------------------------------------------------------
# basis changes for P = 1 (I), 2 (QU), 3 (IQU)      src/proj_lambert.jl:245-300

def qu_to_harm(proj, fl):
    P = fl.shape[-3]
    if P == 1:
        return fl
    if P == 2:
        return qu2eb(proj, fl)
    return np.concatenate([fl[..., :1, :, :], qu2eb(proj, fl[..., 1:, :, :])], axis=-3)


def harm_to_qu(proj, fh):
    P = fh.shape[-3]
    if P == 1:
        return fh
    if P == 2:
        return eb2qu(proj, fh)
    return np.concatenate([fh[..., :1, :, :], eb2qu(proj, fh[..., 1:, :, :])], axis=-3)


def to_harm(proj, fmap):
    return qu_to_harm(proj, rfft2(fmap))


def from_harm(proj, fh):
    return irfft2(harm_to_qu(proj, fh), proj.Ny).astype(proj.T)


# ---------------------------------------------------------------------------------------

def _sqrt2x2(a, b_unused, c, d):
    """src/field_vectors.jl:68-73 -- reads b = A[2,1] (quirk Q2)."""
    b = c
    s = np.sqrt(a * d - b * c)
    t = pinv(np.sqrt(a + (d + 2 * s)))
    return t * (a + s), t * b, t * c, t * (d + s)


def _pinv2x2(a, b_unused, c, d):
    """src/field_vectors.jl:80-84 -- reads b = A[2,1] (quirk Q2)."""
    b = c
    idet = pinv(a * d - b * c)
    return d * idet, -(b * idet), -(c * idet), a * idet


class HarmOp:
    """Real operator diagonal in ℓ in the harmonic basis.
    P=1: Diagonal(Fourier); P=2: Diagonal(EBFourier); P=3: BlockDiagIEB
    (src/specialops.jl:61-118): 2x2 block (a b; c d) on (I,E) plus e on B.  Arrays [x, ky]."""

    def __init__(self, P, diag=None, te=None, bb=None):
        self.P = P
        if P == 3:
            self.te = tuple(np.asarray(x) for x in te)      # (a, b, c, d)
            self.bb = np.asarray(bb)
        else:
            self.d = np.asarray(diag)                       # (P, Nx, Nyh)

    # -- construction helpers
    @staticmethod
    def from_cls(pol, proj, cls, scale=1.0, units=None, te_zero=False):
        """`Cℓ_to_Cov(pol, proj, Cℓ[k] for k in ks)` (src/proj_lambert.jl:361-371, dataset.jl:266-271).
        `cls` maps 'TT','EE','BB','TE' -> Cls (or a single Cls used for every auto-spectrum
        when it is not a dict, as for masks/beams dataset.jl:279,300)."""
        get = (lambda k: cls[k]) if isinstance(cls, dict) else (lambda k: cls)
        c2d = lambda k: cl_to_2d(get(k), proj, units) * proj.T(scale)
        if pol == "I":
            return HarmOp(1, diag=c2d("TT")[None])
        if pol == "P":
            return HarmOp(2, diag=np.stack([c2d("EE"), c2d("BB")]))
        tt, ee, bb = c2d("TT"), c2d("EE"), c2d("BB")
        te = np.zeros_like(tt) if te_zero else c2d("TE")
        return HarmOp(3, te=(tt, te, te.copy(), ee), bb=bb)

    def _like(self, *, diag=None, te=None, bb=None):
        return HarmOp(self.P, diag=diag, te=te, bb=bb)

    # -- algebra (src/specialops.jl:98-110)
    def __add__(self, o):
        if np.isscalar(o):                                   # + UniformScaling
            if self.P == 3:
                a, b, c, d = self.te
                return self._like(te=(a + o, b, c, d + o), bb=self.bb + o)
            return self._like(diag=self.d + o)
        if self.P == 3:
            return self._like(te=tuple(x + y for x, y in zip(self.te, o.te)), bb=self.bb + o.bb)
        return self._like(diag=self.d + o.d)

    __radd__ = __add__

    def scale(self, s):
        if self.P == 3:
            return self._like(te=tuple(x * s for x in self.te), bb=self.bb * s)
        return self._like(diag=self.d * s)

    def __matmul__(self, o):
        """La * Lb"""
        if self.P == 3:
            a, b, c, d = self.te
            A, B, C, D = o.te
            return self._like(te=(a * A + b * C, a * B + b * D, c * A + d * C, c * B + d * D), bb=self.bb * o.bb)
        return self._like(diag=self.d * o.d)

    def pinv(self):
        if self.P == 3:
            return self._like(te=_pinv2x2(*self.te), bb=pinv(self.bb))
        return self._like(diag=pinv(self.d))

    def sqrt(self):
        if self.P == 3:
            return self._like(te=_sqrt2x2(*self.te), bb=np.sqrt(self.bb))
        return self._like(diag=np.sqrt(self.d))

    def T_(self):
        """adjoint (real entries => transpose)"""
        if self.P == 3:
            a, b, c, d = self.te
            return self._like(te=(a, c, b, d), bb=self.bb)
        return self

    # -- application
    def __call__(self, fh):
        """`L * f` with f in the harmonic basis (B,P,Nx,Nyh)."""
        if self.P == 3:
            a, b, c, d = self.te
            I, E, Bm = fh[:, 0], fh[:, 1], fh[:, 2]
            return np.stack([a * I + b * E, c * I + d * E, self.bb * Bm], axis=1)
        return self.d * fh

    def solve(self, fh):
        """`L \\ f`: Diagonal -> nan2zero(diag .\\ f) (specialops.jl:10); BlockDiagIEB -> pinv(L)*f (:78)."""
        if self.P == 3:
            return self.pinv()(fh)
        with np.errstate(divide="ignore", invalid="ignore"):
            return nan2zero(fh / self.d)

    def logdet(self, proj):
        """src/proj_lambert.jl:331-336, specialops.jl:96 (logdet(det ΣTE) + logdet ΣB)."""
        if self.P == 3:
            a, b_, c, d = self.te
            b = c                                             # det reads A[2,1] twice (field_vectors.jl:75-78)
            return logdet_fourier(proj, (a * d - b * c)[None, None]) + logdet_fourier(proj, self.bb[None, None])
        return logdet_fourier(proj, self.d[None])

    def arrays(self):
        """Flat list of the real [x,ky] arrays that define the operator (for the C-ABI)."""
        if self.P == 3:
            return list(self.te) + [self.bb]
        return [self.d[i] for i in range(self.P)]


# ---------------------------------------------------------------------------------------

class DataSet:
    """`BaseDataSet` (src/dataset.jl:37-57) at fiducial θ.  All HarmOp; Mpix is a map mask
    (Nx,Ny) or None (= I); Cphi, G, Nphi are [x,ky] arrays; d is harmonic (B,P,Nx,Nyh)."""

    def __init__(self, proj, P, Cf, Cn, Cphi, Mf, B, Mpix=None, Cnhat=None, Mfhat=None, Bhat=None,
                 D=None, G=None, Nphi=None, d=None, nsteps=7, Cftilde=None):
        self.proj, self.P = proj, P
        self.Cf, self.Cn, self.Cphi, self.Mf, self.B, self.Mpix = Cf, Cn, Cphi, Mf, B, Mpix
        self.Cnhat = Cn if Cnhat is None else Cnhat
        self.Mfhat = Mf if Mfhat is None else Mfhat
        self.Bhat = B if Bhat is None else Bhat
        self.D, self.G, self.Nphi, self.d, self.nsteps, self.Cftilde = D, G, Nphi, d, nsteps, Cftilde
        self._L = None

    # lensing operator, re-cached only when ϕ changes (src/lenseflow.jl:123-129)
    def L(self, phi_l):
        key = phi_l.tobytes()
        if self._L is None or self._L[0] != key:
            self._L = (key, LenseFlow(self.proj, None, self.nsteps, phi_l=phi_l))      # Fourier ϕ used as is (lenseflow.jl:135)
        return self._L[1]

    def dot(self, a, b):
        return dot_fourier(self.proj, a, b)

    # M = Mfourier * Mpix ; M' = Mpix' * Mfourier'   (dataset.jl:279-285, specialops.jl:393)
    def M(self, fh):
        if self.Mpix is None:
            return self.Mf(fh)
        return self.Mf(to_harm(self.proj, self.Mpix * from_harm(self.proj, fh)))

    def Mt(self, fh):
        y = self.Mf.T_()(fh)
        if self.Mpix is None:
            return y
        return to_harm(self.proj, self.Mpix * from_harm(self.proj, y))

    # ---- model ------------------------------------------------------------------------
    def lense(self, L, fh):
        return to_harm(self.proj, L.apply(from_harm(self.proj, fh)))

    def mean(self, L, fh):
        """μ = M (B (Lϕ f))   (dataset.jl:59-66)"""
        return self.M(self.B(self.lense(L, fh)))

    def logpdf(self, fh, phi_l, d=None):
        """`logpdf(ds; f, ϕ)` = Σ Gaussian terms  −(z†Σ⁻¹z + logdet Σ)/2  (distributions.jl:11-15)."""
        d = self.d if d is None else d
        proj = self.proj
        L = self.L(phi_l)
        z = self.mean(L, fh) - d
        lp = -(self.dot(fh, self.Cf.solve(fh)) + self.Cf.logdet(proj)) / 2
        lp = lp - (dot_fourier(proj, phi_l, pinv(self.Cphi) * phi_l) + logdet_fourier(proj, self.Cphi[None, None])) / 2
        lp = lp - (self.dot(z, self.Cn.pinv()(z)) + self.Cn.logdet(proj)) / 2
        return lp

    def gradientf_logpdf(self, fh, L, d):
        """src/dataset.jl:76-80:  Lϕ'B'M'Cn⁻¹(d − M B Lϕ f) − Cf⁻¹ f   (harmonic in, harmonic out)."""
        proj = self.proj
        r = self.Cn.pinv()(d - self.mean(L, fh))
        y = self.B.T_()(self.Mt(r))
        g = qu_to_harm(proj, L.adj(harm_to_qu(proj, y)))
        return g - self.Cf.pinv()(fh)

    def gradientphi_logpdf(self, fh, phi_l, d=None, alias_quirk=False):
        """∂/∂ϕ logpdf(ds; f, ϕ, d) at fixed f -- `gradient(ϕ -> logpdf(dsθ; f=f_wf, ϕ, dsθ.d), ϕ)` of MAP_marg
        (maximization.jl:307): pullback of Lϕ*f (flowops.jl:40-54) of ∂/∂f̃ = −B'M'Cn⁻¹z, plus the prior term −Cϕ⁻¹ϕ."""
        d = self.d if d is None else d
        proj = self.proj
        L = self.L(phi_l)
        ftil = L.apply(from_harm(proj, fh))
        z = self.M(self.B(to_harm(proj, ftil))) - d
        dtil = -self.B.T_()(self.Mt(self.Cn.pinv()(z)))
        _, _, dphi = L.grad_apply(ftil, harm_to_qu(proj, dtil), alias_quirk)
        with np.errstate(divide="ignore", invalid="ignore"):
            return dphi - nan2zero(phi_l / self.Cphi)

    def simulate_data(self, phi_l, white_f, white_n):
        """`simulate(rng, ds; ϕ).d`: d = M B Lϕ f + n with f = sqrt(Cf)·rfft(white), n = sqrt(Cn)·rfft(white) (specialops.jl:6,93)"""
        f = self.Cf.sqrt()(rfft2(white_f).astype(ctype(self.proj.T)))
        n = self.Cn.sqrt()(rfft2(white_n).astype(ctype(self.proj.T)))
        return self.mean(self.L(phi_l), f) + n

    # ---- mixing (dataset.jl:96-117) -----------------------------------------------------
    def mix(self, fh, phi_l):
        """f° = Lϕ·D·f (QU map), ϕ° = G·ϕ (Fourier)."""
        L = self.L(phi_l)
        fo = L.apply(from_harm(self.proj, self.D(fh)))
        return fo, self.G * phi_l

    def unmix(self, fo_map, phio_l):
        with np.errstate(divide="ignore", invalid="ignore"):
            phi_l = nan2zero(phio_l / self.G)
        L = self.L(phi_l)
        fh = self.D.solve(to_harm(self.proj, L.inv(fo_map)))
        return fh, phi_l

    def logpdf_mixed(self, fo_map, phio_l, d=None):
        """`logpdf(Mixed(ds); f°, ϕ°)` (dataset.jl:84-87); logdet(D,θ)=logdet(G,θ)=0 at fiducial θ
        (src/generic.jl:269)."""
        fh, phi_l = self.unmix(fo_map, phio_l)
        return self.logpdf(fh, phi_l, d)

    def grad_logpdf_mixed(self, fo_map, phio_l, d=None, alias_quirk=False):
        """∇_(f°,ϕ°) logpdf(Mixed(ds)) by the chain rule Zygote walks in the reference
        (maximization.jl:178): pullbacks of `\\`/`*` of the flow are the δ-flows
        (flowops.jl:40-68), of `D \\ v` is `D' \\ Δ` (autodiff.jl:127-134), of a Gaussian term
        −½ z†Σ⁻¹z is −Σ⁻¹z (autodiff.jl:107).  Returns (logpdf, ∇f° [map], ∇ϕ° [Fourier])."""
        d = self.d if d is None else d
        proj = self.proj
        with np.errstate(divide="ignore", invalid="ignore"):
            phi_l = nan2zero(phio_l / self.G)
        L = self.L(phi_l)
        fhat = L.inv(fo_map)                                  # f̂ = Lϕ \ f°      (map)
        fh = self.D.solve(to_harm(proj, fhat))                # f = D \ f̂        (harm)
        fmap = from_harm(proj, fh)
        ftil = L.apply(fmap)                                  # f̃ = Lϕ f
        z = self.M(self.B(to_harm(proj, ftil))) - d
        Cninv_z = self.Cn.pinv()(z)
        Cfinv_f = self.Cf.solve(fh)
        with np.errstate(divide="ignore", invalid="ignore"):
            Cpinv_p = nan2zero(phi_l / self.Cphi)
        lp = -(self.dot(fh, Cfinv_f) + self.Cf.logdet(proj)) / 2 \
             - (dot_fourier(proj, phi_l, Cpinv_p) + logdet_fourier(proj, self.Cphi[None, None])) / 2 \
             - (self.dot(z, Cninv_z) + self.Cn.logdet(proj)) / 2
        # ∂/∂f̃ = −B'M'Cn⁻¹ z
        dtil = -self.B.T_()(self.Mt(Cninv_z))
        _, df1, dphi1 = L.grad_apply(ftil, harm_to_qu(proj, dtil), alias_quirk)
        gf = qu_to_harm(proj, df1) - self.Cf.pinv()(fh)       # ∂/∂f
        dhat = self.D.T_().solve(gf)                          # ∂/∂f̂ = D' \ g
        _, df2, dphi2 = L.grad_inv(fhat, harm_to_qu(proj, dhat), alias_quirk)
        gphi = dphi1 + dphi2 - Cpinv_p
        with np.errstate(divide="ignore", invalid="ignore"):
            gphio = nan2zero(gphi / self.G)                   # ∂/∂ϕ° = G' \ gϕ
        gfo = irfft2(df2, proj.Ny).astype(proj.T)             # ∂/∂f° in f°'s basis (QU map)
        return lp, gfo, gphio

    # ---- Wiener filter (maximization.jl:17-42, dataset.jl:129-132) ----------------------
    def precond_f(self):
        """Hessian_logpdf_preconditioner(:f) = Cf⁻¹ + B̂'M̂'Cn̂⁻¹M̂B̂"""
        return self.Cf.pinv() + (self.Bhat.T_() @ self.Mfhat.T_() @ self.Cnhat.pinv() @ self.Mfhat @ self.Bhat)

    def argmaxf_logpdf(self, phi_l, d=None, fstart=None, tol=1e-1, nsteps=500, offset=False):
        d = self.d if d is None else d
        L = self.L(phi_l)
        zero_f = np.zeros_like(d)
        Pc = self.precond_f()
        b = -self.gradientf_logpdf(zero_f, L, d)
        a0 = self.gradientf_logpdf(zero_f, L, np.zeros_like(d))
        if offset:
            b = b + a0
        A = lambda f: self.gradientf_logpdf(f, L, np.zeros_like(d)) - a0
        x0 = zero_f if fstart is None else fstart
        return conjugate_gradient(Pc.solve, A, b, x0, self.dot, nsteps, tol)


# ---------------------------------------------------------------------------------------

def load_sim(theta_pix, Nside, pol, T=np.float64, muK_arcmin_T=3.0, lknee=100.0, alphaknee=3.0,
             beam_fwhm=0.0, pixel_mask=None, bandpass_lmax=3000, nsteps=7, Nbatch=1,
             seeds=(1, 2, 3), Nphi="white", Nphi_fac=2, cls=None):
    """`load_sim` (src/dataset.jl:186-338) with the deterministic inputs of SURVEY.md §8(d).

    pixel_mask : None | dict(pad_deg=, apod_deg=)  -> border_mask stand-in for make_mask.
    Nphi       : 'white' (cheap stand-in: flat N⁰ = mean Cϕ level), an [x,ky] array, or a
                 callable ds -> array (tests pass oracle.quadratic_estimate for the faithful value).
    Returns dict(f, phi, ftilde, d, n, ds, proj, cls) with f,d,n harmonic and phi Fourier.
    """
    Ny, Nx = (Nside, Nside) if np.isscalar(Nside) else Nside
    proj = Proj(Ny, Nx, theta_pix, T)
    T = proj.T
    P = {"I": 1, "P": 2, "IP": 3}[pol]
    lmax = proj.lmax
    cls = load_camb() if cls is None else cls
    ncl = noise_cls(muK_arcmin_T, lknee, alphaknee, lmax)
    mk = lambda c, **kw: HarmOp.from_cls(pol, proj, c, **kw)

    Cphi = cl_to_2d(cls["total"]["pp"], proj)                              # :267
    Cfs, Cten = mk(cls["unlensed_scalar"]), mk(cls["tensor"])             # :268-269
    Cf = Cfs + Cten                                                        # :273 at r = r₀
    Cft = mk(cls["total"])                                                 # :270
    Cn = mk(ncl)                                                           # :271-272
    Mf = mk(lowpass(bandpass_lmax), units=1, te_zero=True)                 # :279
    Mpix = border_mask(proj, **pixel_mask) if pixel_mask is not None else None
    bcl = beam_cls(beam_fwhm, lmax)
    Bop = mk(Cls(bcl.ell, np.sqrt(bcl.cl)), units=1, te_zero=True)         # :300

    ds = DataSet(proj, P, Cf, Cn, Cphi, Mf, Bop, Mpix=Mpix, nsteps=nsteps, Cftilde=Cft)

    # simulate (simpleppl; specialops.jl:6): x = sqrt(C)·rfft(white)
    shp = (Nbatch, P, Nx, Ny)
    f = Cf.sqrt()(rfft2(white_noise(seeds[0], shp, T)))
    phi = (np.sqrt(Cphi) * rfft2(white_noise(seeds[1], (Nbatch, 1, Nx, Ny), T))).astype(ctype(T))
    n = Cn.sqrt()(rfft2(white_noise(seeds[2], shp, T)))
    L = ds.L(phi)
    ftilde = ds.lense(L, f)
    d = ds.M(ds.B(ftilde)) + n
    ds.d = d

    # mixing matrices (dataset.jl:316-329)
    if callable(Nphi):
        Nphi = Nphi(ds)
    elif isinstance(Nphi, str):
        lm = proj.lmag
        sel = (lm > 100) & (lm < 2000)
        Nphi = np.where(Cphi > 0, np.float64(np.exp(np.mean(np.log(Cphi[sel])))), 0).astype(T)
    ds.Nphi = (np.asarray(Nphi) / Nphi_fac).astype(T)
    ds.G = np.sqrt(1 + 2 * ds.Nphi * pinv(Cphi)).astype(T)                 # G₀⁻¹·G at fiducial Aϕ == I?  see note
    # note: G(θ) = pinv(G₀)·sqrt(I + 2Nϕ·pinv(Cϕ(Aϕ))) equals the identity at fiducial Aϕ
    # (dataset.jl:317-320); MAP_joint overrides G = I as well (maximization.jl:146).
    ds.G = np.ones_like(ds.G)
    s2len = T(np.deg2rad(5 / 60) ** 2)
    ds.D = ((Cf + (Cn.scale(2) + s2len)) @ Cf.pinv()).sqrt()              # :322-328
    return dict(f=f, phi=phi, ftilde=ftilde, d=d, n=n, ds=ds, proj=proj, cls=cls, Cfs=Cfs, Cten=Cten)
