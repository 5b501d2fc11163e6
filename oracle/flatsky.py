"""Flat-sky geometry, FFT basis transforms, diagonal operators, reductions.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
    src/proj_lambert.jl:48-75,146-175,245-371,423-430
    src/util_fft.jl:20-44,137-143
    src/specialops.jl:6-23,146-188,232-241
    src/numerical_algorithms.jl:148-177   (LinearInterpolation)
    src/cls.jl:11-35,288-309
of /root/reference.
"""
import os
import numpy as np
import scipy.fft as sfft

__all__ = [
    "Proj", "rfft2", "irfft2", "Cls", "cl_to_2d", "nan2zero", "pinv", "qu2eb", "eb2qu",
    "dot_map", "dot_fourier", "logdet_fourier", "tr_fourier", "noise_cls", "beam_cls",
    "lowpass", "load_camb", "grad_mults", "gradhess", "white_noise", "border_mask",
    "diag_mul", "diag_div", "ctype",
]

_WORKERS = int(os.environ.get("CMBL_ORACLE_FFT_WORKERS", os.cpu_count() or 1))


def ctype(T):
    return np.complex64 if np.dtype(T) == np.float32 else np.complex128


class Proj:
    """`ProjLambert` metadata (src/proj_lambert.jl:48-75)."""

    def __init__(self, Ny, Nx, theta_pix=1.0, T=np.float64):
        T = np.dtype(T).type
        self.Ny, self.Nx, self.theta_pix, self.T = int(Ny), int(Nx), float(theta_pix), T
        self.dx = T(np.deg2rad(theta_pix / 60))                       # :58
        self.dlx = T(2 * np.pi / float(T(Nx) * self.dx))              # :59
        self.dly = T(2 * np.pi / float(T(Ny) * self.dx))              # :60
        self.nyquist = T(2 * np.pi / float(T(2) * self.dx))           # :61
        self.Opix = T(self.dx * self.dx)                              # :62
        ky = np.fft.ifftshift(np.arange(-(Ny // 2), (Ny - 1) // 2 + 1))
        kx = np.fft.ifftshift(np.arange(-(Nx // 2), (Nx - 1) // 2 + 1))
        self.ly = (ky.astype(T) * self.dly)[: Ny // 2 + 1]            # :63  (last entry NEGATIVE for even Ny)
        self.lx = kx.astype(T) * self.dlx                             # :64
        # arrays indexed [x, ky] (Julia [ky, x] column-major)
        LX, LY = self.lx[:, None], self.ly[None, :]
        self.lmag = np.sqrt(LX * LX + LY * LY).astype(T)              # :65
        phi = np.arctan2(LY + 0 * LX, LX + 0 * LY).astype(T)          # :66 angle(ℓx + iℓy)
        self.sin2phi = np.sin(2 * phi).astype(T)                      # :67
        self.cos2phi = np.cos(2 * phi).astype(T)
        lam = np.full(Ny // 2 + 1, 2, dtype=T)                        # util_fft.jl:137-143
        lam[0] = 1
        if Ny % 2 == 0:
            lam[-1] = 1
            # :69-71  sin2ϕ[end, end:-1:(Nx÷2+2)] .= sin2ϕ[end, 2:Nx÷2]   (1-based)
            src = self.sin2phi[1: Nx // 2, -1].copy()                 # x = 2..Nx÷2 (1-based)
            dst_idx = np.arange(Nx - 1, Nx // 2, -1)                  # x = Nx, Nx-1, ..., Nx÷2+2 (1-based) -> 0-based
            self.sin2phi[dst_idx, -1] = src
        self.lam = lam
        self.Nyh = Ny // 2 + 1

    @property
    def lmax(self):
        """src/dataset.jl:232"""
        return int(round(np.ceil(np.sqrt(2) * float(self.nyquist)) + 1))


def rfft2(a):
    """`m_rfft(arr,(1,2))` (src/util_fft.jl:20): unnormalised R2C over (y, x); y halved."""
    return sfft.rfft2(a, axes=(-2, -1), workers=_WORKERS)


def irfft2(A, Ny):
    """`m_irfft(arr, Ny, (1,2))` (src/util_fft.jl:21-25): inverse, normalised by 1/(Ny·Nx).

    Like FFTW's multi-dim c2r (and cuFFT's) the complex inverse runs over x first and the
    c2r over y last, so imaginary parts of the ky=0 / ky=Nyquist rows *after the x transform*
    are ignored.  pocketfft (scipy) has the same structure and the same behaviour.
    """
    Nx = A.shape[-2]
    return sfft.irfft2(A, s=(Nx, Ny), axes=(-2, -1), workers=_WORKERS)


def nan2zero(x):
    """src/util.jl:32"""
    return np.where(np.isfinite(x), x, 0).astype(x.dtype, copy=False)


def pinv(x):
    """scalar `pinv` broadcast (Base; GPU form ext/CMBLensingCUDAExt.jl:55)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        xi = 1 / x
    return np.where(np.isfinite(xi), xi, 0).astype(x.dtype, copy=False)


class Cls:
    """`Cℓs` with linear interpolation, NaN outside the tabulated range
    (src/cls.jl:11-29, src/numerical_algorithms.jl:148-177)."""

    def __init__(self, ell, cl):
        ell = np.asarray(ell, dtype=np.float64)
        cl = np.asarray(cl, dtype=np.float64)
        m = ~np.isnan(cl)
        self.ell, self.cl = ell[m], cl[m]

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        y = np.interp(x, self.ell, self.cl)
        return np.where((x < self.ell[0]) | (x > self.ell[-1]), np.nan, y)


def cl_to_2d(cls, proj, units=None):
    """`Cℓ_to_2D` / `Cℓ_to_Cov(:I,…)` (src/proj_lambert.jl:173-175,362-364):
    T.(nan2zero.(Cℓ(ℓmag))) / units, units = Ωpix by default.  Shape [x, ky]."""
    c = nan2zero(cls(proj.lmag)).astype(proj.T)
    u = proj.Opix if units is None else proj.T(units)
    return (c / u).astype(proj.T)


def noise_cls(muK_arcmin_T=3.0, lknee=100.0, alphaknee=3.0, lmax=8000):
    """`noiseCℓs` with beamFWHM=0 (src/cls.jl:288-299).  Returns dict TT,EE,BB,TE."""
    ell = np.arange(2, lmax + 1)
    n1f = 1 + (lknee / ell) ** alphaknee
    base = np.deg2rad(muK_arcmin_T / 60) ** 2
    out = {k: Cls(ell, (1 if k == "TT" else 2) * base * n1f) for k in ("TT", "EE", "BB")}
    out["TE"] = Cls(ell, np.zeros(ell.size))
    return out


def beam_cls(beam_fwhm=0.0, lmax=8000):
    """`beamCℓs` (src/cls.jl:307-309): Wℓ; a map is multiplied by sqrt of this."""
    ell = np.arange(2, lmax + 1)
    return Cls(ell, np.exp(-ell.astype(np.float64) ** 2 * np.deg2rad(beam_fwhm / 60) ** 2 / (8 * np.log(2))))


def lowpass(l, dl=50):
    """`LowPass(ℓ; Δℓ=50)` (src/specialops.jl:236-240)."""
    up = (np.cos(np.linspace(np.pi, 0, dl)) + 1) / 2
    return Cls(np.arange(0, l + 1), np.concatenate([np.ones(l - dl + 1), 1 - up]))


def load_camb(path=None):
    """The decoded dat/default_camb_Cls.jld2 (tools/extract_cls.py) as nested dicts of Cls."""
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "camb_cls.npz")
    z = np.load(path)
    ell = z["ell"]
    out = {}
    for g in ("unlensed_scalar", "lensed_scalar", "tensor", "unlensed_total", "total"):
        out[g] = {k: Cls(ell, z[f"{g}_{k}"]) for k in ("TT", "EE", "BB", "TE")}
        out[g]["pp"] = Cls(ell, z["phiphi"])
    return out


# ---------------------------------------------------------------------------------------
# polarisation rotation (src/proj_lambert.jl:253-258, 266-271)

def qu2eb(proj, QU):
    """QUFourier -> EBFourier: E = -Q c - U s ; B = Q s - U c.  QU[..., 2, x, ky]."""
    c, s = proj.cos2phi, proj.sin2phi
    Q, U = QU[..., 0, :, :], QU[..., 1, :, :]
    return np.stack([-Q * c - U * s, Q * s - U * c], axis=-3)


def eb2qu(proj, EB):
    """EBFourier -> QUFourier: Q = -E c + B s ; U = -E s - B c."""
    c, s = proj.cos2phi, proj.sin2phi
    E, B = EB[..., 0, :, :], EB[..., 1, :, :]
    return np.stack([-E * c + B * s, -E * s - B * c], axis=-3)


# ---------------------------------------------------------------------------------------
# reductions (src/proj_lambert.jl:318-353); one value per batch slot

def dot_map(a, b):
    """Σ a·b over (y,x,pol), per batch (src/proj_lambert.jl:318-321)."""
    return np.sum(a * b, axis=(-1, -2, -3))


def dot_fourier(proj, a, b):
    """Σ λ·Re(conj(a) b) / (Ny Nx) (src/proj_lambert.jl:322-325)."""
    z = np.real(np.conj(a) * b)
    return np.sum(z * proj.lam, axis=(-1, -2, -3)) / (proj.Ny * proj.Nx)


def logdet_fourier(proj, d):
    """Σ λ·log|d| with non-finite -> 0 (src/proj_lambert.jl:331-336). d[..., x, ky]."""
    with np.errstate(divide="ignore"):
        v = np.log(np.abs(d)) * proj.lam
    v = np.where(np.isfinite(v), v, 0)
    return np.real(np.sum(v, axis=(-1, -2, -3) if d.ndim >= 3 else None))


def tr_fourier(proj, d):
    """src/proj_lambert.jl:346-350"""
    return np.real(np.sum(d * proj.lam, axis=(-1, -2, -3) if d.ndim >= 3 else None))


# ---------------------------------------------------------------------------------------
# diagonal operators (src/specialops.jl:9-10)

def diag_mul(d, f):
    return d * f


def diag_div(d, f):
    """`D \\ f = nan2zero.(diag .\\ f)` (src/specialops.jl:10)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return nan2zero(f / d)


# ---------------------------------------------------------------------------------------
# gradients (src/specialops.jl:146-188, src/proj_lambert.jl:146-159)

def grad_mults(proj):
    """(i·ℓx, i·ℓy) broadcastable over [x, ky]: ∇[1] <-> x (Julia dim 2), ∇[2] <-> y (dim 1)."""
    C = ctype(proj.T)
    return (1j * proj.lx[:, None]).astype(C), (1j * proj.ly[None, :]).astype(C)


def gradhess(proj, phi_l):
    """`gradhess(ϕ)` (src/specialops.jl:184-188) in Map space.
    Returns g = (∂xϕ, ∂yϕ), H = ((Hxx, Hxy), (Hyx, Hyy)) with H[i][j] = ∇ᵢ[j]*g[i]."""
    ilx, ily = grad_mults(proj)
    Ny = proj.Ny
    gl = (ilx * phi_l, ily * phi_l)
    g = tuple(irfft2(x, Ny) for x in gl)
    H = tuple(tuple(irfft2(m * gl[i], Ny) for m in (ilx, ily)) for i in range(2))
    return g, H


# ---------------------------------------------------------------------------------------
# synthetic inputs

def white_noise(seed, shape, T):
    """𝒩(0,1) maps from NumPy PCG64(seed), drawn in float64 then cast (shape = (B,P,Nx,Ny))."""
    return np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(T)


def border_mask(proj, pad_deg=1.0, apod_deg=1.0):
    """Deterministic cosine-apodised border mask [x, y]; stand-in *input* for `make_mask`
    (src/masking.jl:1-25 needs ImageMorphology; the mask is data, not part of the path)."""
    def prof(n):
        pix_deg = proj.theta_pix / 60
        d = (np.minimum(np.arange(n), n - 1 - np.arange(n)) + 0.5) * pix_deg
        t = np.clip((d - pad_deg) / max(apod_deg, 1e-30), 0, 1)
        return (1 - np.cos(np.pi * t)) / 2
    return np.outer(prof(proj.Nx), prof(proj.Ny)).astype(proj.T)
