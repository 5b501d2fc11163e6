"""Independent known answers for the flat-sky lensing path (used by the oracle tests AND the -m gpu tests).

Nothing here imports `oracle/` or the product: the answers come from first principles, so they pin conventions that the
reference's own self-consistency properties (adjoint identity, finite differences, round trips) cannot see -- e.g. the sign
of the deflection.

  * `remap_exact`  : f̃(x) = f(x + ∇ϕ(x)) -- what `LenseFlow(ϕ)*f` integrates (src/lenseflow.jl:1-17 docstring) -- evaluated
                     by direct summation of the band-limited Fourier series of f at the displaced positions, the
                     deflection itself by central differences of the Fourier interpolant of ϕ (no ℓ-multiplier convention used).
  * `bandlimited`  : smooth periodic test maps with no power at or near Nyquist, so that the interpolant is unambiguous.
"""
import numpy as np


def bandlimited(seed, Nx, Ny, kmax_frac=0.35, slope=2.0, shape=(1, 1)):
    """real periodic maps [*shape, Nx, Ny] with power only at |k| <= kmax_frac * N/2 (index units), red spectrum"""
    rng = np.random.default_rng(seed)
    kx = np.fft.fftfreq(Nx) * Nx
    ky = np.fft.fftfreq(Ny) * Ny
    kk = np.sqrt((kx[:, None] / (Nx / 2)) ** 2 + (ky[None, :] / (Ny / 2)) ** 2)
    amp = np.where((kk > 0) & (kk <= kmax_frac), (kk + 0.05) ** (-slope), 0.0)
    w = rng.standard_normal(shape + (Nx, Ny))
    m = np.fft.ifft2(np.fft.fft2(w) * amp).real
    return m / m.std()


def interp_fourier(m, xs, ys):
    """value of the band-limited periodic interpolant of m[x, y] (pixel units) at the points (xs, ys), by direct summation"""
    Nx, Ny = m.shape
    F = np.fft.fft2(m) / (Nx * Ny)
    kx = np.fft.fftfreq(Nx) * Nx
    ky = np.fft.fftfreq(Ny) * Ny
    assert np.abs(F[Nx // 2]).max() < 1e-12 * np.abs(F).max() and np.abs(F[:, Ny // 2]).max() < 1e-12 * np.abs(F).max(), "Nyquist power"
    ex = np.exp(2j * np.pi * np.outer(xs.ravel(), kx) / Nx)            # (npts, Nx)
    ey = np.exp(2j * np.pi * np.outer(ys.ravel(), ky) / Ny)            # (npts, Ny)
    return np.einsum("pa,ab,pb->p", ex, F, ey).real.reshape(xs.shape)


def deflection(phi, dx_rad, eps=1e-3):
    """(∂ϕ/∂x, ∂ϕ/∂y) in PIXEL units per radian^0: ∇ϕ / Δx, by central differences of the Fourier interpolant of ϕ[x, y]"""
    Nx, Ny = phi.shape
    X, Y = np.meshgrid(np.arange(Nx, dtype=float), np.arange(Ny, dtype=float), indexing="ij")
    gx = (interp_fourier(phi, X + eps, Y) - interp_fourier(phi, X - eps, Y)) / (2 * eps * dx_rad)
    gy = (interp_fourier(phi, X, Y + eps) - interp_fourier(phi, X, Y - eps)) / (2 * eps * dx_rad)
    return gx / dx_rad, gy / dx_rad


def remap_exact(f, phi, theta_pix_arcmin, sign=+1.0):
    """f̃[x,y] = f((x,y) + sign·∇ϕ) for maps f[..., Nx, Ny], ϕ[Nx, Ny]; θpix in arcmin.  Returns (f̃, rms deflection in pixels)."""
    dx = np.deg2rad(theta_pix_arcmin / 60.0)
    Nx, Ny = phi.shape
    ax, ay = deflection(phi, dx)
    X, Y = np.meshgrid(np.arange(Nx, dtype=float), np.arange(Ny, dtype=float), indexing="ij")
    out = np.empty_like(f, dtype=float)
    for idx in np.ndindex(*f.shape[:-2]):
        out[idx] = interp_fourier(f[idx], X + sign * ax, Y + sign * ay)
    return out, float(np.sqrt(np.mean(ax ** 2 + ay ** 2)))
