"""LenseFlow: cache, velocities, RK4, the four flow operators and the δ-flow gradient.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
    src/lenseflow.jl:19-214     (LenseFlow, precompute!, velocity, velocityᴴ, negδvelocityᴴ)
    src/flowops.jl:11-14,40-68  (L*f, L'*f, L\\f, L'\\f and their pullbacks)
    src/numerical_algorithms.jl:11-24  (RK4Solver)
    src/field_vectors.jl:42-55,86-94   (fused mul!, 2x2 pinv!)
of /root/reference.
"""
import numpy as np
from .flatsky import rfft2, irfft2, grad_mults, gradhess, pinv

__all__ = ["LenseFlow", "rk4", "get_max_lensing_step"]


def rk4(F, y0, t0, t1, nsteps):
    """`RK4Solver` (src/numerical_algorithms.jl:11-24). `y0` is an array or a tuple of arrays;
    F(t, y) returns the same structure."""
    tup = isinstance(y0, tuple)

    def axpy(y, c, k):
        return tuple(a + c * b for a, b in zip(y, k)) if tup else y + c * k

    h = (t1 - t0) / nsteps
    y = tuple(a.copy() for a in y0) if tup else y0.copy()
    ts = np.linspace(t0, t1, nsteps + 1)[:-1]
    for t in ts:
        k1 = F(t, y)
        k2 = F(t + h / 2, axpy(y, h / 2, k1))
        k3 = F(t + h / 2, axpy(y, h / 2, k2))
        k4 = F(t + h, axpy(y, h, k3))
        if tup:
            y = tuple(a + h * (b1 + 2 * (b2 + b3) + b4) / 6 for a, b1, b2, b3, b4 in zip(y, k1, k2, k3, k4))
        else:
            y = y + h * (k1 + 2 * (k2 + k3) + k4) / 6
    return y


class LenseFlow:
    """`CachedLenseFlow` (src/lenseflow.jl:33-60,131-142).

    phi : real map, shape (Bphi, 1, Nx, Ny), or `phi_l`: its half-plane Fourier coefficients (Bphi, 1, Nx, Nyh).
    `gradhess(ϕ)` acts on ϕ in whatever basis it arrives (`∇ⁱ*f` only transforms a Map, src/specialops.jl:184-188,
    src/lenseflow.jl:135): a Fourier ϕ whose ky = 0 / Nyquist rows are not Hermitian-consistent -- every ϕ produced by a gradient
    step is -- is NOT projected through a map first, so ∂ϕ = irfft(iℓ·ϕ_l) sees the unprojected coefficients.
    Caches p(t), M⁻¹(t) at the 2n+1 times k/(2n).
    """

    def __init__(self, proj, phi, nsteps=7, phi_l=None):
        self.proj, self.n = proj, int(nsteps)
        self.T = proj.T
        if phi_l is None:
            phi = np.asarray(phi, dtype=proj.T)
            assert phi.ndim == 4 and phi.shape[1] == 1
            phi_l = rfft2(phi)
        self.phi_l = np.asarray(phi_l)
        assert self.phi_l.ndim == 4 and self.phi_l.shape[1] == 1
        self.phi = phi
        self._precompute()

    # src/lenseflow.jl:131-142
    def _precompute(self):
        T, n = self.T, self.n
        (gx, gy), ((Hxx, Hxy), (Hyx, Hyy)) = gradhess(self.proj, self.phi_l)
        gx, gy, Hxx, Hxy, Hyx, Hyy = (a.astype(T) for a in (gx, gy, Hxx, Hxy, Hyx, Hyy))
        self.p, self.Minv = {}, {}
        for k in range(2 * n + 1):
            t = T(k) / T(2 * n)
            # M = I + t∇∇ϕ ; pinv! reads b = A[2,1] for both off-diagonals (field_vectors.jl:86-94, quirk Q2)
            a, c, d = 1 + t * Hxx, t * Hyx, 1 + t * Hyy
            b = c
            idet = pinv(a * d - b * c)
            M11, M12, M21, M22 = idet * d, -idet * b, -idet * c, idet * a
            # p = M⁻¹' * ∇ϕ   (field_vectors.jl:46-47 with the adjoint of :40)
            px = M11 * gx + M21 * gy
            py = M12 * gx + M22 * gy
            self.p[k] = (px.astype(T), py.astype(T))
            self.Minv[k] = (M11.astype(T), M12.astype(T), M21.astype(T), M22.astype(T))

    def _k(self, t):
        k = int(round(float(t) * 2 * self.n))
        assert abs(float(t) * 2 * self.n - k) < 1e-6, "RK stage time off the 2n+1 grid (quirk Q3)"
        return k

    # ---- velocities -------------------------------------------------------------------
    def _vel(self, t, f):
        """src/lenseflow.jl:150-161 : v = p₁·∂x f + p₂·∂y f  (Map state)."""
        ilx, ily = grad_mults(self.proj)
        px, py = self.p[self._k(t)]
        fl = rfft2(f)
        Ny = self.proj.Ny
        return (px * irfft2(ilx * fl, Ny) + py * irfft2(ily * fl, Ny)).astype(self.T)

    def _velH(self, t, yl):
        """src/lenseflow.jl:163-174 : v = iℓx·rfft(p₁·irfft y) + iℓy·rfft(p₂·irfft y)  (Fourier state).
        (`-∇ᵢ'` = +iℓ: conj(∇diag) flips the prefactor, src/specialops.jl:161-165.)"""
        ilx, ily = grad_mults(self.proj)
        px, py = self.p[self._k(t)]
        y = irfft2(yl, self.proj.Ny)
        return ilx * rfft2((px * y).astype(self.T)) + ily * rfft2((py * y).astype(self.T))

    def _veldelta(self, t, state, alias_quirk):
        """src/lenseflow.jl:176-214, state = (f [Map], δf [Fourier], δϕ [Fourier])."""
        f, dfl, dpl = state
        proj, T, Ny = self.proj, self.T, self.proj.Ny
        ilx, ily = grad_mults(proj)
        k = self._k(t)
        px, py = self.p[k]
        M11, M12, M21, M22 = self.Minv[k]
        tt = T(k) / T(2 * self.n)
        # dδf/dt (:184-188)
        Ldf = irfft2(dfl, Ny).astype(T)
        ddf = ilx * rfft2(px * Ldf) + ily * rfft2(py * Ldf)
        # df/dt (:191-195)
        fl = rfft2(f)
        gfx, gfy = irfft2(ilx * fl, Ny).astype(T), irfft2(ily * fl, Ny).astype(T)
        df = px * gfx + py * gfy
        # dδϕ/dt (:198-206):  w_k = Σ_pol Łδf·∇_k f  (spin-adjoint product, proj_lambert.jl:423-430)
        w1 = np.sum(Ldf * gfx, axis=1, keepdims=True)
        w2 = np.sum(Ldf * gfy, axis=1, keepdims=True)
        u1 = M11 * w1 + M12 * w2
        if alias_quirk:
            # Q1: input and output of mul! share memory (lenseflow.jl:198-200 with
            # field_vectors.jl:48-49): v[2] is computed from the *already overwritten* w[1]
            u2 = M21 * u1 + M22 * w2
        else:
            u2 = M21 * w1 + M22 * w2
        u = (u1.astype(T), u2.astype(T))
        ddp = ilx * rfft2(u[0]) + ily * rfft2(u[1])          # -∇ⁱ' * Ð(u)  (:201-202)
        pj = (px, py)
        il = (ilx, ily)
        for i in range(2):                                   # :204-206
            for j in range(2):
                # ∇ⁱ[i]' * ∇ᵢ[j]' * Ð(t·p_j·u_i) = (-iℓ_i)(-iℓ_j)(…) = -ℓ_iℓ_j (…)
                ddp = ddp + (-il[i]) * ((-il[j]) * rfft2((tt * pj[j] * u[i]).astype(T)))
        return (df.astype(T), ddf, ddp)

    # ---- operators (src/flowops.jl:11-14) -------------------------------------------------
    def apply(self, f):
        """`Lϕ * f` : forward velocity, t: 0 -> 1.  f Map (B,P,Nx,Ny)."""
        return rk4(self._vel, np.asarray(f, self.T), 0.0, 1.0, self.n)

    def inv(self, f):
        """`Lϕ \\ f` : forward velocity, t: 1 -> 0."""
        return rk4(self._vel, np.asarray(f, self.T), 1.0, 0.0, self.n)

    def adj(self, gl):
        """`Lϕ' * g` : adjoint velocity, t: 1 -> 0, Fourier in/out."""
        return rk4(self._velH, gl, 1.0, 0.0, self.n)

    def invadj(self, gl):
        """`Lϕ' \\ g` : adjoint velocity, t: 0 -> 1, Fourier in/out."""
        return rk4(self._velH, gl, 0.0, 1.0, self.n)

    # ---- pullbacks (src/flowops.jl:40-68) -------------------------------------------------
    def grad_apply(self, ftilde, delta_l, alias_quirk=False):
        """Pullback of f̃ = Lϕ*f: δ-flow t: 1 -> 0 from (f̃, Δ, 0).
        Returns (f, δf [Fourier], δϕ [Fourier (Bϕ,1,Nx,Nyh)])."""
        return self._delta(ftilde, delta_l, 1.0, 0.0, alias_quirk)

    def grad_inv(self, f, delta_l, alias_quirk=False):
        """Pullback of f = Lϕ\\f̃: δ-flow t: 0 -> 1 from (f, Δ, 0)."""
        return self._delta(f, delta_l, 0.0, 1.0, alias_quirk)

    def _delta(self, f0, delta_l, t0, t1, alias_quirk):
        B = max(f0.shape[0], self.phi_l.shape[0])
        C = delta_l.dtype
        dp0 = np.zeros((B, 1, self.proj.Nx, self.proj.Nyh), dtype=C)
        F = lambda t, s: self._veldelta(t, s, alias_quirk)
        return rk4(F, (np.asarray(f0, self.T), delta_l, dp0), t0, t1, self.n)


def get_max_lensing_step(proj, phi, eta):
    """src/lenseflow.jl:242-256 (phi, eta real maps)."""
    _, ((p11, p12), (p21, p22)) = gradhess(proj, rfft2(phi))
    _, ((e11, e12), (e21, e22)) = gradhess(proj, rfft2(eta))
    a = e11 * e22 - e12 ** 2
    b = e11 * (1 + p22) + e22 * (1 + p11) - 2 * e12 * p12
    c = (1 + p11) * (1 + p22) - p12 ** 2
    with np.errstate(invalid="ignore", divide="ignore"):
        sq = np.sqrt(b * b - 4 * a * c)
        a1 = (-b + sq) / (2 * a)
        a2 = (-b - sq) / (2 * a)
    return min(a1[a1 > 0].min(), a2[a2 > 0].min())
