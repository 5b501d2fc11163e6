"""Oracle pinning, part 1: geometry, basis transforms, reductions.

Re-runs the reference's own known-answer / property tests on the NumPy restatement:
    test/runtests.jl:116-131  basis round trips
    test/runtests.jl:249-285  logdet / tr vs dense fft
    test/runtests.jl:289-295  EB-diagonal operator as QU blocks
"""
import numpy as np
import pytest

import oracle as O

NSIDES = [(8, 8), (4, 8), (8, 4)]
NSIDES_BIG = [(128, 128), (64, 128), (128, 64)]


@pytest.mark.parametrize("Ny,Nx", NSIDES + NSIDES_BIG)
def test_proj_geometry(Ny, Nx):
    p = O.Proj(Ny, Nx, 3.0, np.float64)
    # src/proj_lambert.jl:63-64: Nyquist entries negative for even N
    assert p.ly.shape == (Ny // 2 + 1,) and p.lx.shape == (Nx,)
    assert p.ly[-1] < 0 and p.lx[Nx // 2] < 0
    np.testing.assert_allclose(p.ly[1], 2 * np.pi / (Ny * np.deg2rad(3 / 60)))
    np.testing.assert_allclose(-p.ly[-1], p.nyquist)
    assert p.lam[0] == 1 and p.lam[-1] == 1 and np.all(p.lam[1:-1] == 2)
    # rotation stays orthogonal entry by entry, also on the patched Nyquist row (:69-71)
    np.testing.assert_allclose(p.sin2phi ** 2 + p.cos2phi ** 2, 1, atol=1e-12)
    # the patch makes sin2ϕ on the ky=Nyquist row even in kx
    for x in range(1, Nx // 2):
        assert p.sin2phi[Nx - x, -1] == p.sin2phi[x, -1]


@pytest.mark.parametrize("Ny,Nx", NSIDES)
@pytest.mark.parametrize("P", [1, 2])
def test_basis_round_trips(Ny, Nx, P):
    # test/runtests.jl:116-131 : Bin(Bout(Bin(f))) ≈ f over Map/Fourier and QUMap/QUFourier/EBMap/EBFourier
    p = O.Proj(Ny, Nx, 1.0, np.float64)
    f = np.random.default_rng(4).random((1, P, Nx, Ny))
    fl = O.rfft2(f)
    np.testing.assert_allclose(O.irfft2(fl, Ny), f, atol=1e-13)
    if P == 2:
        eb = O.qu2eb(p, fl)
        np.testing.assert_allclose(O.eb2qu(p, eb), fl, atol=1e-12)
        ebmap = O.irfft2(eb, Ny)
        np.testing.assert_allclose(O.irfft2(O.eb2qu(p, O.rfft2(ebmap)), Ny), f, atol=1e-12)
        np.testing.assert_allclose(O.from_harm(p, O.to_harm(p, f)), f, atol=1e-12)


def test_logdet_tr_map_known_answers():
    # test/runtests.jl:251-256,268-273 (Map-basis known answers; here just the arithmetic)
    x = np.array([[1, -2], [3, -4]], dtype=float)
    assert np.isclose(np.sum(np.log(np.abs(x))) + np.log(np.prod(np.sign(x))), np.log(24))
    assert np.isclose(np.sum(x), -2)


@pytest.mark.parametrize("Ny,Nx", NSIDES_BIG)
def test_logdet_tr_fourier_vs_dense_fft(Ny, Nx):
    # test/runtests.jl:258-265, 275-282
    p = O.Proj(Ny, Nx, 1.0, np.float64)
    x = np.random.default_rng(4).random((Nx, Ny))
    full = np.fft.fft2(x)
    xl = O.rfft2(x[None, None])
    ld = np.sum(np.log(np.abs(full)))
    np.testing.assert_allclose(O.logdet_fourier(p, xl)[0], ld, rtol=1e-10)
    for P in (2, 3):
        xs = np.repeat(xl, P, axis=1)
        np.testing.assert_allclose(O.logdet_fourier(p, xs)[0], P * ld, rtol=1e-10)
    # batched (runtests.jl:264)
    xb = np.concatenate([xl, xl], axis=0)
    np.testing.assert_allclose(O.logdet_fourier(p, xb), [ld, ld], rtol=1e-10)
    np.testing.assert_allclose(O.tr_fourier(p, xl)[0], np.real(np.sum(full)), rtol=1e-10)
    # Parseval: util_fft.jl:137-143 docstring identity
    np.testing.assert_allclose(np.sum(np.abs(full) ** 2), np.sum(p.lam * np.abs(xl) ** 2), rtol=1e-12)
    np.testing.assert_allclose(O.dot_fourier(p, xl, xl)[0], np.sum(x * x), rtol=1e-12)


@pytest.mark.parametrize("Ny,Nx", NSIDES)
def test_eb_diag_as_qu_blocks(Ny, Nx):
    # test/runtests.jl:289-295 with src/proj_lambert.jl:304-314
    p = O.Proj(Ny, Nx, 1.0, np.float64)
    rng = np.random.default_rng(4)
    C = O.rfft2(rng.random((1, 2, Nx, Ny)))
    El, Bl = C[0, 0], C[0, 1]
    f = O.rfft2(rng.random((1, 2, Nx, Ny)))
    Q, U = f[0, 0], f[0, 1]
    s, c = p.sin2phi, p.cos2phi
    QQ, QU, UU = Bl * s ** 2 + El * c ** 2, (El - Bl) * s * c, Bl * c ** 2 + El * s ** 2
    lhs = O.eb2qu(p, C * O.qu2eb(p, f))
    rhs = np.stack([QQ * Q + QU * U, UU * U + QU * Q])[None]
    np.testing.assert_allclose(lhs, rhs, atol=1e-10)


def test_cls_interpolation_and_cov():
    # src/numerical_algorithms.jl:148-177, src/cls.jl:288-309, src/specialops.jl:236-240
    c = O.Cls([2, 3, 4], [1.0, 3.0, 2.0])
    np.testing.assert_allclose(c([2, 2.5, 4]), [1, 2, 2])
    assert np.all(np.isnan(c([1.9, 4.1])))
    lp = O.lowpass(3000)
    assert lp.ell[0] == 0 and lp.ell[-1] == 3000 and lp.cl[0] == 1 and abs(lp.cl[-1]) < 1e-15
    assert np.all(lp.cl[:2951] == 1) and lp.cl[2951] == 1.0 and lp.cl[2952] < 1
    n = O.noise_cls(3.0, 100, 3, 5000)
    np.testing.assert_allclose(n["EE"].cl, 2 * n["TT"].cl)
    np.testing.assert_allclose(n["TT"](100.0), 2 * np.deg2rad(3 / 60) ** 2)
    p = O.Proj(64, 64, 2.0, np.float32)
    C = O.cl_to_2d(O.load_camb()["total"]["TT"], p)
    assert C.dtype == np.float32 and C[0, 0] == 0 and np.all(np.isfinite(C)) and C[1, 1] > 0


def test_camb_fixture(camb):
    # dat/default_camb_Cls.jld2 via tools/extract_cls.py: Dℓ^TT first peak near ℓ≈220, ~5700 μK²
    tt = camb["total"]["TT"]
    ell = np.arange(100, 400)
    D = tt(ell) * ell * (ell + 1) / 2 / np.pi
    assert 200 < ell[np.argmax(D)] < 240 and 5000 < D.max() < 6500
    assert np.all(camb["unlensed_scalar"]["BB"].cl == 0)
