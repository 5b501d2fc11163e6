"""Small maps (32..128 per side): the one-launch flows of csrc/kernels_small.hpp -- a whole LenseFlow as ONE launch, one workgroup per
(pol, batch) slice with the half plane resident in LDS -- against the oracle, against the two-launches-per-stage path they replace
(option `small_flow`), and for independence of the batch composition.  Reference: src/lenseflow.jl:150-174 under
src/numerical_algorithms.jl:11-24; the sizes are the reference's own test sizes (test/runtests.jl:53)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as O
from oracle.lenseflow import LenseFlow as OLenseFlow
from _tol import close
from test_gpu_parity import DT, TOL, sims, camb      # noqa: F401  (camb: fixture)

SHAPES = [(128, 128, 2, 1, 1), (64, 128, 2, 2, 2), (128, 64, 3, 1, 1), (64, 64, 2, 3, 1), (32, 32, 1, 2, 2), (32, 128, 2, 1, 1), (128, 32, 1, 1, 1)]


def _flows(C, L, F, f, gl):
    return dict(Lf=(L * F(f, C.MAP)).arr.clone(), Linv=L.ldiv(F(f, C.MAP)).arr.clone(), adj=(L.adjoint * F(gl, C.FOURIER)).arr.clone(),
                invadj=L.adjoint.ldiv(F(gl, C.FOURIER)).arr.clone())


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("Ny,Nx,P,B,Bphi", SHAPES)
def test_small_flows_vs_oracle_and_vs_the_staged_path(camb, prec, Ny, Nx, P, B, Bphi):
    import cmblensing_jl_amd as C
    tT, nT = DT[prec]
    n = 7
    oproj, simf, simp = sims(camb, Ny, Nx, P, B)
    f = simf(1).astype(nT).astype(np.float64)
    g = simf(11).astype(nT).astype(np.float64)
    phi = simp(2, Bphi).astype(nT).astype(np.float64)
    gl = O.rfft2(g)
    OL = OLenseFlow(oproj, phi, n)
    want = dict(Lf=OL.apply(f), Linv=OL.inv(f), adj=OL.adj(gl), invadj=OL.invadj(gl))
    p = C.ProjLambert(Ny, Nx, 2.0, tT)
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    L = C.LenseFlow(p, n)(F(phi, C.MAP))
    p.set_option("small_flow", 0)
    staged = _flows(C, L, F, f, gl)
    p.set_option("small_flow", 2)                                            # wherever a kernel exists (1, the default: up to 64 x 64)
    small = _flows(C, L, F, f, gl)
    has_kernel = prec == "f32" or Ny * Nx <= 4096                            # SmallGeom::fits (double precision up to 64 x 64)
    for k in want:
        tol = TOL[prec]["flow" if k in ("Lf", "Linv") else "adj"]
        close(f"small {k} vs oracle", small[k].cpu().numpy(), want[k], tol)
        close(f"small {k} vs staged", small[k].cpu().numpy(), staged[k].cpu().numpy(), tol)
        same = bool(torch.equal(small[k], staged[k]))
        assert same == (not has_kernel), (k, "the option must select a different kernel exactly where one exists", same, has_kernel)
    # the default setting takes the one-launch path up to 64 x 64 pixels -- by shape alone, never by batch size
    p.set_option("small_flow", 1)
    dflt = (L * F(f, C.MAP)).arr
    assert torch.equal(dflt, small["Lf"] if Ny * Nx <= 4096 else staged["Lf"])


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_small_flow_result_does_not_depend_on_the_batch(camb, prec):
    """slot b of a batch of 5 (own phi each) == the same slot run alone: bit for bit (one workgroup per slice, no cross-slice arithmetic)"""
    import cmblensing_jl_amd as C
    tT, nT = DT[prec]
    Ny, Nx, P, B = 64, 64, 2, 5
    oproj, simf, simp = sims(camb, Ny, Nx, P, B)
    f = simf(1).astype(nT).astype(np.float64)
    gl = O.rfft2(simf(11).astype(nT).astype(np.float64))
    phi = simp(2, B).astype(nT).astype(np.float64)
    p = C.ProjLambert(Ny, Nx, 2.0, tT)
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    assert p.get_option("small_flow") == 1
    allb = _flows(C, C.LenseFlow(p, 7)(F(phi, C.MAP)), F, f, gl)
    for b in (0, 3):
        one = _flows(C, C.LenseFlow(p, 7)(F(phi[b:b + 1], C.MAP)), F, f[b:b + 1], gl[b:b + 1])
        for k in allb:
            assert torch.equal(allb[k][b:b + 1], one[k]), (k, b)


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("Ny,Nx,P,B,Bphi", [(64, 64, 2, 2, 2), (32, 64, 3, 1, 1), (64, 32, 1, 3, 1), (32, 32, 2, 1, 1)])
@pytest.mark.parametrize("mode", ["fwd", "inv"])
def test_small_delta_flow_vs_oracle_and_vs_the_staged_path(camb, prec, Ny, Nx, P, B, Bphi, mode):
    """the pullback of L*f / L\\f (the delta flow: f, delta f and the delta-phi quadrature) as ONE launch per flow (k_small_delta + the shared
    end-of-flow quadrature), both settings of the alias quirk"""
    import cmblensing_jl_amd as C
    tT, nT = DT[prec]
    n = 7
    oproj, simf, simp = sims(camb, Ny, Nx, P, B)
    f = simf(1).astype(nT).astype(np.float64)
    phi = simp(2, Bphi).astype(nT).astype(np.float64)
    OL = OLenseFlow(oproj, phi, n)
    fe = (OL.apply(f) if mode == "fwd" else OL.inv(f)).astype(nT).astype(np.float64)
    delta = O.rfft2(simf(7)).astype(np.complex64 if prec == "f32" else np.complex128).astype(np.complex128)
    p = C.ProjLambert(Ny, Nx, 2.0, tT)
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    L = C.LenseFlow(p, n)(F(phi, C.MAP))
    fm = C.FLOW_FWD if mode == "fwd" else C.FLOW_INV
    has_kernel = Ny * Nx * (4 if prec == "f32" else 8) <= 4096 * 4              # SmallGeom::delta_fits
    for quirk in (False, True):
        f0, df, dp = (OL.grad_apply if mode == "fwd" else OL.grad_inv)(fe, delta, alias_quirk=quirk)
        p.set_option("small_flow", 0)
        sdp, sdf, sf0 = L.gradient(fm, F(fe, C.MAP), F(delta, C.FOURIER), alias_quirk=quirk)
        p.set_option("small_flow", 1)
        gdp, gdf, gf0 = L.gradient(fm, F(fe, C.MAP), F(delta, C.FOURIER), alias_quirk=quirk)
        close(("small f", quirk), gf0.arr.cpu().numpy(), f0, TOL[prec]["flow"])
        close(("small df", quirk), gdf.arr.cpu().numpy(), df, TOL[prec]["adj"])
        close(("small dphi", quirk), gdp.arr.cpu().numpy(), dp, TOL[prec]["grad"])
        close(("small vs staged dphi", quirk), gdp.arr.cpu().numpy(), sdp.arr.cpu().numpy(), TOL[prec]["grad"])
        assert bool(torch.equal(gdp.arr, sdp.arr)) == (not has_kernel)


def test_small_delta_flow_result_does_not_depend_on_the_batch(camb):
    import cmblensing_jl_amd as C
    Ny, Nx, P, B = 64, 64, 2, 4
    oproj, simf, simp = sims(camb, Ny, Nx, P, B)
    f = simf(1).astype(np.float32)
    phi = simp(2, B).astype(np.float32)
    delta = O.rfft2(simf(7)).astype(np.complex64)
    p = C.ProjLambert(Ny, Nx, 2.0, torch.float32)
    F = lambda a, b: C.Field(p, p.tensor(a), b)
    allb = C.LenseFlow(p, 7)(F(phi, C.MAP)).gradient(C.FLOW_FWD, F(f, C.MAP), F(delta, C.FOURIER))
    for b in (1, 3):
        one = C.LenseFlow(p, 7)(F(phi[b:b + 1], C.MAP)).gradient(C.FLOW_FWD, F(f[b:b + 1], C.MAP), F(delta[b:b + 1], C.FOURIER))
        for x, y in zip(allb, one):
            assert torch.equal(x.arr[b:b + 1], y.arr)
