"""CPU oracle for the flat-sky lensing hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy restatement of the algorithms on the hot path of
marius311/CMBLensing.jl (reference @ v0.10.1), written by reading the Julia
sources file-by-file; every function cites the reference `file:line` it follows.

Rules (enforced by tests/test_boundary.py::test_product_does_not_import_oracle):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import anything from `oracle/` -- and only as the checker / reported baseline;
  * the product package (`cmblensing.jl_amd/`) never imports, links or executes it.

Pinning status
--------------
The reference is 100 % Julia; `julia` is not installed in the build image and none of
its dependencies (FFTW.jl, Zygote, ...) are present, so the reference itself can not be
run to generate outputs, and its test-suite (test/runtests.jl) holds NO numeric golden
vectors for this path -- every check there is a seeded self-consistency property or a
small known-answer identity.  The oracle is therefore pinned by exactly those:
    test/runtests.jl:116-131  basis round trips
    test/runtests.jl:249-285  logdet / tr known answers and dense-fft identities
    test/runtests.jl:289-295  EB-diagonal operator as QU blocks
    test/runtests.jl:533-581  LenseFlow adjoint identity, finite-difference gradient
    test/runtests.jl:585-621  logpdf == mixed logpdf, finite-difference gradients
(tests/test_oracle_*.py re-run them at the reference's sizes and tolerances), and by
the reference's own data fixture dat/default_camb_Cls.jld2 (decoded by
tools/extract_cls.py into tests/golden/camb_cls.npz).
**Parity against numeric outputs of the reference itself is unpinned** (no such outputs
exist anywhere without a Julia runtime).

Array convention: Julia `arr[y,x,p,b]` (column-major, `src/proj_cartesian.jl:13-36`) is
the NumPy C-order array `a[b,p,x,y]` -- identical memory.  Half-plane Fourier arrays are
`a[b,p,x,ky]`, ky = 0..Ny//2 (`rfft` over Julia dims (1,2) == `rfft2(axes=(-2,-1))`).
"""
from .flatsky import *          # noqa: F401,F403
from .lenseflow import *        # noqa: F401,F403
from .cg import *               # noqa: F401,F403
from .dataset import *          # noqa: F401,F403
from .quadratic_estimate import *   # noqa: F401,F403
from .maximization import *     # noqa: F401,F403
from .sampling import *         # noqa: F401,F403
