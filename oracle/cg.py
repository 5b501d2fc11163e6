"""Preconditioned conjugate gradient, restated from src/numerical_algorithms.jl:73-134.

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import numpy as np

__all__ = ["conjugate_gradient"]


def conjugate_gradient(Minv, A, b, x0, dot, nsteps, tol):
    """`conjugate_gradient(M, A, b, x; nsteps, tol)`.

    Minv(r) = M \\ r ; A(p) = A*p ; dot(a,b) -> per-batch vector (length B).
    Fields are arrays with the batch on axis 0.  Returns (bestx, history) where history is
    a list of (i, res) exactly as `history_keys=(:i,:res)` records them (:85,:96,:116-118).
    Quirks kept: `res = dot(r,z)` (not ‖r‖²); stop when `all(res < tol)` (absolute);
    the best-`res` iterate is returned (:110-112,:133); A may be negative definite (α < 0).
    """
    def bc(s):  # per-batch scalar -> broadcastable over (B,P,Nx,Nyh)
        return np.asarray(s).reshape((-1,) + (1,) * (b.ndim - 1))

    T = b.real.dtype.type
    x = x0
    r = b - A(x)
    z = Minv(r)
    p = z
    res = np.asarray(dot(r, z), dtype=T)
    assert not np.any(np.isnan(res))
    bestres, bestx = res, x
    hist = [(1, res.copy())]
    for i in range(2, nsteps + 1):
        Ap = A(p)
        alpha = (res / dot(p, Ap)).astype(T)
        x = x + bc(alpha) * p
        r = r - bc(alpha) * Ap
        z = Minv(r)
        res2 = np.asarray(dot(r, z), dtype=T)
        p = z + bc((res2 / res).astype(T)) * p
        res = res2
        if np.all(res < bestres):
            bestres, bestx = res, x
        hist.append((i, res.copy()))
        if np.all(res < tol):
            break
    return bestx, hist
