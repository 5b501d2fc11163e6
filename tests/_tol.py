"""Parity assertions that report what they measured.

`close(what, got, want, tol)` asserts relative L2 error < tol; `scalars_close` does the same for per-batch scalars (max relative error).
Both append {test, what, err, tol} to the JSON-lines file named by CMBL_PARITY_LOG (tools/parity_report.py turns a GPU run's log into
profiles/rNN_parity_measured.txt), and the assertion message carries the measured value, so a failure -- or a tolerance that has
become loose -- shows by how much.  Rule for every tolerance of the GPU suite: tol <= 3 x the largest error
measured on MI355X for that class of comparison (fp32), resp. a fixed 1e-10..1e-12 floor in fp64 where the measured 1e-14..1e-13 is
rounding noise that moves with the size."""
import json
import os

import numpy as np


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def _record(what, err, tol):
    path = os.environ.get("CMBL_PARITY_LOG")
    if not path:
        return
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    with open(path, "a") as f:
        f.write(json.dumps({"test": test, "what": str(what), "err": err, "tol": tol}) + "\n")


def close(what, got, want, tol):
    e = rel(got, want)
    _record(what, e, tol)
    assert e < tol, f"{what}: relative L2 error {e:.3e} >= tolerance {tol:.1e}"
    return e


def scalars_close(what, got, want, rtol, atol=0.0):
    got, want = np.atleast_1d(np.asarray(got, float)), np.atleast_1d(np.asarray(want, float))
    e = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
    _record(what, e, rtol)
    assert np.all(np.abs(got - want) <= rtol * np.abs(want) + atol), f"{what}: max relative error {e:.3e} > rtol {rtol:.1e} (atol {atol:.1e}); got {got}, want {want}"
    return e
