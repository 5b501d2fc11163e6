"""Parity assertions that know what was measured.

`close(what, got, want, tol)` asserts relative L2 error < tolerance; `scalars_close` does the same for per-batch scalars (max relative
error).  The tolerance is the SMALLER of the literal `tol` at the call site (the bound of that class of comparison) and 3 x the error
this very comparison showed on MI355X when tests/golden/parity_measured.json was last regenerated (every parity
tolerance <= 3 x measured, so that a precision regression of the FFT twiddle scheme, a wrong constant in the last bits, ... fails
although it would sit inside the class bound).  Keys of the table: pytest node id | what # occurrence.  The assertion message carries
the measured-then value, the value now and the tolerance.  Comparisons without an entry (new tests, new parametrisations) use the
literal bound until the table is regenerated:

    CMBL_PARITY_LOG=gpurun_out/parity.jsonl python -m pytest tests -m gpu          (on the GPU box; every comparison appends a line)
    python tools/parity_report.py gpurun_out/parity.jsonl --json > tests/golden/parity_measured.json
    python tools/parity_report.py gpurun_out/parity.jsonl --collapse > profiles/rNN_parity_measured.txt

Regenerating is a deliberate, reviewed act (the diff shows every error that moved).  Floor 1e-12: double-precision comparisons
measure 1e-14..1e-13, which is rounding noise."""
import collections
import json
import os

import numpy as np

FACTOR, FLOOR = 3.0, 1e-12
_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_measured.json")
try:
    MEASURED = {} if os.environ.get("CMBL_PARITY_NO_TABLE") else json.load(open(_TABLE))      # NO_TABLE: re-measure against the class bounds
except OSError:
    MEASURED = {}
_seen = collections.Counter()


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def _key(what):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    k = f"{test}|{what}"
    n = _seen[k]
    _seen[k] += 1
    return f"{k}#{n}"


def bound(key, tol):
    """(effective tolerance, error measured when the table was made or None)"""
    m = MEASURED.get(key)
    if m is None or not m > 0:
        return tol, m
    return min(tol, max(FACTOR * m, FLOOR)), m


def _record(key, err, tol):
    path = os.environ.get("CMBL_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"key": key, "err": err, "tol": tol}) + "\n")


def _msg(what, kind, e, tol, lit, m):
    then = "no entry in tests/golden/parity_measured.json" if m is None else f"measured {m:.3e} when the table was made"
    return f"{what}: {kind} {e:.3e} >= tolerance {tol:.2e} (class bound {lit:.1e}; {then})"


def close(what, got, want, tol):
    key = _key(what)
    e = rel(got, want)
    eff, m = bound(key, tol)
    _record(key, e, eff)
    assert e < eff, _msg(what, "relative L2 error", e, eff, tol, m)
    return e


def scalars_close(what, got, want, rtol, atol=0.0):
    key = _key(what)
    got, want = np.atleast_1d(np.asarray(got, float)), np.atleast_1d(np.asarray(want, float))
    e = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
    eff, m = bound(key, rtol)
    _record(key, e, eff)
    assert np.all(np.abs(got - want) <= eff * np.abs(want) + atol), _msg(what, "max relative error", e, eff, rtol, m) + f"; got {got}, want {want}"
    return e


def sample_close(what, err, tol):
    """for comparisons that form their own error figure (golden-sample checks)"""
    key = _key(what)
    eff, m = bound(key, tol)
    _record(key, float(err), eff)
    assert err < eff, _msg(what, "relative L2 error", float(err), eff, tol, m)
