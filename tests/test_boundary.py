"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol that
include/cmblens.h declares, fails loudly without a GPU, and the product never touches the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cmblensing.jl_amd")


@pytest.fixture(scope="module")
def libpath():
    import __graft_entry__ as g
    return g.build()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "cmblens.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cmbl_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(libpath):
    syms = declared_symbols()
    assert len(syms) >= 30
    lib = ctypes.CDLL(libpath)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    from cmblensing_jl_amd.lib import SYMBOLS
    assert sorted(SYMBOLS) == syms                     # the ctypes binding covers exactly the header


def test_exports_are_extern_c(libpath):
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    for s in declared_symbols():
        assert s in exported, s


def test_no_gpu_fails_loudly(libpath):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = ctypes.CDLL(libpath)
    lib.cmbl_last_error.restype = ctypes.c_char_p
    h = ctypes.c_void_p()
    rc = lib.cmbl_ctx_create(64, 64, ctypes.c_double(1.0), 0, 0, None, ctypes.byref(h))
    assert rc == 3 and b"no HIP device" in lib.cmbl_last_error()          # CMBL_ERR_HIP, no fallback
    import cmblensing_jl_amd as C
    with pytest.raises(RuntimeError):
        C.ProjLambert(64, 64)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src and "oracle." not in src.replace("oracle.quadratic_estimate", ""), f
    # importing the product does not pull the oracle in
    code = "import sys; sys.path.insert(0, %r); import cmblensing_jl_amd; assert 'oracle' not in sys.modules" % ROOT
    subprocess.run([sys.executable, "-c", code], check=True)


def test_host_operator_algebra_matches_oracle():
    """sim.HarmOp (product, host side) restates the same BlockDiagIEB algebra as the oracle (independent code)."""
    import numpy as np
    import oracle as O
    import cmblensing_jl_amd as C
    rng = np.random.default_rng(0)
    a, d, e = rng.random((3, 8, 5)) + 1.0
    b = 0.3 * rng.random((8, 5))
    H = C.HarmOp([a, b, b, d, e])
    Ho = O.HarmOp(3, te=(a, b, b.copy(), d), bb=e)
    for got, want in ((H.pinv(), Ho.pinv()), (H.sqrt(), Ho.sqrt()), (H @ H.pinv(), Ho @ Ho.pinv()), ((H + 0.5).scale(2.0), (Ho + 0.5).scale(2.0))):
        np.testing.assert_allclose(got.p, np.stack(list(want.te) + [want.bb]), rtol=1e-13)
    class P:                                             # minimal geometry stand-in
        lam = np.array([1, 2, 2, 2, 1.0])
    np.testing.assert_allclose(H.logdet(P), Ho.logdet(type("Q", (), {"lam": P.lam, "Ny": 8, "Nx": 8})()), rtol=1e-12)
    # Cls / noise / lowpass agree with the oracle's
    for x in (0.0, 1.5, 2.0, 777.3, 3000.0, 3000.1):
        np.testing.assert_allclose(C.lowpass(3000)(np.array([x])), O.lowpass(3000)(np.array([x])), equal_nan=True)
    np.testing.assert_allclose(C.noise_cls(3, 100, 3, 4000)["EE"].cl, O.noise_cls(3, 100, 3, 4000)["EE"].cl)
    np.testing.assert_allclose(C.beam_cls(3.0, 4000).cl, O.beam_cls(3.0, 4000).cl)


def test_every_context_option_is_documented_in_the_header():
    """cmbl_ctx_set_option accepts exactly the names Ctx::opt_ptr knows (csrc/engine.hpp); include/cmblens.h must list each of them"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    eng = open(os.path.join(root, "cmblensing.jl_amd", "csrc", "engine.hpp"), encoding="utf-8").read()
    hdr = open(os.path.join(root, "include", "cmblens.h"), encoding="utf-8").read()
    names = re.findall(r'if \(k == "(\w+)"\) return &opts\.', eng)
    assert len(names) >= 20
    missing = [n for n in names if f'"{n}"' not in hdr]
    assert not missing, missing

def test_plan_length_list_of_the_gpu_tests_is_the_header_s():
    """tests/test_gpu_anysize.py::CT_LIST (every length against NumPy and the run-time plans) and tools/gpu_ct_sweep.py::CT are the compile-time-plan
    list of csrc/kernels_ct.hpp (CMBL_CT_LIST_A + _B): a length added to the header without its tests fails here"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "cmblensing.jl_amd", "csrc", "kernels_ct.hpp")).read()
    lens = []
    for half in ("A", "B"):
        m = re.search(r"#define CMBL_CT_LIST_%s\(X\)(.*)" % half, hdr)
        lens += [int(v) for v in re.findall(r"X\((\d+)\)", m.group(1))]
    assert len(lens) == len(set(lens)) and len(lens) >= 20
    for path, name in (("tests/test_gpu_anysize.py", "CT_LIST"), ("tools/gpu_ct_sweep.py", "CT")):
        src = open(os.path.join(root, path)).read()
        m = re.search(r"^%s = \(([0-9, ]+)\)" % name, src, re.M)
        assert m, path
        assert sorted(int(v) for v in m.group(1).split(",")) == sorted(lens), path
    for n in lens:                                                       # what the kernels assume: 4 | N (tiled blocks), factors 2, 3, 5 only
        k = n
        for pr in (2, 3, 5):
            while k % pr == 0: k //= pr
        assert k == 1 and n % 4 == 0, n
