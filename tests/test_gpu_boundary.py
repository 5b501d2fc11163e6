"""Boundary promises of include/cmblens.h that had no test (VERDICT r04 "limits without tests"):

  * "Independent contexts may be driven from different host threads" (cmblens.h:23): two host threads, each with its OWN context, flow and
    dataset, run the hot path concurrently; every result equals the same computation run alone, bit for bit.  The error text is per thread.
  * CMBL_ERR_ALLOC: a request the device cannot satisfy -- directly (cmbl_device_malloc) and inside an operation (the per-stage product
    scratch of a delta flow with 512 RK steps on 64 batch slots) -- returns the status code, and the context stays usable afterwards.

Reference counterparts: one GPU worker per process / thread (src/util_parallel.jl:73-102); Julia's OutOfGPUMemoryError surfaces as an
exception and leaves the session alive."""
import ctypes
import threading

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from bench import synthetic_cls


def _workload(C, N, pol, seed):
    s = C.load_sim(3.0, N, pol, synthetic_cls(), T=torch.float32, pixel_mask=dict(pad_deg=0.4, apod_deg=0.4), seeds=(seed, seed + 1, seed + 2))
    ds = s["ds"]
    fo, po = ds.mix(s["f"], s["phi"])
    return s, ds, fo, po


def test_two_host_threads_two_contexts():
    import cmblensing_jl_amd as C
    work = [_workload(C, 256, "P", 11), _workload(C, 128, "IP", 21)]              # different sizes, pols: different kernels, LDS limits, streams
    def run(k, n):
        s, ds, fo, po = work[k]
        out = []
        for _ in range(n):
            lp, gf, gp = ds.gradient_logpdf_mixed(fo, po)
            out.append((np.asarray(lp).copy(), gf.arr.clone(), gp.arr.clone()))
        return out
    alone = [run(0, 1)[0], run(1, 1)[0]]
    res, errs = [None, None], []
    def body(k):
        try:
            res[k] = run(k, 25)
        except Exception as e:                                                  # noqa: BLE001
            errs.append((k, repr(e)))
    th = [threading.Thread(target=body, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for k in range(2):
        for lp, gf, gp in res[k]:
            assert np.array_equal(lp, alone[k][0]) and torch.equal(gf, alone[k][1]) and torch.equal(gp, alone[k][2]), f"thread {k}: result differs from the run alone"


def test_error_text_is_per_thread():
    import cmblensing_jl_amd as C
    lib = C.load_library()
    s, ds, fo, po = _workload(C, 64, "I", 5)
    p = s["proj"]
    seen = {}
    def bad():
        v = ctypes.c_int(0)
        rc = lib.cmbl_ctx_get_option(p._h, b"no_such_option", ctypes.byref(v))
        seen["rc"], seen["msg"] = rc, lib.cmbl_last_error().decode()
    t = threading.Thread(target=bad); t.start(); t.join()
    assert seen["rc"] == 1 and "no_such_option" in seen["msg"]                   # CMBL_ERR_ARG, text available in the thread that failed
    assert "no_such_option" not in lib.cmbl_last_error().decode()                # ... and not in this one
    ds.gradient_logpdf_mixed(fo, po)                                             # the context was not disturbed


def test_err_alloc_is_reported_and_the_context_survives():
    import cmblensing_jl_amd as C
    from cmblensing_jl_amd.lib import CmblError
    lib = C.load_library()
    s, ds, fo, po = _workload(C, 256, "P", 7)
    p = s["proj"]
    before = ds.gradient_logpdf_mixed(fo, po)
    # (1) directly: more than the device has (2^46 bytes = 64 TiB)
    ptr = ctypes.c_void_p(1)
    rc = lib.cmbl_device_malloc(p._h, ctypes.c_size_t(1 << 46), ctypes.byref(ptr))
    assert rc == 6 and ptr.value is None and "hipMalloc" in lib.cmbl_last_error().decode()       # CMBL_ERR_ALLOC, *out = NULL
    # (2) inside an operation: a delta flow keeps 4n x 2 product maps per slice -- n = 512 RK steps (the maximum) on 64 batch slots of 512^2 QU
    #     asks for 2048 x 2 x 128 MiB = 512 GiB of scratch, more than the 288 GB of the part
    p2 = C.ProjLambert(512, 512, 2.0, torch.float32)
    rng = np.random.default_rng(3)
    F = lambda a, b: C.Field(p2, p2.tensor(a), b)
    phi = F(1e-6 * rng.standard_normal((1, 1, 512, 512)), C.MAP)
    fs = F(rng.standard_normal((1, 2, 512, 512)), C.MAP)
    L7 = C.LenseFlow(p2, 7)(phi)
    ref = (L7 * fs).arr.clone()
    fbig = F(rng.standard_normal((64, 2, 512, 512)), C.MAP)
    Lbig = C.LenseFlow(p2, 512)(phi)
    with pytest.raises(CmblError) as ei:
        Lbig.gradient(C.FLOW_FWD, fbig, fbig.to(C.FOURIER))
    assert ei.value.code == 6 and "hipMalloc" in str(ei.value), ei.value
    del Lbig, fbig
    # the contexts, their flows and the dataset keep working, with the same results as before the failures
    assert torch.equal((L7 * fs).arr, ref)
    after = ds.gradient_logpdf_mixed(fo, po)
    assert np.array_equal(np.asarray(before[0]), np.asarray(after[0])) and torch.equal(before[1].arr, after[1].arr) and torch.equal(before[2].arr, after[2].arr)


@pytest.mark.parametrize("N,pol", [(128, "P"), (256, "IP"), ((512, 64), "I"), ((64, 512), "P")])
def test_small_map_launch_geometry_changes_no_result(N, pol):
    """Round 5's occupancy-aware launch geometry (options `occupancy_tiles`, `fill_target`, `row_fill_target`) decides WHICH
    workgroup computes a column / row, never the arithmetic on it: every setting gives bit-identical results -- also when the flow is split
    into one launch chain per pol slice (the tile choice follows the slices of the whole operation, not of one chain: a regression caught by
    test_slice_streams_give_identical_results at 1024² while this was being built)."""
    import cmblensing_jl_amd as C
    s, ds, fo, po = _workload(C, N, pol, 31)
    p = s["proj"]
    fm = s["f"].to(C.MAP)
    gl = fm.to(C.FOURIER)
    def run():
        L = ds.L(s["phi"])
        ft = L * fm
        dphi, df, _ = L.gradient(C.FLOW_FWD, ft, gl)
        lp, gf, gp = ds.gradient_logpdf_mixed(fo, po)
        return [ft.arr.clone(), (L.adjoint * gl).arr.clone(), dphi.arr.clone(), df.arr.clone(), gf.arr.clone(), gp.arr.clone(), torch.tensor(np.asarray(lp))]
    saved = {k: p.get_option(k) for k in ("occupancy_tiles", "fill_target", "row_fill_target", "slice_streams", "slice_streams_min_pix")}
    try:
        p.set_option("occupancy_tiles", 0)
        ref = run()
        for occ, ft_, rt_, ss, mp in [(3, 0, 0, saved["slice_streams"], saved["slice_streams_min_pix"]), (1, 0, 0, 4, 0), (2, 0, 0, 4, 0), (3, 0, 0, 4, 0),
                                      (3, 4096, 4096, 1, 0), (3, 64, 64, 4, 0)]:
            for k, v in (("occupancy_tiles", occ), ("fill_target", ft_), ("row_fill_target", rt_), ("slice_streams", ss), ("slice_streams_min_pix", mp)):
                p.set_option(k, v)
            got = run()
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), (occ, ft_, rt_, ss, mp)
    finally:
        for k, v in saved.items():
            p.set_option(k, v)
