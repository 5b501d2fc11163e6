"""ctypes loader for libcmblens_hip.so and the in-tree build recipe."""
import ctypes
import os
import subprocess
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/cmblens.h declares (tests/test_boundary.py checks the .so exports exactly these)
SYMBOLS = [
    "cmbl_last_error", "cmbl_version", "cmbl_abi_version", "cmbl_ctx_set_option", "cmbl_ctx_get_option", "cmbl_ctx_create", "cmbl_ctx_destroy",
    "cmbl_ctx_synchronize", "cmbl_ctx_geometry_host", "cmbl_prof_enable", "cmbl_prof_reset", "cmbl_prof_count", "cmbl_prof_name", "cmbl_prof_get",
    "cmbl_rfft", "cmbl_irfft", "cmbl_convert", "cmbl_diag_apply", "cmbl_blockdiag_ieb_apply", "cmbl_dot", "cmbl_logdet", "cmbl_lenseflow_create",
    "cmbl_lenseflow_destroy", "cmbl_lenseflow_set_phi", "cmbl_lenseflow_apply", "cmbl_lenseflow_grad", "cmbl_dataset_create", "cmbl_dataset_destroy",
    "cmbl_dataset_set_op", "cmbl_dataset_set_data", "cmbl_dataset_set_logdet", "cmbl_max_lensing_step", "cmbl_axpby", "cmbl_qe_leg",
    "cmbl_fourier_lmul", "cmbl_map_fma", "cmbl_randn", "cmbl_gradientf_logpdf", "cmbl_wiener_cg", "cmbl_logpdf_mixed", "cmbl_grad_logpdf_mixed",
    "cmbl_hmc_step", "cmbl_map_joint_step", "cmbl_quadratic_estimate", "cmbl_norm", "cmbl_logdet_diag", "cmbl_tr_diag", "cmbl_set_sum_accuracy_mode",
    "cmbl_timer_report", "cmbl_device_malloc", "cmbl_device_free", "cmbl_copy_to_device", "cmbl_copy_to_host",
]


ABI_VERSION = 3          # CMBL_ABI_VERSION of include/cmblens.h this binding was written against


class CmblError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcmblens_hip error {code}: {msg}")
        self.code = code


def library_path():
    """In-tree library; CMBL_LIB points at an alternative build of the same sources (tile-shape experiments)."""
    return os.environ.get("CMBL_LIB") or os.path.join(_HERE, "libcmblens_hip.so")


# translation units of the library (csrc/api_decl.hpp has the map): the entry points, and per precision the typed bodies with the
# power-of-two kernels, the any-size transform launches and (tu_cty / tu_ctx, two halves of the list of lengths) their compile-time-plan kernels.
# One object each, then linked.
# (longest first: the build is a pool of as many compilers as the host has cores)
UNITS = ["tu_cty_f32_b", "tu_cty_f64_b", "tu_ctx_f32_b", "tu_ctx_f64_b", "tu_cty_f32_a", "tu_cty_f64_a", "tu_main_f32", "tu_main_f64",
         "tu_ctx_f32_a", "tu_ctx_f64_a", "tu_small_f32", "tu_small_f64", "tu_gen_f32", "tu_gen_f64", "api"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def build(force=False, verbose=False, jobs=None, extra_flags=(), out=None, objdir=None):
    """hipcc cross-compiles for gfx950 without a GPU.  In-tree output: cmblensing.jl_amd/libcmblens_hip.so

    One object per translation unit under build/obj (git-ignored), compiled `jobs` at a time (default: one per core, the longest units first), re-compiled only when a source it includes (-MMD dependency file) or the flags changed.  `extra_flags`
    / `out` / `objdir`: variant builds of the same sources (tools/devbuild.py)."""
    csrc = os.path.join(_HERE, "csrc")
    out = out or library_path()
    objdir = objdir or os.path.join(_HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = HIPCC_FLAGS + list(extra_flags)
    stamp = " ".join([hipcc] + flags)

    def stale(unit):
        obj, dep, flg = (os.path.join(objdir, unit + e) for e in (".o", ".d", ".flags"))
        if force or not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(flg)) or open(flg).read() != stamp:
            return True
        deps = open(dep).read().replace("\\\n", " ").split(":", 1)[1].split()
        t = os.path.getmtime(obj)
        return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)

    todo = [u for u in UNITS if stale(u)]
    procs, failed = [], []
    jobs = jobs or int(os.environ.get("CMBL_BUILD_JOBS", "0")) or min(len(UNITS), os.cpu_count() or 8)

    def reap(block):
        """collect the compilers that have finished; block: wait until at least one has"""
        while True:
            done = [item for item in procs if item[1].poll() is not None]
            for unit, p, log in done:
                log.close()
                procs.remove((unit, p, log))
                if p.returncode != 0:
                    failed.append(unit)
                else:
                    open(os.path.join(objdir, unit + ".flags"), "w").write(stamp)
            if done or not block or not procs:
                return
            time.sleep(0.2)

    for unit in todo:
        while len(procs) >= jobs:
            reap(True)
        obj = os.path.join(objdir, unit + ".o")
        cmd = [hipcc] + flags + ["-MMD", "-MF", os.path.join(objdir, unit + ".d"), "-c", os.path.join(csrc, unit + ".hip"), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        for e in (".flags",):
            if os.path.exists(os.path.join(objdir, unit + e)):
                os.remove(os.path.join(objdir, unit + e))
        log = open(os.path.join(objdir, unit + ".log"), "w")
        procs.append((unit, subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT), log))
    while procs:
        reap(True)
    if failed:
        msgs = "".join(f"\n--- {u} ---\n" + open(os.path.join(objdir, u + ".log")).read()[-4000:] for u in failed)
        raise RuntimeError("hipcc failed for " + ", ".join(failed) + msgs)
    objs = [os.path.join(objdir, u + ".o") for u in UNITS]
    if todo or not os.path.exists(out) or any(os.path.getmtime(o) > os.path.getmtime(out) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


def load_library():
    """Load the HIP library; fails loudly when it has not been built (no fallback path exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    pd = ctypes.POINTER(ctypes.c_double)
    lib.cmbl_last_error.restype = ctypes.c_char_p
    lib.cmbl_last_error.argtypes = []
    lib.cmbl_version.restype = ci
    lib.cmbl_abi_version.restype = ci
    lib.cmbl_abi_version.argtypes = []
    if lib.cmbl_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.cmbl_abi_version()} but this package binds version {ABI_VERSION} (include/cmblens.h): rebuild")
    lib.cmbl_prof_count.restype = ci
    lib.cmbl_prof_count.argtypes = []
    lib.cmbl_prof_name.restype = ctypes.c_char_p
    lib.cmbl_prof_name.argtypes = [ci]
    sig = {
        "cmbl_ctx_create": [ci, ci, cd, ci, ci, vp, ctypes.POINTER(vp)],
        "cmbl_ctx_destroy": [vp],
        "cmbl_ctx_set_option": [vp, ctypes.c_char_p, ci],
        "cmbl_ctx_get_option": [vp, ctypes.c_char_p, ctypes.POINTER(ci)],
        "cmbl_ctx_synchronize": [vp],
        "cmbl_ctx_geometry_host": [vp, ci, pd, ctypes.c_size_t],
        "cmbl_prof_enable": [vp, ci],
        "cmbl_prof_reset": [vp],
        "cmbl_prof_get": [vp, ci, pd, ctypes.POINTER(ctypes.c_long)],
        "cmbl_rfft": [vp, vp, vp, ci, ci],
        "cmbl_irfft": [vp, vp, vp, ci, ci],
        "cmbl_convert": [vp, ci, vp, ci, vp, ci, ci],
        "cmbl_diag_apply": [vp, ci, ci, vp, ci, vp, ci, vp, ci, ci],
        "cmbl_blockdiag_ieb_apply": [vp, vp, ci, ci, vp, ci, vp, ci],
        "cmbl_dot": [vp, ci, vp, vp, ci, ci, pd],
        "cmbl_logdet": [vp, vp, ci, pd],
        "cmbl_lenseflow_create": [vp, ci, ctypes.POINTER(vp)],
        "cmbl_lenseflow_destroy": [vp],
        "cmbl_lenseflow_set_phi": [vp, ci, vp, ci],
        "cmbl_lenseflow_apply": [vp, ci, ci, vp, ci, vp, ci, ci],
        "cmbl_lenseflow_grad": [vp, ci, vp, ci, vp, vp, ci, vp, vp, ci, ci, ci],
        "cmbl_max_lensing_step": [vp, ci, vp, vp, ci, pd],
        "cmbl_axpby": [vp, ci, pd, vp, pd, vp, vp, ci, ci],
        "cmbl_qe_leg": [vp, vp, ci, ci, ci, vp, ci],
        "cmbl_fourier_lmul": [vp, vp, ci, ci, ci, vp, ci],
        "cmbl_map_fma": [vp, vp, vp, cd, vp, ci, ci],
        "cmbl_randn": [vp, ctypes.POINTER(ctypes.c_uint64), ci, ctypes.c_uint64, vp, ctypes.c_long],
        "cmbl_dataset_create": [vp, ci, ctypes.POINTER(vp)],
        "cmbl_dataset_destroy": [vp],
        "cmbl_dataset_set_op": [vp, ci, vp, ci],
        "cmbl_dataset_set_data": [vp, vp, ci],
        "cmbl_dataset_set_logdet": [vp, cd],
        "cmbl_gradientf_logpdf": [vp, vp, vp, vp, ci, vp, ci],
        "cmbl_wiener_cg": [vp, vp, vp, vp, cd, ci, vp, pd, ctypes.POINTER(ci), ci],
        "cmbl_logpdf_mixed": [vp, vp, vp, vp, pd, ci],
        "cmbl_grad_logpdf_mixed": [vp, vp, vp, vp, pd, vp, vp, ci, ci],
        "cmbl_hmc_step": [vp, vp, vp, vp, vp, vp, pd, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64, ci, cd, ci, ci, ci, vp, pd, ctypes.POINTER(ci)],
        "cmbl_quadratic_estimate": [vp, ci, pd, pd, pd, pd, pd, ci, pd, vp, pd, ci],
        "cmbl_map_joint_step": [vp, vp, vp, vp, vp, cd, cd, cd, ci, ci, ci, vp, vp, pd, pd, ctypes.POINTER(ci), ctypes.POINTER(ci)],
        "cmbl_norm": [vp, ci, vp, ci, ci, pd],
        "cmbl_logdet_diag": [vp, ci, vp, ci, ci, pd],
        "cmbl_tr_diag": [vp, ci, vp, ci, ci, pd],
        "cmbl_set_sum_accuracy_mode": [vp, ci],
        "cmbl_timer_report": [vp, ctypes.c_char_p, ctypes.c_size_t],
        "cmbl_device_malloc": [vp, ctypes.c_size_t, ctypes.POINTER(vp)],
        "cmbl_device_free": [vp, vp],
        "cmbl_copy_to_device": [vp, vp, vp, ctypes.c_size_t],
        "cmbl_copy_to_host": [vp, vp, vp, ctypes.c_size_t],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ci
    _LIB = lib
    return lib


def check(code):
    if code != 0:
        raise CmblError(code, load_library().cmbl_last_error().decode("utf-8", "replace"))
