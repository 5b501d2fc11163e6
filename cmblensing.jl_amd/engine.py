"""Host-side mirror of the reference operator surface for the hot path, on top of the C ABI.

Names follow CMBLensing.jl: `ProjLambert` (src/proj_lambert.jl:48-75), `LenseFlow` / cached flow with
`L*f`, `L\\f`, `L'*f`, `L'\\f` and the pullbacks (src/lenseflow.jl, src/flowops.jl), `BaseDataSet` with
`gradientf_logpdf`, `argmaxf_logpdf`, `logpdf(Mixed(ds))` and its gradient (src/dataset.jl,
src/maximization.jl).  Device memory is held in torch tensors (plumbing only); every computation is a call
into libcmblens_hip.so.  Tensor layouts are the reference's: map (B,P,Nx,Ny) real == Julia (Ny,Nx,P,B);
Fourier (B,P,Nx,Ny//2+1) complex.
"""
import ctypes

import numpy as np
import torch

from .lib import load_library, check

MAP, FOURIER, HARMONIC = 0, 1, 2
FLOW_FWD, FLOW_INV, FLOW_ADJ, FLOW_INVADJ = 0, 1, 2, 3
DIAG_MUL, DIAG_DIV_NAN2ZERO = 1, 3
(OP_CF_INV, OP_CN_INV, OP_B, OP_MF, OP_D, OP_D_INV, OP_PRECOND_INV, OP_CPHI_INV, OP_G_INV, OP_MPIX) = range(10)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def reference_exact():
    """CMBL_REFERENCE_EXACT=1: the two defaults in which the engine deviates from the reference as written become the reference's --
    the δϕ velocity with the in-place aliasing of src/lenseflow.jl:198-200 (`alias_quirk=True`) and plain sums in the
    working precision (`sum_accuracy_mode = nothing`, src/util.jl:288-316; read by the library when a context is created).  This
    is the mode to run when results are compared with output of the Julia package itself (tests/golden/ref_*.npy)."""
    import os
    return os.environ.get("CMBL_REFERENCE_EXACT", "0") not in ("", "0")


class ProjLambert:
    """Context: geometry + FFT tables + stream.  `T` is torch.float32 or torch.float64."""

    def __init__(self, Ny, Nx, theta_pix=1.0, T=torch.float32, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("cmblensing_jl_amd needs a HIP device (no CPU fallback)")
        self.lib = load_library()
        self.Ny, self.Nx, self.Nyh, self.theta_pix = int(Ny), int(Nx), int(Ny) // 2 + 1, float(theta_pix)
        self.T = T
        self.CT = torch.complex64 if T == torch.float32 else torch.complex128
        self.device = torch.device("cuda", device)
        self._h = ctypes.c_void_p()
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.cmbl_ctx_create(self.Ny, self.Nx, self.theta_pix, 0 if T == torch.float32 else 1,
                                       device, ctypes.c_void_p(stream), ctypes.byref(self._h)))
        self._geom = {}

    def __del__(self):
        try:
            if self._h:
                self.lib.cmbl_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # geometry (host numpy, float64 copies of the T-precision values the kernels use)
    def _g(self, which, shape):
        if which not in self._geom:
            n = int(np.prod(shape))
            out = np.empty(n, dtype=np.float64)
            check(self.lib.cmbl_ctx_geometry_host(self._h, which, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n))
            self._geom[which] = out.reshape(shape)
        return self._geom[which]

    lx = property(lambda s: s._g(0, (s.Nx,)))
    ly = property(lambda s: s._g(1, (s.Nyh,)))
    lam = property(lambda s: s._g(2, (s.Nyh,)))
    sin2phi = property(lambda s: s._g(3, (s.Nx, s.Nyh)))
    cos2phi = property(lambda s: s._g(4, (s.Nx, s.Nyh)))
    lmag = property(lambda s: s._g(5, (s.Nx, s.Nyh)))

    @property
    def Opix(self):
        return np.deg2rad(self.theta_pix / 60) ** 2

    @property
    def nyquist(self):
        return np.pi / np.deg2rad(self.theta_pix / 60)

    @property
    def lmax(self):
        return int(round(np.ceil(np.sqrt(2) * self.nyquist) + 1))       # src/dataset.jl:232

    def synchronize(self):
        check(self.lib.cmbl_ctx_synchronize(self._h))

    # ---- optional per-kernel-class HIP-event timing (cmbl_prof_*)
    def prof_enable(self, on=True):
        check(self.lib.cmbl_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        check(self.lib.cmbl_prof_reset(self._h))

    def prof_table(self):
        """{kernel class: (total_ms, launches)} accumulated since the last reset"""
        out = {}
        for k in range(self.lib.cmbl_prof_count()):
            ms, n = ctypes.c_double(0), ctypes.c_long(0)
            check(self.lib.cmbl_prof_get(self._h, k, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                out[self.lib.cmbl_prof_name(k).decode()] = (ms.value, n.value)
        return out

    # ---- tensors
    def empty(self, basis, P, B):
        if basis == MAP:
            return torch.empty((B, P, self.Nx, self.Ny), dtype=self.T, device=self.device)
        return torch.empty((B, P, self.Nx, self.Nyh), dtype=self.CT, device=self.device)

    def tensor(self, a, basis=None):
        """numpy / torch -> contiguous device tensor of the context's precision"""
        if torch.is_tensor(a) and a.device.type != "cpu":
            dt = self.CT if a.is_complex() else self.T
            return a.to(device=self.device, dtype=dt).contiguous()
        # host data: convert with NumPy (single-threaded, 0.2 ms for a 1024x513 plane), never with a torch CPU op -- torch's CPU
        # thread pool (128 threads on the GPU boxes) stalls for 20-70 ms when woken between GPU calls (measured: it made one HMC
        # step 1.0 s instead of 0.17 s)
        arr = a.numpy() if torch.is_tensor(a) else np.asarray(a)
        npdt = {torch.float32: np.float32, torch.float64: np.float64, torch.complex64: np.complex64, torch.complex128: np.complex128}[
            self.CT if np.iscomplexobj(arr) else self.T]
        arr = np.ascontiguousarray(arr, dtype=npdt)
        return torch.from_numpy(arr).to(device=self.device)

    def _check(self, t, basis):
        B, P = t.shape[0], t.shape[1]
        want = (B, P, self.Nx, self.Ny if basis == MAP else self.Nyh)
        if tuple(t.shape) != want or t.dtype != (self.T if basis == MAP else self.CT) or not t.is_contiguous() or t.device != self.device:
            raise ValueError(f"field tensor has shape/dtype/device {tuple(t.shape)}/{t.dtype}/{t.device}, expected {want}")
        return P, B

    # ---- basis conversion (src/proj_lambert.jl:245-300)
    def convert(self, t, basis_in, basis_out):
        P, B = self._check(t, basis_in)
        out = self.empty(basis_out, P, B)
        check(self.lib.cmbl_convert(self._h, basis_in, _ptr(t), basis_out, _ptr(out), P, B))
        return out

    def rfft(self, m):
        return self.convert(m, MAP, FOURIER)

    def irfft(self, f):
        return self.convert(f, FOURIER, MAP)

    # ---- DiagOp * / \ (src/specialops.jl:9-10)
    def diag_apply(self, diag, t, basis_diag, basis_in, basis_out=None, kind=DIAG_MUL):
        P, B = self._check(t, basis_in)
        basis_out = basis_in if basis_out is None else basis_out
        d = self.tensor(diag)
        out = self.empty(basis_out, P, B)
        if d.shape[0] == 5 and P == 3:
            check(self.lib.cmbl_blockdiag_ieb_apply(self._h, _ptr(d), 0, basis_in, _ptr(t), basis_out, _ptr(out), B))
        else:
            assert tuple(d.shape) == (P, self.Nx, self.Nyh), d.shape
            check(self.lib.cmbl_diag_apply(self._h, kind, basis_diag, _ptr(d), basis_in, _ptr(t), basis_out, _ptr(out), P, B))
        return out

    # ---- reductions (src/proj_lambert.jl:318-342)
    def dot(self, a, b, basis):
        P, B = self._check(a, basis)
        self._check(b, basis)
        out = (ctypes.c_double * B)()
        check(self.lib.cmbl_dot(self._h, basis, _ptr(a), _ptr(b), P, B, out))
        return np.array(out[:])

    # ---- helpers for the drivers (CG / line-search / leapfrog axpys, quadratic-estimate legs)
    def axpby(self, a, x, b=None, y=None, basis=None):
        """a·x + b·y with scalars or per-batch vectors (src/batching.jl BatchedReal broadcasting)"""
        arr = x.arr if isinstance(x, Field) else x
        basis = x.basis if isinstance(x, Field) else basis
        P, B = self._check(arr, basis)
        av = (ctypes.c_double * B)(*np.broadcast_to(np.asarray(a, float), (B,)))
        out = self.empty(basis, P, B)
        if y is None:
            check(self.lib.cmbl_axpby(self._h, basis, av, _ptr(arr), None, None, _ptr(out), P, B))
        else:
            yarr = y.to(basis).arr if isinstance(y, Field) else y
            self._check(yarr, basis)
            bv = (ctypes.c_double * B)(*np.broadcast_to(np.asarray(b, float), (B,)))
            check(self.lib.cmbl_axpby(self._h, basis, av, _ptr(arr), bv, _ptr(yarr), _ptr(out), P, B))
        return Field(self, out, basis) if isinstance(x, Field) else out

    def qe_leg(self, fl, n, p1, p2):
        """QE_leg (src/quadratic_estimate.jl:89-91): Fourier S0 tensor (B,1,Nx,Nyh) -> map tensor"""
        P, B = self._check(fl, FOURIER)
        assert P == 1
        out = self.empty(MAP, 1, B)
        check(self.lib.cmbl_qe_leg(self._h, _ptr(fl), n, p1, p2, _ptr(out), B))
        return out

    def fourier_lmul(self, m, p1, p2, take_abs=False):
        P, B = self._check(m, MAP)
        assert P == 1
        out = self.empty(FOURIER, 1, B)
        check(self.lib.cmbl_fourier_lmul(self._h, _ptr(m), p1, p2, 1 if take_abs else 0, _ptr(out), B))
        return out

    def map_fma(self, a, b, scale=1.0, out=None):
        P, B = self._check(a, MAP)
        self._check(b, MAP)
        acc = out is not None
        if out is None:
            out = self.empty(MAP, P, B)
        check(self.lib.cmbl_map_fma(self._h, _ptr(a), _ptr(b), float(scale), _ptr(out), 1 if acc else 0, P * B))
        return out

    def randn(self, seeds, stream, P):
        """White-noise maps (len(seeds), P, Nx, Ny): slot b ~ N(0,1) from Philox4x32-10 keyed by seeds[b], sequence `stream`
        (`randn!`, src/base_fields.jl:169-170).  Generated on the device."""
        seeds = [int(s) & 0xFFFFFFFFFFFFFFFF for s in (seeds if isinstance(seeds, (list, tuple)) else [seeds])]   # exact 64-bit ints
        out = self.empty(MAP, P, len(seeds))
        arr = (ctypes.c_uint64 * len(seeds))(*seeds)
        check(self.lib.cmbl_randn(self._h, arr, len(seeds), int(stream) & 0xFFFFFFFFFFFFFFFF, _ptr(out), P * self.Nx * self.Ny))
        return out

    def logdet(self, diag):
        d = self.tensor(diag)
        d = d.reshape(-1, self.Nx, self.Nyh)
        out = (ctypes.c_double * 1)()
        check(self.lib.cmbl_logdet(self._h, _ptr(d), d.shape[0], out))
        return out[0]

    def norm(self, a, basis):
        """norm(f) = sqrt(dot(f, f)) (src/generic.jl:373), per batch slot"""
        P, B = self._check(a, basis)
        out = (ctypes.c_double * B)()
        check(self.lib.cmbl_norm(self._h, basis, _ptr(a), P, B, out))
        return np.array(out[:])

    def logdet_diag(self, d, basis):
        """logdet(Diagonal(field)) (src/proj_lambert.jl:331-342): Map basis with the sign term, Fourier bases λ-weighted"""
        P, B = self._check(d, basis)
        out = (ctypes.c_double * B)()
        check(self.lib.cmbl_logdet_diag(self._h, basis, _ptr(d), P, B, out))
        return np.array(out[:])

    def tr_diag(self, d, basis):
        """tr(Diagonal(field)) (src/proj_lambert.jl:346-353)"""
        P, B = self._check(d, basis)
        out = (ctypes.c_double * B)()
        check(self.lib.cmbl_tr_diag(self._h, basis, _ptr(d), P, B, out))
        return np.array(out[:])

    def set_sum_accuracy_mode(self, mode):
        """`set_sum_accuracy_mode!` (src/util.jl:288-292): None | "working" (the reference's default, plain sum in T), "float64"
        (the engine's default), "kahan"."""
        m = {None: 0, "working": 0, "float64": 1, float: 1, np.float64: 1, "kahan": 2}[mode]
        check(self.lib.cmbl_set_sum_accuracy_mode(self._h, m))

    def set_option(self, name, value):
        """behaviour switch of this context (`cmbl_ctx_set_option`, include/cmblens.h): "slice_streams", "pcache", "fused_harm", ...
        Returns the previous value."""
        old = ctypes.c_int(0)
        check(self.lib.cmbl_ctx_get_option(self._h, name.encode(), ctypes.byref(old)))
        check(self.lib.cmbl_ctx_set_option(self._h, name.encode(), int(value)))
        return old.value

    def get_option(self, name):
        v = ctypes.c_int(0)
        check(self.lib.cmbl_ctx_get_option(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    def timer_report(self):
        """text table of the per-kernel-class HIP-event timings (cmbl_timer_report)"""
        n = self.lib.cmbl_timer_report(self._h, None, 0)
        if n < 0:
            check(-n)
        buf = ctypes.create_string_buffer(n + 1)
        self.lib.cmbl_timer_report(self._h, buf, n + 1)
        return buf.value.decode()


class Field:
    """A field tensor tagged with its basis (tiny stand-in for BaseField{B,...}, src/base_fields.jl:14-21)."""

    def __init__(self, proj, arr, basis):
        self.proj, self.arr, self.basis = proj, arr, basis

    def to(self, basis):
        return self if basis == self.basis else Field(self.proj, self.proj.convert(self.arr, self.basis, basis), basis)

    def dot(self, other):
        o = other.to(self.basis)
        return self.proj.dot(self.arr, o.arr, self.basis)

    def norm(self):
        return self.proj.norm(self.arr, self.basis)

    def __add__(self, o):
        return self.proj.axpby(1.0, self, 1.0, o)

    def __sub__(self, o):
        return self.proj.axpby(1.0, self, -1.0, o)

    def __rmul__(self, s):
        return self.proj.axpby(s, self)

    def __neg__(self):
        return self.proj.axpby(-1.0, self)


class _Adjoint:
    def __init__(self, L):
        self.L = L

    def __mul__(self, g):            # L' * g
        return self.L._apply(FLOW_ADJ, g)

    def ldiv(self, g):               # L' \ g
        return self.L._apply(FLOW_INVADJ, g)


class LenseFlow:
    """`LenseFlow(ϕ, n)` / `CachedLenseFlow` (src/lenseflow.jl:19-60): `L(ϕ)` re-caches only when ϕ is a
    different object (src/lenseflow.jl:123-129)."""

    def __init__(self, proj, nsteps=7):
        self.proj, self.nsteps = proj, int(nsteps)
        self.lib = proj.lib
        self._h = ctypes.c_void_p()
        check(self.lib.cmbl_lenseflow_create(proj._h, self.nsteps, ctypes.byref(self._h)))
        self._phi = None

    def __del__(self):
        try:
            if self._h:
                self.lib.cmbl_lenseflow_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def __call__(self, phi):
        """phi: Field (MAP or FOURIER, P=1)."""
        if self._phi is not phi:
            P, B = self.proj._check(phi.arr, phi.basis)
            assert P == 1
            check(self.lib.cmbl_lenseflow_set_phi(self._h, phi.basis, _ptr(phi.arr), B))
            self._phi = phi
        return self

    def invalidate(self):
        self._phi = None

    @property
    def phi(self):                                   # getϕ (src/lenseflow.jl:69-70)
        return self._phi

    def _apply(self, mode, f, basis_out=None):
        P, B = self.proj._check(f.arr, f.basis)
        if basis_out is None:
            basis_out = MAP if mode in (FLOW_FWD, FLOW_INV) else FOURIER      # Ł / Ð of the result (src/flowops.jl:11-14)
        out = self.proj.empty(basis_out, P, B)
        check(self.lib.cmbl_lenseflow_apply(self._h, mode, f.basis, _ptr(f.arr), basis_out, _ptr(out), P, B))
        return Field(self.proj, out, basis_out)

    def __mul__(self, f):            # L * f
        return self._apply(FLOW_FWD, f)

    def ldiv(self, f):               # L \ f
        return self._apply(FLOW_INV, f)

    @property
    def adjoint(self):
        return _Adjoint(self)

    def max_lensing_step(self, phi, eta):
        """get_max_lensing_step (src/lenseflow.jl:242-256), one value per batch slot"""
        eta = eta.to(phi.basis)
        P, B = self.proj._check(phi.arr, phi.basis)
        out = (ctypes.c_double * B)()
        check(self.lib.cmbl_max_lensing_step(self._h, phi.basis, _ptr(phi.arr), _ptr(eta.arr), B, out))
        return np.array(out[:])

    def gradient(self, mode, f_end, delta, alias_quirk=None, basis_df=None):
        """Pullback of `L*f` (mode=FLOW_FWD) or `L\\f` (FLOW_INV) (src/flowops.jl:40-68).  alias_quirk=None: `reference_exact()`.
        f_end: primal OUTPUT (map Field); delta: cotangent.  Returns (δϕ [FOURIER], δf, f_start)."""
        P, B = self.proj._check(f_end.arr, MAP)
        self.proj._check(delta.arr, delta.basis)
        basis_df = delta.basis if basis_df is None else basis_df
        alias_quirk = reference_exact() if alias_quirk is None else alias_quirk
        dphi = self.proj.empty(FOURIER, 1, B)
        df = self.proj.empty(basis_df, P, B)
        fstart = self.proj.empty(MAP, P, B)
        check(self.lib.cmbl_lenseflow_grad(self._h, mode, _ptr(f_end.arr), delta.basis, _ptr(delta.arr), _ptr(dphi),
                                           basis_df, _ptr(df), _ptr(fstart), P, B, 1 if alias_quirk else 0))
        return Field(self.proj, dphi, FOURIER), Field(self.proj, df, basis_df), Field(self.proj, fstart, MAP)


class BaseDataSet:
    """`BaseDataSet` at fiducial θ (src/dataset.jl:37-57) with the operators resident on the device.

    ops: dict name -> real planes (numpy/torch, reference layout):
        'Cf_inv','Cn_inv','B','Mf','D','D_inv','precond_inv' : (P or 5, Nx, Nyh)   harmonic basis
        'Cphi_inv','G_inv' : (1, Nx, Nyh) ;  'Mpix' : (Nx, Ny) optional
    d: harmonic-basis data (B,P,Nx,Nyh); logdet_sum: logdet Cf + logdet Cϕ + logdet Cn.
    """
    _ids = dict(Cf_inv=OP_CF_INV, Cn_inv=OP_CN_INV, B=OP_B, Mf=OP_MF, D=OP_D, D_inv=OP_D_INV,
                precond_inv=OP_PRECOND_INV, Cphi_inv=OP_CPHI_INV, G_inv=OP_G_INV, Mpix=OP_MPIX)

    def __init__(self, proj, P, ops, d=None, logdet_sum=0.0, nsteps=7):
        self.proj, self.P, self.lib = proj, int(P), proj.lib
        self._h = ctypes.c_void_p()
        check(self.lib.cmbl_dataset_create(proj._h, self.P, ctypes.byref(self._h)))
        self.L = LenseFlow(proj, nsteps)
        self.ops = {}
        for k, v in ops.items():
            if v is None:
                continue
            t = proj.tensor(v)
            t = t.reshape(1, *t.shape) if t.dim() == 2 else t
            self.ops[k] = t
            check(self.lib.cmbl_dataset_set_op(self._h, self._ids[k], _ptr(t), t.shape[0]))
        self.d = None
        if d is not None:
            self.set_data(d)
        self.logdet_sum = float(logdet_sum)
        self.logdet_mix = 0.0          # logdet(D,θ) + logdet(G,θ) of the mixed parametrisation (src/dataset.jl:86); 0 at fiducial θ
        # ϕ-gradients: False = the mathematically consistent δϕ velocity (default), True = the reference exactly as written, with
        # the in-place aliasing of src/lenseflow.jl:198-200 (differs by ~3e-4).  Every driver (MAP_joint, MAP_marg,
        # hmc_step, sample_joint) takes `alias_quirk=None` = this dataset-level setting.  CMBL_REFERENCE_EXACT=1 flips the default
        # (and starts every context with plain working-precision sums, the reference's `sum_accuracy_mode`).
        self.alias_quirk = reference_exact()
        check(self.lib.cmbl_dataset_set_logdet(self._h, self.logdet_sum))

    def __del__(self):
        try:
            if self._h:
                self.lib.cmbl_dataset_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_logdet(self, logdet_sum):
        self.logdet_sum = float(logdet_sum)
        check(self.lib.cmbl_dataset_set_logdet(self._h, self.logdet_sum))

    def set_op(self, name, planes):
        t = self.proj.tensor(planes)
        t = t.reshape(1, *t.shape) if t.dim() == 2 else t
        self.ops[name] = t
        if name == "Mpix":
            self._mask_full = (None, None)          # expanded copy of the pixel mask is stale
        self.__dict__.pop("_qe_planes", None)       # device copies of the estimator's planes (drivers.quadratic_estimate_native) are stale too
        check(self.lib.cmbl_dataset_set_op(self._h, self._ids[name], _ptr(t), t.shape[0]))

    def _apply(self, name, f, basis_out=HARMONIC):
        """harmonic-basis operator `name` applied to Field f"""
        return Field(self.proj, self.proj.diag_apply(self.ops[name], f.arr, HARMONIC, f.basis, basis_out), basis_out)

    def _applyT(self, name, f, basis_out=HARMONIC):
        """transpose of the (real) harmonic-basis operator `name`: for BlockDiagIEB planes (TT,TE,ET,EE,BB) swap TE <-> ET"""
        op = self.ops[name]
        if op.shape[0] == 5:
            op = op[[0, 2, 1, 3, 4]].contiguous()
        return Field(self.proj, self.proj.diag_apply(op, f.arr, HARMONIC, f.basis, basis_out), basis_out)

    def _maskT(self, f):
        """M' = Mpix' * Mfourier' (src/dataset.jl:279-285) on a Field; returns HARMONIC"""
        f = self._applyT("Mf", f)
        if "Mpix" not in self.ops:
            return f
        m = f.to(MAP)
        key = tuple(m.arr.shape)
        if getattr(self, "_mask_full", (None, None))[0] != key:
            self._mask_full = (key, self.ops["Mpix"].reshape(1, 1, self.proj.Nx, self.proj.Ny).expand(*key).contiguous())
        return Field(self.proj, self.proj.map_fma(m.arr, self._mask_full[1]), MAP).to(HARMONIC)

    def gradientphi_logpdf(self, f, phi, d=None, alias_quirk=None):
        """∂/∂ϕ logpdf(ds; f, ϕ, d) at fixed f -- what `gradient(ϕ -> logpdf(dsθ; f=f_wf, ϕ, dsθ.d), ϕ)` evaluates in MAP_marg
        (src/maximization.jl:301): the δ-flow pullback of L(ϕ)*f (src/flowops.jl:40-54) applied to ∂/∂f̃ = B'M'Cn⁻¹(d − MBLf),
        minus Cϕ⁻¹ϕ.  f, d may carry B batch slots against one ϕ."""
        d = self.d if d is None else d
        alias_quirk = self.alias_quirk if alias_quirk is None else alias_quirk
        ft = self.L(phi) * f.to(MAP)
        z = d.to(HARMONIC) - self._mask(self._apply("B", ft))
        w = self._applyT("B", self._maskT(self._apply("Cn_inv", z)), basis_out=FOURIER)
        dphi, _, _ = self.L(phi).gradient(FLOW_FWD, ft, w, alias_quirk=alias_quirk)
        prior = self.proj.diag_apply(self.ops["Cphi_inv"], phi.to(FOURIER).arr, FOURIER, FOURIER)
        if prior.shape[0] != dphi.arr.shape[0]:
            prior = prior.expand(dphi.arr.shape[0], -1, -1, -1).contiguous()
        return dphi - Field(self.proj, prior, FOURIER)

    def _mask(self, f):
        """M = Mfourier * Mpix (src/dataset.jl:279-285) on a Field; returns HARMONIC"""
        if "Mpix" in self.ops:
            m = f.to(MAP)
            key = tuple(m.arr.shape)
            if getattr(self, "_mask_full", (None, None))[0] != key:
                self._mask_full = (key, self.ops["Mpix"].reshape(1, 1, self.proj.Nx, self.proj.Ny).expand(*key).contiguous())
            f = Field(self.proj, self.proj.map_fma(m.arr, self._mask_full[1]), MAP)
        return self._apply("Mf", f)

    def mean(self, f, phi):
        """μ = M·B·L(ϕ)·f (src/dataset.jl:59-66)"""
        ft = self.L(phi) * f.to(MAP)
        return self._mask(self._apply("B", ft))

    def logpdf(self, f, phi, d=None):
        """logpdf(ds; f, ϕ) (src/dataset.jl:59-66, src/distributions.jl:11-15), per batch slot"""
        f, phi = f.to(HARMONIC), phi.to(FOURIER)
        d = self.d if d is None else d
        z = self.mean(f, phi) - d
        q1 = f.dot(self._apply("Cf_inv", f))
        q3 = z.dot(self._apply("Cn_inv", z))
        cp = Field(self.proj, self.proj.diag_apply(self.ops["Cphi_inv"], phi.arr, FOURIER, FOURIER), FOURIER)
        q2 = phi.dot(cp)
        return -0.5 * (q1 + q2 + q3 + self.logdet_sum)

    def set_data(self, d):
        d = d if isinstance(d, Field) else Field(self.proj, self.proj.tensor(d), HARMONIC)
        d = d.to(HARMONIC)
        self.d = d
        check(self.lib.cmbl_dataset_set_data(self._h, _ptr(d.arr), d.arr.shape[0]))

    def gradientf_logpdf(self, f, phi, d=None, zero_d=False):
        """src/dataset.jl:76-80"""
        f = f.to(HARMONIC)
        L = self.L(phi)
        B = f.arr.shape[0]
        out = self.proj.empty(HARMONIC, self.P, B)
        dd = None if d is None else d.to(HARMONIC).arr
        check(self.lib.cmbl_gradientf_logpdf(self._h, L._h, _ptr(f.arr), _ptr(dd), 1 if zero_d else 0, _ptr(out), B))
        return Field(self.proj, out, HARMONIC)

    def argmaxf_logpdf(self, phi, d=None, fstart=None, tol=1e-1, nsteps=500):
        """Wiener filter (src/maximization.jl:17-42): returns (f [HARMONIC], history [(i, res_per_batch)])."""
        L = self.L(phi)
        dd = self.d if d is None else d.to(HARMONIC)
        B = dd.arr.shape[0]
        out = self.proj.empty(HARMONIC, self.P, B)
        hist = (ctypes.c_double * (nsteps * B))()
        nit = ctypes.c_int(0)
        fs = None if fstart is None else fstart.to(HARMONIC).arr
        check(self.lib.cmbl_wiener_cg(self._h, L._h, _ptr(dd.arr), _ptr(fs), float(tol), int(nsteps), _ptr(out),
                                      hist, ctypes.byref(nit), B))
        h = np.array(hist[: nit.value * B]).reshape(nit.value, B)
        return Field(self.proj, out, HARMONIC), [(i + 1, h[i]) for i in range(nit.value)]

    def logpdf_mixed(self, fo, phio):
        """logpdf(Mixed(ds); f°, ϕ°) (src/dataset.jl:84-87)"""
        fo, phio = fo.to(MAP), phio.to(FOURIER)
        B = fo.arr.shape[0]
        lp = (ctypes.c_double * B)()
        self.L.invalidate()
        check(self.lib.cmbl_logpdf_mixed(self._h, self.L._h, _ptr(fo.arr), _ptr(phio.arr), lp, B))
        return np.array(lp[:]) - self.logdet_mix

    def gradient_logpdf_mixed(self, fo, phio, alias_quirk=None):
        """(logpdf, ∇f° [MAP], ∇ϕ° [FOURIER]) — the "∇lnP" step (test/runbenchmarks.jl:120).  A NaN logpdf is returned as NaN."""
        alias_quirk = self.alias_quirk if alias_quirk is None else alias_quirk
        fo, phio = fo.to(MAP), phio.to(FOURIER)
        B = fo.arr.shape[0]
        lp = (ctypes.c_double * B)()
        gfo = self.proj.empty(MAP, self.P, B)
        gpo = self.proj.empty(FOURIER, 1, B)
        self.L.invalidate()
        check(self.lib.cmbl_grad_logpdf_mixed(self._h, self.L._h, _ptr(fo.arr), _ptr(phio.arr), lp, _ptr(gfo), _ptr(gpo),
                                              B, 1 if alias_quirk else 0))
        return np.array(lp[:]) - self.logdet_mix, Field(self.proj, gfo, MAP), Field(self.proj, gpo, FOURIER)

    def mix(self, f, phi, G=None):
        """f° = L(ϕ)·D·f, ϕ° = G·ϕ (src/dataset.jl:96-101)"""
        fo = self.L(phi) * self._apply("D", f.to(HARMONIC))
        if G is None:                                  # G = pinv(G⁻¹): ϕ° = nan2zero(ϕ / G⁻¹), one library call (DiagOp `\`)
            phio = self.proj.diag_apply(self.ops["G_inv"], phi.to(FOURIER).arr, FOURIER, FOURIER, kind=DIAG_DIV_NAN2ZERO)
        else:
            phio = self.proj.diag_apply(np.asarray(G)[None], phi.to(FOURIER).arr, FOURIER, FOURIER)
        return fo, Field(self.proj, phio, FOURIER)

    def unmix(self, fo, phio, G=None):
        """ϕ = G \\ ϕ°, f = D \\ (L(ϕ) \\ f°) (src/dataset.jl:111-117)"""
        Gi = self.ops["G_inv"] if G is None else (1.0 / np.asarray(G))[None]
        phi = Field(self.proj, self.proj.diag_apply(Gi, phio.to(FOURIER).arr, FOURIER, FOURIER), FOURIER)
        fhat = self.L(phi).ldiv(fo.to(MAP))
        return self._apply("D_inv", fhat), phi
