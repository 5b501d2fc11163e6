"""MUSE adapter: the problem interface MuseInference.jl drives (ext/CMBLensingMuseInferenceExt.jl:19-91), on top of the device dataset.

MuseInference itself (the score-matching outer loop) is a third-party package and is not part of the reference repository; what
the reference ships is the adapter -- `logLike`, `∇θ_logLike`, `sample_x_z`, `ẑ_at_θ` for a `DataSet` -- and that is what this module
mirrors, method for method:

    logLike(d, z, θ)        = logpdf(ds; z..., θ, d)                         (:44-46)
    ∇θ_logLike(d, z, θ)     = gradient of the above w.r.t. θ                   (:51-53; ForwardDiff over θ in the reference -- third-party
                              AD, unpinned; here central differences of the device logpdf, step `fd_step` relative)
    sample_x_z(rng, θ)      = simulate(rng, ds_for_sims; θ) -> (x = d, z = (f, ϕ)) (:55-65)
    ẑ_at_θ(d, zguess, θ)    = MAP_joint(θ, ds with d, ϕ start; fstart)          (:67-72)
θ is a dict over the dataset's parameters (r, Aϕ, bandpower amplitude vectors: theta.py); `θ_fixed` entries are merged in (:30).
"""
import numpy as np

from .engine import Field, FOURIER, HARMONIC
from . import theta as TH


class CMBLensingMuseProblem:
    def __init__(self, ds, theta_fixed=None, MAP_joint_kwargs=None, base_seed=0, fd_step=1e-3):
        self.ds, self.theta_fixed = ds, dict(theta_fixed or {})
        self.MAP_joint_kwargs = {"nsteps": 5, **(MAP_joint_kwargs or {})}
        self.base_seed, self.fd_step = int(base_seed), float(fd_step)
        self.x = ds.d

    def merge(self, theta):
        return {**self.theta_fixed, **dict(theta)}

    def standardize_theta(self, theta):                      # standardizeθ (:32-35): a flat float vector with names
        if not isinstance(theta, dict):
            raise TypeError("θ should be a dict of parameter name -> value(s)")
        return {k: np.asarray(v, float) for k, v in theta.items()}

    # -- the four interface functions
    def logLike(self, d, z, theta):
        """logpdf(ds; f, ϕ, θ, d), one value per batch slot"""
        TH.set_theta(self.ds, **{k: (float(v) if np.ndim(v) == 0 else v) for k, v in self.merge(theta).items()})
        return self.ds.logpdf(z["f"], z["phi"], d=d)

    def grad_theta_logLike(self, d, z, theta):
        """∂ logLike / ∂θ, same structure as θ; central differences with a relative step"""
        th = self.standardize_theta(theta)
        out = {}
        for k, v in th.items():
            g = np.zeros(v.shape)
            for idx in np.ndindex(*v.shape) if v.ndim else [()]:
                h = self.fd_step * max(abs(float(v[idx])), 1e-3)
                up, dn = {**th, k: v.copy()}, {**th, k: v.copy()}
                up[k][idx] += h
                dn[k][idx] -= h
                g[idx] = float(np.sum(self.logLike(d, z, up) - self.logLike(d, z, dn))) / (2 * h)
            out[k] = g
        TH.set_theta(self.ds, **{k: (float(v) if np.ndim(v) == 0 else v) for k, v in self.merge(th).items()})
        return out

    def sample_x_z(self, seed, theta):
        """simulate(rng, ds; θ): device RNG keyed by `seed` (draw kinds f, ϕ, n); returns (x = d [HARMONIC], z = dict(f, phi))"""
        from . import rng as R
        ds, proj, h = self.ds, self.ds.proj, self.ds.host
        TH.set_theta(ds, **{k: (float(v) if np.ndim(v) == 0 else v) for k, v in self.merge(theta).items()})
        draw = lambda kind, P: proj.randn([self.base_seed + int(seed)], R.stream_id(kind, 0), P)
        col = lambda w, planes, basis: Field(proj, proj.diag_apply(planes, proj.rfft(w), basis, basis), basis)
        f = col(draw(R.STREAM_F, ds.P), h["Cf"].sqrt().p, HARMONIC)
        phi = col(draw(R.STREAM_P, 1), np.sqrt(np.asarray(h["Cphi"], float))[None], FOURIER)
        n = col(draw(R.STREAM_N, ds.P), h["Cn"].sqrt().p, HARMONIC)
        return ds.mean(f, phi) + n, dict(f=f, phi=phi)

    def zhat_at_theta(self, d, zguess, theta):
        """ẑ_at_θ: MAP_joint at θ with the data replaced by d, started from zguess; returns (dict(f, phi), history)"""
        from .drivers import MAP_joint
        ds = self.ds
        TH.set_theta(ds, **{k: (float(v) if np.ndim(v) == 0 else v) for k, v in self.merge(theta).items()})
        keep = ds.d
        ds.set_data(d)
        try:
            f, phi, hist = MAP_joint(ds, phi_start=(zguess or {}).get("phi"), fstart=(zguess or {}).get("f"), **self.MAP_joint_kwargs)
        finally:
            ds.set_data(keep)
        return dict(f=f, phi=phi), hist
