"""MI355X-native flat-sky CMB lensing field engine (host-side mirror of the CMBLensing.jl operator surface).

The compute path is the hand-written HIP library `libcmblens_hip.so` (csrc/, C ABI in include/cmblens.h).
This package is plumbing: it loads the library with ctypes, holds device memory in torch tensors, and mirrors
the reference's names (ProjLambert, LenseFlow, BaseDataSet, argmaxf_logpdf, ...) for the path in scope.
There is NO CPU fallback: without the built library or without a GPU every compute call raises.
"""
from .lib import load_library, library_path, CmblError, build            # noqa: F401
from .engine import (ProjLambert, LenseFlow, BaseDataSet, Field, MAP, FOURIER, HARMONIC,   # noqa: F401
                     FLOW_FWD, FLOW_INV, FLOW_ADJ, FLOW_INVADJ, reference_exact)
from .sim import (Cls, load_sim, noise_cls, beam_cls, lowpass, cl_to_2d, HarmOp, border_mask)   # noqa: F401
from .chains import partition_chains, chain_seed, gather_chain_values, allreduce_sum   # noqa: F401
from .drivers import (quadratic_estimate, MAP_joint, MAP_joint_step, hmc_step, sample_f, gibbs_step, symplectic_integrate,   # noqa: F401
                      mass_matrix_phi, brent_minimize, sample_joint, MAP_marg, simulate_data, hmc_step_native, MAP_joint_step_native, quadratic_estimate_native)
from . import rng                                                         # noqa: F401
from .chainfile import load_chains, Chain, Chains                         # noqa: F401
from .theta import (set_theta, logpdf_mixed_theta, grid_and_sample, gibbs_sample_theta, findbin, bandpower_rescale,   # noqa: F401
                    BinRescaledCov, use_bandpowers)
from .muse import CMBLensingMuseProblem                                   # noqa: F401
