"""Chain files: what `sample_joint` writes every `nfilewrite` steps and `load_chains` reads back.

Mirrors the reference's on-disk structure (src/sampling.jl:230-256,311-320; src/chains.jl:48-100): one file holding `rundat`
(the run's settings) and numbered chunks `chunks_1, chunks_2, ...`; a chunk is, per chain, the list of samples since the
previous write; a sample is a dict of scalars (every step) plus maps (first step, every `nsavemaps`-th step, and -- so that a
resumed run continues exactly -- the last step of every chunk).  Two containers, chosen by the file name's extension:

  `.jld2`  EXPERIMENTAL output in the reference's container (JLD2, an HDF5 dialect), written by jld2_writer.py: `rundat` and `chunks_k` =
           Vector{Vector{Any}} of `Dict{Symbol,Any}` samples under the reference's state keys (ϕ, f, logpdf, ΔH, accept, step, θ;
           src/sampling.jl:290,396-402,446) with the reference's step numbering (initial state = step 1), appended like
           `jldopen(filename, "a+")`.  Stated limits: maps are stored as plain arrays, not as `BaseField` structs, θ as a `Dict` rather than a
           `NamedTuple`, the initial state is not saved as a first sample, and NO file written here has ever been opened by JLD2.jl (no Julia
           in the build image) -- the layout follows the reference's own data file structure by structure and round-trips through the
           reader here, which cannot prove that Julia's `load_chains` accepts it.  `.zip` is the documented container of this package.
  `.zip`   a plain zip archive whose members are `.npy` arrays, `chunks_<k>/chain<c>/<i>/<key>.npy`, appended in place.

Reading: `load_chains`, `read_rundat`, `chunk_indices`, `read_chunk` and `last_state` open both, including `.jld2` chains written by
the Julia package itself (decoded by jld2.py).  Field values (structs with `arr` and `metadata`) come back as their arrays in this
package's axis order, NamedTuple θ as `theta_<name>` scalars, and the reference's keys are renamed to the ones used here (ϕ -> phi,
ΔH -> dH, ...; unknown keys keep their Julia names).  A chain file the Julia package wrote is READ-ONLY here: analyse it with
`load_chains`, or continue it in a NEW file with `sample_joint(filename=new, resume=path_of_the_jld2)` -- `last_state` then converts
what differs between the two: a Map-basis ϕ goes through rfft2, and the reference's step counter, which stores the initial state as
step 1 (src/sampling.jl:268,277), is shifted to this package's (first Gibbs pass = step 1).
"""
import io
import json
import os
import zipfile

import numpy as np

EXT = ".zip"
JLD2_EXT = ".jld2"
EXTS = (EXT, JLD2_EXT)
# reference state keys (src/sampling.jl:388-464; `i` / `lnP`: legacy aliases, read side only) -> the names this package's chain files use
JLD2_KEYS = {"ϕ": "phi", "ϕ°": "phi_mixed", "f°": "f_mixed", "f̃": "ftilde", "ΔH": "dH", "lnP": "logpdf", "i": "step"}


def _is_jld2(filename):
    return os.path.splitext(filename)[1] == JLD2_EXT


_jld2_open = {}


def _jld2(filename):
    """one decoded file per (path, mtime, size): a chain file is read several times by load_chains / last_state"""
    from .jld2 import JLD2File
    st = os.stat(filename)
    key = (os.path.abspath(filename), st.st_mtime_ns, st.st_size)
    if key not in _jld2_open:
        _jld2_open.clear()
        _jld2_open[key] = JLD2File(filename)
    return _jld2_open[key]


def _jld2_value(v):
    """a decoded Julia value as a chain-sample entry: Field struct -> its array, scalars -> Python numbers.  A Map-basis field
    (`BaseField{Map,...}`: real array) stays real, a Fourier-basis one complex: `last_state` tells them apart by dtype."""
    if isinstance(v, dict):
        body = {k: x for k, x in v.items() if k != "__julia_type__"}
        if "arr" in body and "metadata" in body:                          # BaseField{B,M,T,A} (src/base_fields.jl:14-21)
            return np.asarray(body["arr"])
        return {k: _jld2_value(x) for k, x in body.items()}
    if isinstance(v, list):
        return [_jld2_value(x) for x in v]
    return v


def _jld2_sample(state, own_legacy=False):
    """Dict{Symbol,Any} state of the reference -> sample dict with this package's key names; a NamedTuple θ is flattened.
    own_legacy: the file was written by THIS package (header stamp, `written_by_julia`) -- only then can the legacy key `i` be its round-4
    writer's, which stored package numbering; in a file of the Julia package `i` carries the reference's numbering like `step`."""
    out = {}
    for k, v in state.items():
        v = _jld2_value(v)
        if k == "θ" and isinstance(v, dict):
            out.update({"theta_" + kk: vv for kk, vv in v.items()})
        elif k == "i" and own_legacy:
            # this package's first `.jld2` writer (round 4): `i` already in PACKAGE numbering (first Gibbs pass = step 1).  Re-expressed the way
            # the reference numbers `step` (initial state = step 1) so that every reader below can shift uniformly (such a file
            # used to be resumed one step early)
            out["step"] = int(v) + 1
        else:
            out[JLD2_KEYS.get(k, k)] = v
    return out


def _put(z, name, arr):
    buf = io.BytesIO()
    np.save(buf, np.asarray(arr), allow_pickle=False)
    z.writestr(name, buf.getvalue())


def _get(z, name):
    return np.load(io.BytesIO(z.read(name)), allow_pickle=False)


def written_by_julia(filename):
    """a `.jld2` file that the Julia package wrote (as opposed to jld2_writer.py)"""
    if not _is_jld2(filename):
        return False
    with open(filename, "rb") as fh:
        return b" (cmblensing.jl_amd" not in fh.read(128)


def check_filename(filename, resume):
    """argument validation of src/sampling.jl:236-241 (the reference accepts `.jld2` only; here also the `.zip` container)"""
    if filename is None:
        return
    if os.path.splitext(filename)[1] not in EXTS:
        raise ValueError(f"Chain filename '{filename}' should have '{JLD2_EXT}' or '{EXT}' extension.")
    if os.path.isfile(filename) and resume is None:
        raise ValueError(f"'{filename}' exists so must specify `resume=True` or `resume=False`.")
    if os.path.isfile(filename) and resume is True and written_by_julia(filename):
        raise ValueError(f"'{filename}' was written by the Julia package and is read-only here: continue it in a new file with "
                         f"`filename=<new file>, resume='{filename}'`.")


# this package's sample keys -> the state keys of the reference (src/sampling.jl:396-402 `@pack! state = ϕ°, Ω, ΔH, accept`, :446
# `@pack! state = f̃, logpdf`, :290 `setindex!.(states, step, :step)`).  `step` and `logpdf` ARE the reference's names and stay; the
# legacy aliases `i` / `lnP` of JLD2_KEYS are understood on the read side only.
_TO_JULIA = {"phi": "ϕ", "phi_mixed": "ϕ°", "f_mixed": "f°", "ftilde": "f̃", "dH": "ΔH"}


def _julia_sample(samp):
    """sample dict with this package's key names -> the reference's (`theta_<name>` scalars fold into one θ entry).  The step counter
    is written in the reference's numbering -- its initial state is step 1 and the first Gibbs pass step 2 (src/sampling.jl:263,288-290),
    here the first pass is step 1 -- so that `@unpack step = states[1]` (:256) of a resuming Julia session continues at the right pass."""
    out, theta = {}, {}
    for k, v in samp.items():
        if k.startswith("theta_"):
            theta[k[6:]] = float(v)
        elif k == "step":
            out["step"] = int(v) + 1
        else:
            out[_TO_JULIA.get(k, k)] = v
    if theta:
        out["θ"] = theta
    return out


def write_chunk(filename, index, chains, rundat=None, clobber=False):
    """chains: list over chains of lists of sample dicts {key: scalar | ndarray}.  `clobber` starts a new file ("w" vs "a+")."""
    if _is_jld2(filename):
        from .jld2_writer import JLD2Writer
        _jld2_open.clear()
        with JLD2Writer(filename, "w" if clobber or not os.path.exists(filename) else "a") as w:
            if "rundat" not in w:                                         # haskey(io, "rundat") || write(io, "rundat", ...)  (:313)
                w.write("rundat", dict(rundat or {}))
            w.write(f"chunks_{index}", [[_julia_sample(s) for s in ch] for ch in chains])
        return
    with zipfile.ZipFile(filename, "w" if clobber else "a", compression=zipfile.ZIP_STORED) as z:
        if "rundat.json" not in z.namelist():
            z.writestr("rundat.json", json.dumps(rundat or {}, default=lambda o: np.asarray(o).tolist()))
        for c, chain in enumerate(chains):
            for i, samp in enumerate(chain):
                for k, v in samp.items():
                    _put(z, f"chunks_{index}/chain{c}/{i}/{k}.npy", v)


def chunk_indices(filename):
    if _is_jld2(filename):
        return sorted(int(k[7:]) for k in _jld2(filename).keys() if k.startswith("chunks_"))      # src/chains.jl:63-64
    with zipfile.ZipFile(filename, "r") as z:
        return sorted({int(n.split("/")[0][7:]) for n in z.namelist() if n.startswith("chunks_")})


def read_rundat(filename):
    if _is_jld2(filename):
        from .jld2 import to_python
        return {k: _jld2_value(v) for k, v in to_python(_jld2(filename)["rundat"]).items()}
    with zipfile.ZipFile(filename, "r") as z:
        return json.loads(z.read("rundat.json"))


def read_chunk(filename, index, dropmaps=False):
    """-> list over chains of lists of sample dicts"""
    if _is_jld2(filename):
        from .jld2 import to_python
        chains = to_python(_jld2(filename)[f"chunks_{index}"])            # Vector (chains) of Vector{Any} (samples) of Dict{Symbol,Any}
        own = not written_by_julia(filename)
        samples = [[_jld2_sample(s, own) for s in ch] for ch in chains]
        if dropmaps:
            samples = [[{k: v for k, v in s.items() if np.ndim(v) == 0 and not isinstance(v, (dict, list))} for s in ch] for ch in samples]
        return samples
    out = {}
    with zipfile.ZipFile(filename, "r") as z:
        pre = f"chunks_{index}/"
        for n in z.namelist():
            if not n.startswith(pre):
                continue
            _, ch, i, key = n.split("/")
            v = _get(z, n)
            if dropmaps and v.ndim > 0:
                continue
            out.setdefault(int(ch[5:]), {}).setdefault(int(i), {})[key[:-4]] = v[()] if v.ndim == 0 else v
    return [[out[c][i] for i in sorted(out[c])] for c in sorted(out)]


class Chain(list):
    """one chain = list of sample dicts.  `chain["key"]` stacks that key over samples (None where a sample lacks it,
    src/chains.jl:131-146); `chain[10:, "key"]` slices first."""

    def __getitem__(self, k):
        if isinstance(k, str):
            vals = [s.get(k) for s in self]
            return np.array(vals) if all(v is not None and np.ndim(v) == 0 for v in vals) else vals
        if isinstance(k, tuple):
            sel = self[k[0]]
            return sel if len(k) == 1 else sel[k[1:] if len(k) > 2 else k[1]]
        r = list.__getitem__(self, k)
        return Chain(r) if isinstance(k, slice) else r


class Chains(list):
    """parallel chains.  `chains["key"]` -> (nchains, nsamples) array for scalar keys (leading colons dropped,
    src/chains.jl:103-111); `chains[c]` a Chain; `chains[c, 5:, "key"]`, `chains[:, -1, "phi"]` index chain, sample, key."""

    def __getitem__(self, k):
        if isinstance(k, str):
            per = [c[k] for c in self]
            return np.array(per) if all(isinstance(p, np.ndarray) for p in per) and len({len(p) for p in per}) == 1 else per
        if isinstance(k, tuple):
            sel, rest = self[k[0]], (k[1:] if len(k) > 2 else k[1])
            if isinstance(sel, Chains):
                per = [c[rest] for c in sel]
                same = all(isinstance(p, np.ndarray) or np.ndim(p) == 0 for p in per) and len({np.shape(p) for p in per}) == 1
                return np.array(per) if same and not isinstance(per[0], dict) else per
            return sel[rest]
        r = list.__getitem__(self, k)
        return Chains(r) if isinstance(k, slice) else r


def load_chains(filename, burnin=0, thin=1, join=False, dropmaps=False, burnin_chunks=0):
    """`load_chains` (src/chains.jl:48-100).  burnin < 0 keeps only that many samples at the end; `thin` is an int, "hasmaps"
    (only samples that carry ϕ) or a predicate on the sample dict; `join` concatenates the chains into one `Chain`."""
    ks = chunk_indices(filename)
    ks = ks[burnin_chunks:]                       # negative: keep only that many chunks at the end
    chains = None
    for k in ks:
        part = read_chunk(filename, k, dropmaps)
        if chains is None:
            chains = part
        else:
            for c, p in zip(chains, part):
                c.extend(p)
    chains = chains or [[]]
    if isinstance(thin, int) and not isinstance(thin, bool):
        chains = [c[burnin::thin] if burnin >= 0 else c[len(c) + burnin::thin] for c in chains]
    elif thin == "hasmaps":
        chains = [[s for s in c[burnin:] if "phi" in s] for c in chains]
    elif callable(thin):
        chains = [[s for s in c if thin(s)] for c in chains]
    else:
        raise ValueError("`thin` should be an int, 'hasmaps', or a filter function")
    chains = Chains([Chain(c) for c in chains])
    return Chain([s for c in chains for s in c]) if join else chains


def last_state(filename):
    """(step, per-chain last sample that carries maps) of the newest chunk -- what `resume=True` restarts from
    (src/sampling.jl:247-256)"""
    ks = chunk_indices(filename)
    if not ks:
        raise ValueError(f"Can't resume chain which contains no samples: {filename}")
    last = [dict(c[-1]) for c in read_chunk(filename, ks[-1])]
    if any("phi" not in s for s in last):
        raise ValueError(f"last sample of {filename} carries no maps")
    step = int(last[0]["step"])
    if _is_jld2(filename):
        # a `.jld2` chain file -- the Julia package's or jld2_writer.py's -- holds the reference's conventions: a field may be saved in
        # the Map basis (a real array: transform it; NumPy (.., Nx, Ny) == Julia (Ny, Nx, ..) so the half-plane is the last axis), and the
        # step counter has the initial state as step 1 (src/sampling.jl:263,288-290)
        for s in last:
            for k in ("phi", "f"):
                if k in s and not np.iscomplexobj(s[k]):
                    s[k] = np.fft.rfft2(np.asarray(s[k], float), axes=(-2, -1))
            s["step"] = int(s["step"]) - 1
        step -= 1
    return ks[-1] + 1, step, last
