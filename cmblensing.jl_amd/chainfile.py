"""Chain files: what `sample_joint` writes every `nfilewrite` steps and `load_chains` reads back.

Mirrors the reference's on-disk structure (src/sampling.jl:230-256,311-320; src/chains.jl:48-100): one file holding `rundat`
(the run's settings) and numbered chunks `chunks_1, chunks_2, ...`; a chunk is, per chain, the list of samples since the
previous write; a sample is a dict of scalars (every step) plus maps (first step, every `nsavemaps`-th step, and -- so that a
resumed run continues exactly -- the last step of every chunk).  The reference stores this in JLD2 (an HDF5 dialect that needs
Julia to read back); here the container is a plain zip archive whose members are `.npy` arrays,
`chunks_<k>/chain<c>/<i>/<key>.npy`, appended in place chunk by chunk.
"""
import io
import json
import os
import zipfile

import numpy as np

EXT = ".zip"


def _put(z, name, arr):
    buf = io.BytesIO()
    np.save(buf, np.asarray(arr), allow_pickle=False)
    z.writestr(name, buf.getvalue())


def _get(z, name):
    return np.load(io.BytesIO(z.read(name)), allow_pickle=False)


def check_filename(filename, resume):
    """argument validation of src/sampling.jl:236-241"""
    if filename is None:
        return
    if os.path.splitext(filename)[1] != EXT:
        raise ValueError(f"Chain filename '{filename}' should have '{EXT}' extension.")
    if os.path.isfile(filename) and resume is None:
        raise ValueError(f"'{filename}' exists so must specify `resume=True` or `resume=False`.")


def write_chunk(filename, index, chains, rundat=None, clobber=False):
    """chains: list over chains of lists of sample dicts {key: scalar | ndarray}.  `clobber` starts a new file ("w" vs "a+")."""
    with zipfile.ZipFile(filename, "w" if clobber else "a", compression=zipfile.ZIP_STORED) as z:
        if "rundat.json" not in z.namelist():
            z.writestr("rundat.json", json.dumps(rundat or {}, default=lambda o: np.asarray(o).tolist()))
        for c, chain in enumerate(chains):
            for i, samp in enumerate(chain):
                for k, v in samp.items():
                    _put(z, f"chunks_{index}/chain{c}/{i}/{k}.npy", v)


def chunk_indices(filename):
    with zipfile.ZipFile(filename, "r") as z:
        return sorted({int(n.split("/")[0][7:]) for n in z.namelist() if n.startswith("chunks_")})


def read_rundat(filename):
    with zipfile.ZipFile(filename, "r") as z:
        return json.loads(z.read("rundat.json"))


def read_chunk(filename, index, dropmaps=False):
    """-> list over chains of lists of sample dicts"""
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
    last = [c[-1] for c in read_chunk(filename, ks[-1])]
    if any("phi" not in s for s in last):
        raise ValueError(f"last sample of {filename} carries no maps")
    return ks[-1] + 1, int(last[0]["step"]), last
