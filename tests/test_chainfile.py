"""Chain-file container (host side, no GPU): chunked append, load_chains options of src/chains.jl:48-100, resume bookkeeping."""
import importlib.util
import os

import numpy as np
import pytest


def _cf():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cmbl_chainfile", os.path.join(here, "cmblensing.jl_amd", "chainfile.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _samples(chain, steps, maps_every=2):
    out = []
    for s in steps:
        d = dict(step=s, logpdf=-100.0 * chain - s, accept=float(s % 2))
        if s % maps_every == 0:
            d["phi"] = np.full((4, 3), complex(chain, s))
            d["f"] = np.full((2, 4, 3), complex(s, chain))
        out.append(d)
    return out


def test_write_append_load(tmp_path):
    CF = _cf()
    fn = str(tmp_path / "chain.zip")
    with pytest.raises(ValueError):
        CF.check_filename(str(tmp_path / "chain.h5"), None)            # `.jld2` (the reference's, src/sampling.jl:236-238) or `.zip`
    CF.check_filename(str(tmp_path / "chain.jld2"), None)
    CF.check_filename(fn, None)
    CF.write_chunk(fn, 1, [_samples(c, range(1, 5)) for c in range(3)], rundat=dict(nchains=3, eps=0.01), clobber=True)
    with pytest.raises(ValueError):
        CF.check_filename(fn, None)                                    # exists: resume must be explicit
    CF.write_chunk(fn, 2, [_samples(c, range(5, 9)) for c in range(3)])
    assert CF.chunk_indices(fn) == [1, 2] and CF.read_rundat(fn)["nchains"] == 3
    ch = CF.load_chains(fn)
    assert len(ch) == 3 and len(ch[0]) == 8
    np.testing.assert_array_equal(ch["step"], np.tile(np.arange(1, 9), (3, 1)))
    np.testing.assert_array_equal(ch[1, "logpdf"], -100.0 - np.arange(1, 9))
    assert ch[0, 0].get("phi") is None and ch[2, 1]["phi"][0, 0] == complex(2, 2)
    assert ch[:, -1, "step"].tolist() == [8, 8, 8]
    # burnin / thin / hasmaps / predicate / join / dropmaps / burnin_chunks
    assert CF.load_chains(fn, burnin=2, thin=3)["step"].tolist() == [[3, 6]] * 3
    assert CF.load_chains(fn, burnin=-3)["step"].tolist() == [[6, 7, 8]] * 3
    assert CF.load_chains(fn, thin="hasmaps")["step"].tolist() == [[2, 4, 6, 8]] * 3
    assert CF.load_chains(fn, thin=lambda s: s["accept"] == 1)["step"].tolist() == [[1, 3, 5, 7]] * 3
    assert len(CF.load_chains(fn, join=True)) == 24
    assert all("phi" not in s for s in CF.load_chains(fn, dropmaps=True)[0])
    assert CF.load_chains(fn, burnin_chunks=1)["step"].tolist() == [[5, 6, 7, 8]] * 3
    assert CF.load_chains(fn, burnin_chunks=-1)["step"].tolist() == [[5, 6, 7, 8]] * 3
    # resume bookkeeping: next chunk index, last step, last samples with maps
    k, step, last = CF.last_state(fn)
    assert (k, step) == (3, 8) and last[2]["phi"][0, 0] == complex(2, 8)
    CF.write_chunk(fn, 3, [_samples(c, [9]) for c in range(3)])       # last sample without maps -> cannot resume from it
    with pytest.raises(ValueError):
        CF.last_state(fn)


def test_container_is_readable_without_the_package(tmp_path):
    """Independent reader: the chain container is a plain zip of .npy members laid out like the reference's JLD2 groups
    (`rundat`, `chunks_<k>` -> per chain -> per sample -> key; src/sampling.jl:311-320), so it can be read with nothing but
    zipfile + numpy -- here without touching cmblensing.jl_amd.chainfile -- and agrees with load_chains."""
    import io, json, zipfile
    import importlib.util
    spec = importlib.util.spec_from_file_location("_cf", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cmblensing.jl_amd", "chainfile.py"))
    CF = importlib.util.module_from_spec(spec); spec.loader.exec_module(CF)
    fn = str(tmp_path / "c.zip")
    rng = np.random.default_rng(0)
    chains = [[dict(step=i + 1, logpdf=float(rng.normal()), theta_Aphi=1.0 + 0.1 * i, **({"phi": rng.normal(size=(4, 3)) + 1j} if i % 2 == 0 else {}))
               for i in range(3)] for _ in range(2)]
    CF.write_chunk(fn, 1, chains, rundat=dict(nchains=2, eps=0.01), clobber=True)
    CF.write_chunk(fn, 2, [[dict(step=4, logpdf=0.5, theta_Aphi=1.3, phi=np.ones((4, 3), complex))]] * 2)
    # --- the independent reader
    got = {}
    with zipfile.ZipFile(fn) as z:
        assert json.loads(z.read("rundat.json"))["nchains"] == 2
        for name in z.namelist():
            if name.startswith("chunks_"):
                chunk, chain, idx, key = name.split("/")
                got.setdefault(int(chain[5:]), {}).setdefault((int(chunk[7:]), int(idx)), {})[key[:-4]] = np.load(io.BytesIO(z.read(name)))
    ch = CF.load_chains(fn)
    for c in (0, 1):
        samples = [got[c][k] for k in sorted(got[c])]
        assert [int(s["step"]) for s in samples] == [1, 2, 3, 4] == ch[c]["step"].tolist()
        np.testing.assert_allclose([float(s["theta_Aphi"]) for s in samples], ch[c]["theta_Aphi"])
        np.testing.assert_allclose(samples[0]["phi"], chains[c][0]["phi"])
        assert "phi" not in samples[1]
