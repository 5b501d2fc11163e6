"""The JLD2 reader (cmblensing.jl_amd/jld2.py) and `load_chains` on the reference's container format.
  * real-world vector: the reference's own dat/default_camb_Cls.jld2 (written by the JLD2 package: version-2 object headers,
    committed compound datatypes with compound `julia_type` attributes, object references, deflate-compressed arrays) decodes to
    the spectra of tests/golden/camb_cls.npz -- only where /root/reference exists (the build container), skipped elsewhere;
  * a hand-built chain file with the layout `sample_joint` writes (src/sampling.jl:311-320), made by tests/_jld2_writer.py from the
    HDF5 specification: `load_chains` / resume read it like the package's own zip container."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib.util                                     # noqa: E402

_PKG = os.path.join(ROOT, "cmblensing.jl_amd")


def _mod(name):
    """jld2.py / chainfile.py without importing the package (which loads the HIP library): they are pure Python"""
    import types
    pkg = sys.modules.setdefault("_cmbl_io", types.ModuleType("_cmbl_io"))
    pkg.__path__ = [_PKG]
    full = "_cmbl_io." + name
    if full not in sys.modules:
        spec = importlib.util.spec_from_file_location(full, os.path.join(_PKG, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[full] = m
        spec.loader.exec_module(m)
    return sys.modules[full]


J, CF = _mod("jld2"), _mod("chainfile")
import _jld2_writer as W                                  # noqa: E402
W.Writer.lookup3 = staticmethod(J.lookup3)

REF_JLD2 = "/root/reference/dat/default_camb_Cls.jld2"


@pytest.mark.skipif(not os.path.isfile(REF_JLD2), reason="the reference's data file only exists in the build container")
def test_reads_the_reference_s_own_jld2_file():
    f = J.JLD2File(REF_JLD2)
    assert sorted(f.keys()) == ["Cℓ", "params"]
    p = f["params"]
    assert p["r"] == 0.2 and p["ℓmax"] == 16000 and abs(p["ωb"] - 0.0224567) < 1e-12 and p["AL"] == 1      # src/cls.jl:135-141 defaults
    cl = f["Cℓ"]
    z = np.load(os.path.join(ROOT, "tests", "golden", "camb_cls.npz"))
    n = len(z["ell"])

    def spec(group, key):
        v = cl[group][key]
        v = next(x for x in v.values() if isinstance(x, dict) and "Cℓ" in x) if "Cℓ" not in v else v
        assert v["ℓ"]["start"] == 2 and v["ℓ"]["stop"] == 15999
        return v["Cℓ"]
    for g in ("unlensed_scalar", "lensed_scalar", "tensor", "unlensed_total", "total"):
        for k in ("TT", "EE", "BB", "TE"):
            np.testing.assert_array_equal(spec(g, k)[:n], z[f"{g}_{k}"])
    np.testing.assert_array_equal(spec("total", "ϕϕ")[:n], z["phiphi"])


def _hand_built_chain(path, nchains=2, nchunks=2, per_chunk=3, Ny=8, Nx=6):
    """the structure of src/sampling.jl:311-320: rundat + chunks_k = [[state, ...] per chain]; states as filter_for_saving leaves them"""
    rs = np.random.default_rng(0)
    w = W.Writer()
    proj = W.Struct("CMBLensing.ProjLambert{Core.Float64}", {"Ny": Ny, "Nx": Nx, "θpix": 2.0, "ℓy": np.arange(Ny // 2 + 1, dtype=float)})
    truth = {}
    root = {"rundat": w.write(dict(nchains=nchains, nsavemaps=1, nfilewrite=per_chunk, filename="chain.jld2", resume=True, Nbatch=None,
                                   θrange=W.Struct("Core.NamedTuple{(:Aϕ,),Tuple{Vector}}", {"Aϕ": np.array([0.5, 1.0, 1.5])})))}
    step = 0
    for k in range(1, nchunks + 1):
        chunk = [[] for _ in range(nchains)]
        for i in range(per_chunk):
            step += 1
            for c in range(nchains):
                phi = rs.standard_normal((1, Nx, Ny // 2 + 1)) + 1j * rs.standard_normal((1, Nx, Ny // 2 + 1))
                f = rs.standard_normal((2, Nx, Ny // 2 + 1)) + 1j * rs.standard_normal((2, Nx, Ny // 2 + 1))
                st = {"step": step, "lnP": float(-100 + step + c), "ΔH": 0.01 * step, "accept": bool(step % 2), "timing": [0.1, 0.2],
                      "θ": W.Struct("Core.NamedTuple{(:Aϕ,)}", {"Aϕ": 1.0 + 0.1 * c}), "logpdfθ": None,
                      "ϕ": W.Struct("CMBLensing.BaseField{Fourier}", {"arr": phi, "metadata": proj}),
                      "f": W.Struct("CMBLensing.BaseField{EBFourier}", {"arr": f, "metadata": proj})}
                truth[(c, step)] = (phi, f)
                chunk[c].append(st)
        root[f"chunks_{k}"] = w.write(chunk)
    w.close(path, root)
    return truth


def test_reader_on_a_hand_built_chain_file(tmp_path):
    p = str(tmp_path / "chain.jld2")
    truth = _hand_built_chain(p)
    f = J.JLD2File(p)
    assert sorted(f.keys()) == ["chunks_1", "chunks_2", "rundat"]
    rd = J.to_python(f["rundat"])
    assert rd["nchains"] == 2 and rd["filename"] == "chain.jld2" and rd["resume"] is True and rd["Nbatch"] is None
    np.testing.assert_array_equal(rd["θrange"]["Aϕ"], [0.5, 1.0, 1.5])
    ch = J.to_python(f["chunks_2"])
    assert len(ch) == 2 and len(ch[0]) == 3 and ch[1][0]["step"] == 4 and ch[1][0]["accept"] is False
    np.testing.assert_array_equal(ch[1][2]["ϕ"]["arr"], truth[(1, 6)][0])
    assert ch[0][0]["ϕ"]["metadata"]["Ny"] == 8 and ch[0][0]["ϕ"]["__julia_type__"].startswith("CMBLensing.BaseField")


def test_load_chains_opens_the_reference_container(tmp_path):
    p = str(tmp_path / "chain.jld2")
    truth = _hand_built_chain(p)
    assert CF.chunk_indices(p) == [1, 2]
    assert CF.read_rundat(p)["nfilewrite"] == 3
    ch = CF.load_chains(p)
    assert len(ch) == 2 and ch["step"].shape == (2, 6) and list(ch["step"][0]) == [1, 2, 3, 4, 5, 6]
    np.testing.assert_allclose(ch["logpdf"][1], -100 + np.arange(1, 7) + 1)                 # lnP -> logpdf
    np.testing.assert_allclose(ch["dH"][0], 0.01 * np.arange(1, 7))                         # ΔH -> dH
    np.testing.assert_allclose(ch["theta_Aϕ"][:, 0], [1.0, 1.1])                            # NamedTuple θ flattened
    np.testing.assert_array_equal(ch[1, -1, "phi"], truth[(1, 6)][0])                       # Field -> its array, (P, Nx, Nyh) axis order
    np.testing.assert_array_equal(ch[0, 2, "f"], truth[(0, 3)][1])
    assert len(CF.load_chains(p, burnin=2, thin=2)[0]) == 2 and len(CF.load_chains(p, burnin_chunks=1)[0]) == 3
    assert len(CF.load_chains(p, join=True)) == 12 and len(CF.load_chains(p, thin="hasmaps")[0]) == 6
    assert "phi" not in CF.load_chains(p, dropmaps=True)[0][0]
    k, step, last = CF.last_state(p)                                                        # what resume=True continues from
    # the reference stores the initial state as step 1 (src/sampling.jl:268,277): its step 6 is this package's step 5
    assert (k, step) == (3, 5) and last[1]["step"] == 5 and np.array_equal(last[1]["phi"], truth[(1, 6)][0])


def test_checksums_are_verified(tmp_path):
    """lookup3 known answers (Bob Jenkins' own test vectors for hashlittle) and corruption of an object header is detected"""
    assert J.lookup3(b"") == 0xDEADBEEF and J.lookup3(b"", 0xDEADBEEF) == 0xBD5B7DDE
    assert J.lookup3(b"Four score and seven years ago") == 0x17770551 and J.lookup3(b"Four score and seven years ago", 1) == 0xCD628161
    p = str(tmp_path / "chain.jld2")
    _hand_built_chain(p)
    raw = bytearray(open(p, "rb").read())
    i = raw.index(b"OHDR", 600) + 12                                   # a byte inside some object header's message area
    raw[i] ^= 0x40
    open(p, "wb").write(bytes(raw))
    with pytest.raises(J.JLD2Error, match="checksum"):
        f = J.JLD2File(p)
        for k in f.keys():
            f[k]
    assert J.JLD2File(p, verify=False).keys() is not None                # the switch exists for salvage work: opening does not raise


def test_unsupported_structures_fail_loudly(tmp_path):
    p = tmp_path / "x.jld2"
    p.write_bytes(b"not a jld2 file")
    with pytest.raises(J.JLD2Error):
        J.JLD2File(str(p))
    p.write_bytes(b"HDF5-based Julia Data Format, version 0.1.1".ljust(512, b"\0") + b"\x89HDF\r\n\x1a\n" + bytes([0, 8, 8, 0]) + bytes(40))
    with pytest.raises(J.JLD2Error, match="superblock"):
        J.JLD2File(str(p))
