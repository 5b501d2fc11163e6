"""The JLD2 WRITER (cmblensing.jl_amd/jld2_writer.py): chain files in the reference's own container (src/sampling.jl:311-320).
  * round trip: every value kind, the chain-file layout, append ("a+"), through the reader with all lookup3 checksums verified;
  * structure: the same message kinds, versions and flags as the file the JLD2 package itself wrote (the reference's
    dat/default_camb_Cls.jld2; only where /root/reference exists) -- superblock, root group, `_types` group, committed datatypes with
    `julia_type` as a value of the first committed datatype, DataType / Int64 parameter datasets, global heap collections;
  * the chain-file API (`write_chunk` / `load_chains` / `last_state`) on `.jld2` file names, and the read-only rule for files the Julia
    package wrote.
No Julia exists here: that JLD2.jl opens these files is NOT tested (stated in jld2_writer.py and INTEGRATION.md)."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_jld2 import _mod, REF_JLD2, J, CF              # noqa: E402

W = _mod("jld2_writer")


def _state(rng, step):
    return {"step": step, "logpdf": -12.5 - step, "dH": 0.25, "accept": True, "ncg": 17,
            "phi": (rng.standard_normal((4, 3)) + 1j * rng.standard_normal((4, 3))), "f": (rng.standard_normal((2, 4, 3)) + 0j),
            "theta_r": 0.21, "theta_Aphi": 1.1}


def test_round_trip_of_every_value_kind(tmp_path):
    fn = str(tmp_path / "v.jld2")
    rng = np.random.default_rng(0)
    vals = {"b": True, "i": -7, "x": 2.5, "s": "Größe ϕ", "sym": W.Symbol("diag"), "none": None,
            "f32": rng.standard_normal((2, 3, 5)).astype(np.float32), "f64": rng.standard_normal((7,)), "i64": np.arange(6).reshape(2, 3),
            "c64": (rng.standard_normal((3, 4)) + 1j * rng.standard_normal((3, 4))).astype(np.complex64),
            "c128": rng.standard_normal((5, 2)) + 1j * rng.standard_normal((5, 2)), "mask": np.array([True, False, True]),
            "big": rng.standard_normal((64, 65)),                       # > 8 KiB: contiguous layout instead of compact
            "list": [1, 2.0, "x", [3, 4]], "nested": {"a": 1.5, "inner": {"k": "v"}}, "empty": [], "emptyd": {}}
    with W.JLD2Writer(fn, "w") as w:
        w.write("vals", vals)
        with pytest.raises(KeyError):
            w.write("vals", 1)                                          # JLD2 never overwrites a dataset
    f = J.JLD2File(fn, verify=True)
    got = J.to_python(f["vals"])
    assert "none" not in got                                            # keys holding None are dropped
    for k in ("b", "i", "x", "s", "sym"):
        assert got[k] == vals[k] and type(got[k]) is type(vals[k]) or (k == "sym" and got[k] == "diag")
    for k in ("f32", "f64", "i64", "c64", "c128", "mask", "big"):
        assert got[k].dtype == vals[k].dtype and got[k].shape == vals[k].shape, k
        np.testing.assert_array_equal(got[k], vals[k])
    assert got["list"] == [1, 2.0, "x", [3, 4]] and got["nested"] == {"a": 1.5, "inner": {"k": "v"}}
    assert got["empty"] == [] and got["emptyd"] in ({}, [])


def test_chain_file_layout_append_and_checksums(tmp_path):
    fn = str(tmp_path / "chain.jld2")
    rng = np.random.default_rng(1)
    chunk1 = [[_state(rng, s) for s in (1, 2)] for _ in range(3)]
    chunk2 = [[_state(rng, s) for s in (3, 4)] for _ in range(3)]
    CF.check_filename(fn, None)
    CF.write_chunk(fn, 1, chunk1, rundat=dict(nchains=3, eps=0.01, rng="device", filename=None), clobber=True)
    size1 = os.path.getsize(fn)
    head = open(fn, "rb").read(size1)
    with pytest.raises(ValueError):
        CF.check_filename(fn, None)                                    # exists: resume must be explicit (src/sampling.jl:239-241)
    CF.write_chunk(fn, 2, chunk2)
    assert open(fn, "rb").read(size1)[560:] == head[560:]              # "a+": everything behind the superblock is left where it was
    f = J.JLD2File(fn, verify=True)
    assert f.keys() == ["rundat", "chunks_1", "chunks_2"] and CF.chunk_indices(fn) == [1, 2]
    rd = CF.read_rundat(fn)
    assert rd["nchains"] == 3 and rd["eps"] == 0.01 and rd["rng"] == "device" and "filename" not in rd
    # the reference's key names are what is stored (src/sampling.jl:388-464) ...
    raw = J.to_python(f["chunks_1"])
    assert len(raw) == 3 and len(raw[0]) == 2
    # `step` and `logpdf` are the reference's own names (src/sampling.jl:290 `setindex!.(states, step, :step)`, :446 `@pack! state = f̃, logpdf`):
    # `@unpack step = states[1]` (:256) of a resuming Julia session and `chain[:logpdf]` need exactly these; `i` / `lnP` appear nowhere
    assert set(raw[0][0]) == {"step", "logpdf", "ΔH", "accept", "ncg", "ϕ", "f", "θ"} and raw[0][0]["θ"] == {"r": 0.21, "Aphi": 1.1}
    # the stored counter is the reference's: initial state = step 1, first Gibbs pass = step 2 (:263,288)
    assert [s["step"] for s in raw[0]] == [2, 3] and raw[0][0]["logpdf"] == -13.5
    # ... and load_chains hands back this package's key names with the file's (= the reference's) step numbers, like a Julia-written file
    ch = CF.load_chains(fn)
    assert len(ch) == 3 and len(ch[0]) == 4
    np.testing.assert_array_equal(ch["step"], np.tile(np.arange(2, 6), (3, 1)))
    np.testing.assert_array_equal(ch[1, 2]["phi"], chunk2[1][0]["phi"])
    assert ch[0, 0]["theta_r"] == 0.21 and ch[2, -1]["logpdf"] == -16.5
    k, step, last = CF.last_state(fn)
    assert (k, step) == (3, 4) and not CF.written_by_julia(fn)
    np.testing.assert_array_equal(last[2]["phi"], chunk2[2][1]["phi"])
    # a truncated file is detected by the checksums / bounds, not decoded into garbage
    open(str(tmp_path / "cut.jld2"), "wb").write(open(fn, "rb").read()[:-40])
    with pytest.raises(Exception):
        J.JLD2File(str(tmp_path / "cut.jld2"), verify=True)["chunks_2"]


def _kinds(f, off):
    return [(t, fl, d[0] if t in (0x01, 0x05, 0x08, 0x0C) else None) for t, fl, d in f._messages(off)]


def test_structure_follows_the_conventions_of_jld2(tmp_path):
    """what can be checked without the reference's file: the conventions jld2_writer.py lists, read back byte-wise"""
    fn = str(tmp_path / "s.jld2")
    with W.JLD2Writer(fn, "w") as w:
        w.write("d", {"a": np.ones((2, 2), np.complex64), "s": W.Symbol("x")})
    b = open(fn, "rb").read()
    assert b.startswith(b"HDF5-based Julia Data Format, version 0.1.1\0") and b[512:520] == b"\x89HDF\r\n\x1a\n" and b[520:524] == bytes([2, 8, 8, 0])
    base, ext, eof, root = struct.unpack_from("<QQQQ", b, 524)
    assert (base, ext, eof) == (512, J.UNDEF, len(b))
    f = J.JLD2File(fn, verify=True)
    f.keys()
    types = f._links(f._messages(f._root_links["_types"]))
    assert sorted(types) == [f"{i + 1:08d}" for i in range(len(types))] and types["00000001"] == 48      # first object behind the superblock
    names = {k: f._committed_type(a).julia_type for k, a in types.items()}
    assert names["00000001"] == "Core.DataType"
    assert set(names.values()) == {"Core.DataType", "Core.Symbol", "Base.Pair{Core.Symbol,Core.Any}", "Base.Dict{Core.Symbol,Core.Any}",
                                   "Base.Complex{Core.Float32}"}
    d = f._committed_type(types[[k for k, v in names.items() if v.startswith("Base.Dict")][0]])
    assert d.cls == 7 and d.size == 8 and d.written_type == "Core.Array{Base.Pair{Core.Symbol,Core.Any},1}"
    # every committed datatype: datatype message with flags 0x40, then julia_type as an attribute of the FIRST committed datatype
    for a in types.values():
        m = f._messages(a)
        assert (m[0][0], m[0][1]) == (0x03, 0x40) and m[1][0] == 0x0C and m[1][2][:2] == bytes([2, 1]) and b"julia_type\0" in m[1][2][:24]
        assert m[1][2][19:29] == bytes([3, 2]) + struct.pack("<Q", 48)
    # datasets: fill value (version 3, flags 9), dataspace version 2, datatype, layout version 4
    assert _kinds(f, f._root_links["d"]) == [(0x05, 0, 3), (0x01, 0, 2), (0x03, 3, None), (0x08, 0, 4)]
    # groups: link info + group info + links
    assert [t for t, _, _ in f._messages(f.root)][:2] == [0x02, 0x0A]
    # global heap collections: >= 4096 bytes, objects numbered from 1, free-space object last
    p = b.index(b"GCOL")
    size = struct.unpack_from("<Q", b, p + 8)[0]
    assert b[p + 4] == 1 and size >= 4096 and struct.unpack_from("<H", b, p + 16)[0] == 1


@pytest.mark.skipif(not os.path.isfile(REF_JLD2), reason="the reference's data file only exists in the build container")
def test_same_structures_as_the_file_jld2_itself_wrote(tmp_path):
    """message kinds, versions and flags of each object class, side by side with dat/default_camb_Cls.jld2"""
    fn = str(tmp_path / "s.jld2")
    with W.JLD2Writer(fn, "w") as w:
        w.write("x", {"a": 1.0, "v": np.arange(3.0)})
    mine, ref = J.JLD2File(fn), J.JLD2File(REF_JLD2)
    mine.keys(); ref.keys()
    bm, br = open(fn, "rb").read(), open(REF_JLD2, "rb").read()
    assert bm[512:524] == br[512:524]                                   # superblock signature, version, offset / length sizes
    assert struct.unpack_from("<QQ", bm, 524) == struct.unpack_from("<QQ", br, 524)      # base address 512, no extension
    # root group and _types group: the same message kinds in the same order (links differ in number only)
    strip = lambda ms: [t for t, _, _ in ms if t != 0x06]
    assert strip(mine._messages(mine.root)) == strip(ref._messages(ref.root))
    tm, tr = mine._links(mine._messages(mine._root_links["_types"])), ref._links(ref._messages(ref._root_links["_types"]))
    assert tm["00000001"] == tr["00000001"] == 48
    # the DataType datatype is identical byte for byte (datatype message and the self-describing attribute's layout)
    m1, r1 = mine._messages(48), ref._messages(48)
    assert m1[0] == r1[0] and (m1[1][0], m1[1][1], m1[1][2][:33]) == (r1[1][0], r1[1][1], r1[1][2][:33])
    # Symbol: committed variable-length string, the same datatype message
    sym_m = [a for a in tm.values() if mine._committed_type(a).julia_type == "Core.Symbol"][0]
    sym_r = [a for a in tr.values() if ref._committed_type(a).julia_type == "Core.Symbol"][0]
    assert mine._messages(sym_m)[0] == ref._messages(sym_r)[0]
    # a custom-serialised type (here Dict, there NTuple{13,Symbol} written as Array{Symbol,1}): reference datatype + julia_type + written_type
    dm = [a for a in tm.values() if mine._committed_type(a).written_type][0]
    dr = [a for a in tr.values() if ref._committed_type(a).written_type and ref._committed_type(a).cls == 7][0]
    km = [(t, fl, d[:8] if t == 0x03 else d[8:8 + 10]) for t, fl, d in mine._messages(dm)]
    kr = [(t, fl, d[:8] if t == 0x03 else d[8:8 + 10]) for t, fl, d in ref._messages(dr)]
    assert km == kr
    # a DataType-valued parameter dataset (Core.Any here, Core.Float64 there) and a plain dataset: the same four messages
    any_m = [o for o in mine.values_debug()] if hasattr(mine, "values_debug") else None
    pm = _kinds(mine, mine._root_links["x"])
    pr = _kinds(ref, ref._root_links["params"])
    assert pm == pr


def test_files_of_the_julia_package_are_read_only_here(tmp_path):
    """a hand-built file in the layout the Julia package writes (tests/_jld2_writer.py): analysable, resumable INTO A NEW FILE with the
    step counter shifted and a Map-basis ϕ transformed, never appended to"""
    import _jld2_writer as TW
    TW.Writer.lookup3 = staticmethod(J.lookup3)
    fn = str(tmp_path / "julia.jld2")
    w = TW.Writer()
    phi_map = np.random.default_rng(3).standard_normal((1, 4, 6))
    field = lambda a: TW.Struct("CMBLensing.BaseField{CMBLensing.Map,CMBLensing.ProjLambert,Core.Float64,Core.Array{Core.Float64,3}}",
                                {"arr": a, "metadata": TW.Struct("CMBLensing.ProjLambert", {"Ny": 6, "Nx": 4})})
    states = [[{"i": 1, "lnP": -1.0, "ϕ": field(phi_map), "f": field(np.ones((2, 4, 6)))}, {"i": 2, "lnP": -2.0, "ϕ": field(2 * phi_map), "f": field(np.ones((2, 4, 6)))}]]
    w.close(fn, {"rundat": w.write({"nchains": 1}), "chunks_1": w.write(states)})
    assert CF.written_by_julia(fn)
    with pytest.raises(ValueError, match="read-only"):
        CF.check_filename(fn, True)
    k, step, last = CF.last_state(fn)
    assert (k, step) == (2, 1) and last[0]["step"] == 1                 # the reference's step 2 = this package's step 1
    np.testing.assert_allclose(last[0]["phi"], np.fft.rfft2(2 * phi_map, axes=(-2, -1)))
    from_writer = None
    with pytest.raises(ValueError, match="this package only"):
        W.JLD2Writer(fn, "a")


JULIA_CHAIN = os.environ.get("CMBL_JULIA_CHAIN_FIXTURE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_outputs", "chain_fixture.jld2")


@pytest.mark.skipif(not os.path.isfile(JULIA_CHAIN), reason="parity unpinned: tests/golden/ref_outputs/chain_fixture.jld2 (a chain file written by JLD2.jl itself, "
                    "julia/make_reference_fixtures.jl) is absent -- no Julia in the build image; the `.jld2` output of jld2_writer.py stays experimental")
def test_same_messages_as_a_chain_file_jld2_jl_wrote(tmp_path):
    """The file JLD2.jl writes for a `Vector{Vector{Any}}` of `Dict{Symbol,Any}` states against the file jld2_writer.py writes for the SAME content:
    same values through the reader, and for every object class (root links, the `Dict` datasets, their committed datatypes, the `Vector{Any}` of object
    references) the same header messages -- kinds, flags and the layout-defining leading bytes -- in the same order."""
    ref = J.JLD2File(JULIA_CHAIN, verify=True)
    def state(step, c):
        return {"step": step - 1, "logpdf": -100.0 - step - c, "dH": 0.25 * step, "accept": bool(step % 2), "ncg": 17.0,
                "phi": np.array([[x + 10 * y + 1j * c for x in range(1, 4)] for y in range(1, 5)], np.complex128),               # NumPy (Nx, Ny) == Julia (Ny, Nx)
                "f": np.array([[[x - y + 1j * step for x in range(1, 4)] for y in range(1, 5)] for p in range(2)], np.complex128),
                "theta_r": 0.21, "theta_Aphi": 1.1}
    fn = str(tmp_path / "mine.jld2")
    CF.write_chunk(fn, 1, [[state(step, c) for step in (2, 3)] for c in (0, 1)], rundat=dict(nchains=2, eps=0.01, rng="device"), clobber=True)
    CF.write_chunk(fn, 2, [[state(4, c)] for c in (0, 1)])
    mine = J.JLD2File(fn, verify=True)
    assert sorted(mine.keys()) == sorted(ref.keys()) == ["chunks_1", "chunks_2", "rundat"]
    # the same values, read through the same reader
    for key in ("chunks_1", "chunks_2"):
        a, b = J.to_python(mine[key]), J.to_python(ref[key])
        assert len(a) == len(b) and all(len(x) == len(y) for x, y in zip(a, b))
        for ca, cb in zip(a, b):
            for sa, sb in zip(ca, cb):
                assert set(sa) == set(sb), (set(sa), set(sb))
                for k in sa:
                    va, vb = CF._jld2_value(sa[k]), CF._jld2_value(sb[k])
                    if isinstance(va, dict):
                        assert va == vb
                    else:
                        np.testing.assert_array_equal(np.asarray(va), np.asarray(vb))
    # the same object structure: message kinds and flags of the dataset headers and of the committed datatypes they point to
    kinds = lambda f, addr: [(t, fl) for t, fl, _ in f._messages(addr) if t != 0x00]
    for key in ("rundat", "chunks_1"):
        assert kinds(mine, mine._root_links[key]) == kinds(ref, ref._root_links[key]), key
    tm = {mine._committed_type(a).julia_type: a for a in mine._links(mine._messages(mine._root_links["_types"])).values()}
    tr = {ref._committed_type(a).julia_type: a for a in ref._links(ref._messages(ref._root_links["_types"])).values()}
    assert set(tr) <= set(tm) | {t for t in tr if t.startswith("Core.")}, (set(tr) - set(tm))
    for name in set(tm) & set(tr):
        assert kinds(mine, tm[name]) == kinds(ref, tr[name]), name
        cm, cr = mine._committed_type(tm[name]), ref._committed_type(tr[name])
        assert (cm.cls, cm.written_type) == (cr.cls, cr.written_type), name
