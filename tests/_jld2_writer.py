"""Test infrastructure: a minimal JLD2 WRITER, enough to hand-build the kind of file the reference's `sample_joint` writes
(src/sampling.jl:311-320), so that the reader (cmblensing.jl_amd/jld2.py) and `load_chains` can be exercised without Julia.
Written from the HDF5 file-format specification (v3.0) and the conventions of the JLD2 package; the structures it emits are the
ones the reader's docstring lists:

    offset 0      512-byte text header "HDF5-based Julia Data Format, version 0.1.1 ..."
    offset 512    superblock version 2 (offsets / lengths 8 bytes, base address 512, root group object header address)
    then          objects, each a version-2 object header "OHDR" (flags 0x02: 4-byte chunk size) + its Jenkins lookup3 checksum (the
                  reader verifies it); all addresses relative to the base address
    groups        link messages (type 0x06, version 1, flags 0x10|size bits: charset byte + 1-byte name length), hard links
    datasets      dataspace v2 (scalar / simple), datatype (inline or shared -> committed datatype object), layout v3 contiguous
    committed dt  object header holding the datatype message + attribute "julia_type" (here a fixed-length string; real JLD2
                  files carry a compound {name, parameters}, which tests/test_jld2.py covers with the reference's own data file)
    Julia values  Float64 / Int64 scalars and arrays (column-major: dims reversed), ComplexF64 arrays = compound {re, im},
                  String / Symbol = variable-length string in a global heap collection "GCOL", Vector{Any} = array of 8-byte object
                  references, struct / NamedTuple = compound whose non-isbits members are references, Dict{Symbol,Any} = (JLD2's
                  custom serialisation) 1-D array of Pair{Symbol,Any} compounds {first: vlen string, second: reference},
                  `nothing` = a scalar of an empty (size 0) committed datatype.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _msg(t, body, flags=0):
    return bytes([t]) + struct.pack("<H", len(body)) + bytes([flags]) + body


def _dt_fixed(size, signed=True):
    return bytes([0x10 | 0, 0x08 if signed else 0, 0, 0]) + struct.pack("<I", size) + struct.pack("<HH", 0, size * 8)


def _dt_float(size):
    props = {8: struct.pack("<HHBBBBII", 0, 64, 52, 11, 0, 52, 1023, 0), 4: struct.pack("<HHBBBBII", 0, 32, 23, 8, 0, 23, 127, 0)}[size]
    return bytes([0x10 | 1, 0x20, 0x3F if size == 8 else 0x1F, 0]) + struct.pack("<I", size) + props[:12]


DT_REF = bytes([0x10 | 7, 0, 0, 0]) + struct.pack("<I", 8)
DT_VLEN_STR = bytes([0x10 | 9, 0x01 | 0x10, 0, 0]) + struct.pack("<I", 16) + (bytes([0x10 | 3, 0x11, 0, 0]) + struct.pack("<I", 1))


def _dt_compound(members, size):
    """members: [(name, offset, datatype message bytes)] -- version 3 compound"""
    nb = 1 if size < 256 else 2 if size < 65536 else 4
    body = b"".join(name.encode() + b"\0" + off.to_bytes(nb, "little") + dt for name, off, dt in members)
    return bytes([0x30 | 6, len(members) & 0xFF, len(members) >> 8, 0]) + struct.pack("<I", size) + body


DT_COMPLEX = _dt_compound([("re", 0, _dt_float(8)), ("im", 8, _dt_float(8))], 16)


def _shared(addr):
    return bytes([3, 2]) + struct.pack("<Q", addr)


class Writer:
    lookup3 = None                                                        # set by the test module: the reader's checksum function

    def __init__(self):
        self.buf = bytearray(b"HDF5-based Julia Data Format, version 0.1.1\0 (hand-built test file)".ljust(512, b"\0"))
        self.buf += bytes(48)                                             # superblock, filled in by close()
        self.types = {}

    def _addr(self):
        return len(self.buf) - 512

    def _object(self, msgs):
        body = b"".join(msgs)
        a = self._addr()
        hdr = b"OHDR" + bytes([2, 0x02]) + struct.pack("<I", len(body)) + body
        self.buf += hdr + struct.pack("<I", self.lookup3(hdr))
        return a

    def _data(self, raw):
        a = self._addr()
        self.buf += raw
        return a

    def _heap(self, objs):
        """one global heap collection holding `objs` (list of bytes) -> (address, [index])"""
        body = b""
        for i, o in enumerate(objs):
            body += struct.pack("<HHIQ", i + 1, 1, 0, len(o)) + o + bytes(-len(o) % 8)
        body += bytes(16)                                                 # free-space object (index 0)
        a = self._addr()
        self.buf += b"GCOL" + bytes([1, 0, 0, 0]) + struct.pack("<Q", 16 + len(body)) + body
        return a, list(range(1, len(objs) + 1))

    def committed(self, julia_type, dt):
        """commit a datatype with its `julia_type` attribute (memoised per Julia type name)"""
        if julia_type not in self.types:
            name = julia_type.encode() + b"\0"
            sdt = bytes([0x10 | 3, 0x00, 0, 0]) + struct.pack("<I", len(name))
            space = bytes([2, 0, 0, 0])
            attr = bytes([3, 0]) + struct.pack("<HHH", len(b"julia_type\0"), len(sdt), len(space)) + bytes([0]) + b"julia_type\0" + sdt + space + name
            self.types[julia_type] = self._object([_msg(0x03, dt), _msg(0x0C, attr)])
        return self.types[julia_type]

    def _dataset(self, dt_msg_body, shared, dims, raw):
        space = bytes([2, 0, 0, 0]) if dims is None else bytes([2, len(dims), 0, 1]) + struct.pack("<" + "Q" * len(dims), *dims)
        addr = self._data(raw) if raw else UNDEF
        layout = bytes([3, 1]) + struct.pack("<QQ", addr, len(raw))
        return self._object([_msg(0x01, space), _msg(0x03, dt_msg_body, 0x02 if shared else 0), _msg(0x08, layout)])

    # ---- Julia values ---------------------------------------------------------------------------------------------------------------
    def write(self, v):
        """-> address of the dataset holding the Python value `v` as Julia would have it"""
        if v is None:
            return self._dataset(_shared(self.committed("Core.Nothing", bytes([0x30 | 6, 0, 0, 0]) + struct.pack("<I", 0))), True, None, b"")
        if isinstance(v, bool):
            return self._dataset(_shared(self.committed("Core.Bool", bytes([0x10 | 4, 0, 0, 0]) + struct.pack("<I", 1) + struct.pack("<HH", 0, 8))), True, None, bytes([v]))
        if isinstance(v, int):
            return self._dataset(_dt_fixed(8), False, None, struct.pack("<q", v))
        if isinstance(v, float):
            return self._dataset(_dt_float(8), False, None, struct.pack("<d", v))
        if isinstance(v, str):
            ha, (i,) = self._heap([v.encode()])
            return self._dataset(DT_VLEN_STR, False, None, struct.pack("<IQI", len(v.encode()), ha, i))
        if isinstance(v, np.ndarray):
            dims = v.shape                                                # NumPy (.., Nx, Ny) == Julia (Ny, Nx, ..): HDF5 dims reversed Julia
            if np.iscomplexobj(v):
                return self._dataset(DT_COMPLEX, False, dims, np.ascontiguousarray(v, np.complex128).tobytes())
            if v.dtype.kind == "i":
                return self._dataset(_dt_fixed(8), False, dims, np.ascontiguousarray(v, "<i8").tobytes())
            return self._dataset(_dt_float(8), False, dims, np.ascontiguousarray(v, "<f8").tobytes())
        if isinstance(v, list):                                           # Vector{Any}
            refs = [self.write(e) for e in v]
            return self._dataset(DT_REF, False, (len(refs),), struct.pack("<" + "Q" * len(refs), *refs))
        if isinstance(v, Struct):                                         # struct / NamedTuple: every member a reference
            refs = [self.write(e) for e in v.fields.values()]
            dt = _dt_compound([(k, 8 * i, DT_REF) for i, k in enumerate(v.fields)], 8 * len(refs))
            return self._dataset(_shared(self.committed(v.julia_type, dt)), True, None, struct.pack("<" + "Q" * len(refs), *refs))
        if isinstance(v, dict):                                           # Dict{Symbol,Any} -> Vector{Pair{Symbol,Any}}
            refs = [self.write(e) for e in v.values()]
            ha, idx = self._heap([k.encode() for k in v])
            pair = _dt_compound([("first", 0, DT_VLEN_STR), ("second", 16, DT_REF)], 24)
            raw = b"".join(struct.pack("<IQIQ", len(k.encode()), ha, i, r) for k, i, r in zip(v, idx, refs))
            return self._dataset(_shared(self.committed("Base.Dict{Core.Symbol,Core.Any}", pair)), True, (len(refs),), raw)
        raise TypeError(type(v))

    def close(self, path, root):
        """root: {name: address}"""
        links = [_msg(0x06, bytes([1, 0x10, 0, len(k.encode())]) + k.encode() + struct.pack("<Q", a)) for k, a in root.items()]
        ra = self._object(links)
        sb = b"\x89HDF\r\n\x1a\n" + bytes([2, 8, 8, 0]) + struct.pack("<QQQQ", 512, UNDEF, len(self.buf), ra)
        sb += struct.pack("<I", self.lookup3(sb))
        self.buf[512:512 + len(sb)] = sb
        with open(path, "wb") as fh:
            fh.write(bytes(self.buf))


class Struct:
    """fields as a dict, not keyword arguments: Python NFKC-normalises identifiers (ϕ U+03D5 -> φ U+03C6), Julia does not"""

    def __init__(self, julia_type, fields):
        self.julia_type, self.fields = julia_type, dict(fields)
