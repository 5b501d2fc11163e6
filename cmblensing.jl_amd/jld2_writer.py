"""Writer for JLD2 chain files -- the container `sample_joint` of the reference appends to (src/sampling.jl:311-320:
`jldopen(filename, clobber ? "w" : "a+")`, `write(io, "rundat", ...)`, `write(io, "chunks_$k", chain_chunks)`) and `load_chains`
reads back (src/chains.jl:48-100).  The counterpart of the reader in jld2.py.

No Julia and no HDF5 library exist in the build image, so every structure below is emitted byte for byte after (a) the HDF5 file
format specification (version 3.0) and (b) the way the JLD2 package lays those structures out in the reference's own data file
`dat/default_camb_Cls.jld2` (decoded message by message with jld2.py; the tests compare the two files structurally):

  offset 0      512-byte text header "HDF5-based Julia Data Format, version 0.1.1"
  offset 512    version-2 superblock: base address 512, end-of-file address (absolute), root group object header (relative), lookup3
  objects       version-2 object headers "OHDR" (1- or 2-byte chunk size), message header = type, size (2), flags; lookup3 checksum
  groups        link info (0x02, no fractal heap) + group info (0x0A) + hard link messages (0x06, UTF-8 names); the root group links
                the user's datasets and `_types`, the group of committed datatypes named 00000001, 00000002, ...
  datatypes     Float32/64, Int64 inline; String = variable-length UTF-8 string inline; Symbol = the same, committed; Bool = 1-byte bit
                field; Complex{T} = committed compound {re, im}; every committed datatype carries the attribute `julia_type`, a
                value of the FIRST committed datatype -- the compound {name: vlen string, parameters: vlen of references} that
                describes a Julia `DataType` -- whose parameters point at other committed datatypes, or at small datasets holding
                a DataType value (`Core.Float64`, `Core.Any`: types without a committed datatype of their own) or an Int64
  datasets      fill value (0x05: version 3, undefined) + dataspace (version 2; dims = Julia's reversed) + datatype (inline or
                shared) + data layout version 4, compact (<= 8 KiB) or contiguous
  variable-length data in global heap collections "GCOL" of >= 4096 bytes (objects numbered 1, 2, ... each padded to 8 bytes, then
                the free-space object 0)
  Vector{Any}   a dataset of 8-byte object references (no `julia_type`: `Any` is the default element type of a reference array)
  Dict{Symbol,Any}  JLD2 serialises a Dict through `Vector{Pair{K,V}}` (its custom-serialisation rule `writeas`): the dataset has a
                committed REFERENCE datatype with `julia_type = Base.Dict{Core.Symbol,Core.Any}` and `written_type =
                Core.Array{Base.Pair{Core.Symbol,Core.Any},1}` -- the same construction the data file shows for `NTuple{13,Symbol}`
                written as `Array{Symbol,1}` -- and its one reference points at the array of `Pair` compounds {first: vlen string,
                second: reference} (committed, with `field_datatypes`).

What Python values become:  None is not written (keys holding None are dropped);  bool -> Bool;  int -> Int64;  float -> Float64;
str -> String;  `Symbol(str)` -> Symbol;  ndarray of float32 / float64 / int64 / complex64 / complex128 / bool -> Array{T,N} in Julia's
axis order (a NumPy (P, Nx, Ny) array is the Julia (Ny, Nx, P) array, as everywhere in this package);  list / tuple -> Vector{Any};
dict with str keys -> Dict{Symbol,Any}.  Fields are written as their arrays, not as `BaseField` structs (those carry the whole
`ProjLambert` metadata as type parameters): the Julia `load_chains` returns chains of `Dict{Symbol,Any}` whose map entries are plain
arrays.

STATUS: round-trips through jld2.py with checksum verification (tests/test_jld2_writer.py) and follows the data file's conventions
structure by structure, but has **never been opened by JLD2.jl itself** (no Julia here); in particular the Dict construction above is
inferred by analogy, not observed.  INTEGRATION.md says the same.
"""
import os
import struct

import numpy as np

from .jld2 import JLD2File, lookup3, UNDEF

BASE = 512
COMPACT_MAX = 8192
HEAP_MIN = 4096


class Symbol(str):
    """a Julia Symbol (as opposed to a String)"""


# ---- datatype messages -------------------------------------------------------------------------------------------------------------
def _dt_float(size):
    props = {8: struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023), 4: struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)}[size]
    return bytes([0x31, 0x20, 0x3F if size == 8 else 0x1F, 0]) + struct.pack("<I", size) + props


def _dt_int(size, signed=True):
    return bytes([0x30, 0x08 if signed else 0x00, 0, 0]) + struct.pack("<I", size) + struct.pack("<HH", 0, 8 * size)


DT_BOOL = bytes([0x34, 0, 0, 0]) + struct.pack("<I", 1) + struct.pack("<HH", 0, 8)
DT_REF = bytes([0x37, 0, 0, 0]) + struct.pack("<I", 8)
DT_VLEN_STR = bytes([0x39, 0x11, 0x01, 0]) + struct.pack("<I", 16) + _dt_int(1, signed=False)
DT_VLEN_REF = bytes([0x39, 0x00, 0x00, 0]) + struct.pack("<I", 16) + DT_REF


def _dt_compound(members, size):
    """members: [(name, offset, datatype message bytes)] -- version 3 compound"""
    nb = 1 if size < 256 else 2 if size < 65536 else 4
    body = b"".join(name.encode() + b"\0" + off.to_bytes(nb, "little") + dt for name, off, dt in members)
    return bytes([0x36, len(members) & 0xFF, len(members) >> 8, 0]) + struct.pack("<I", size) + body


DT_DATATYPE = _dt_compound([("name", 0, DT_VLEN_STR), ("parameters", 16, DT_VLEN_REF)], 32)


def _shared(addr):
    return bytes([3, 2]) + struct.pack("<Q", addr)


def _msg(t, body, flags=0):
    return bytes([t]) + struct.pack("<H", len(body)) + bytes([flags]) + body


SPACE_SCALAR = bytes([2, 0, 0, 0])


def _space(dims):
    return SPACE_SCALAR if dims is None else bytes([2, len(dims), 0, 1]) + struct.pack("<" + "Q" * len(dims), *dims)


_NP = {np.dtype("float32"): ("Core.Float32", lambda: _dt_float(4)), np.dtype("float64"): ("Core.Float64", lambda: _dt_float(8)),
       np.dtype("int64"): ("Core.Int64", lambda: _dt_int(8)), np.dtype("int32"): ("Core.Int32", lambda: _dt_int(4))}


class JLD2Writer:
    def __init__(self, path, mode="w"):
        """mode "w": new file; "a": append datasets to a file THIS writer produced (the reference's `jldopen(filename, "a+")`)"""
        self.path = path
        self.types = {}                  # julia type string -> address of its committed datatype
        self.type_order = []             # addresses in commit order (names 00000001, ...)
        self.values = {}                 # memo of small DataType / Int64 parameter datasets
        self.links = {}                  # root group: name -> address
        self._heap = None                # (address, bytes used, capacity, next index) of the open global heap collection
        self.off = 0                     # absolute file offset of buf[0]: appending keeps only the new tail in memory
        if mode == "a" and os.path.exists(path):
            rd = JLD2File(path)
            rd.keys()
            links = dict(rd._root_links)
            tg = links.pop("_types", None)
            if not bytes(rd.buf[:120]).split(b"\0")[1].startswith(b" (cmblensing.jl_amd") or tg is None:
                raise ValueError(f"{path}: appending is supported for files written by this package only")
            self.off = len(rd.buf)
            self.buf = bytearray()
            self.links = links
            tl = rd._links(rd._messages(tg))
            for name in sorted(tl):
                dt = rd._committed_type(tl[name])
                self.types[dt.julia_type] = tl[name]
                self.type_order.append(tl[name])
        elif mode in ("w", "a"):
            hdr = b"HDF5-based Julia Data Format, version 0.1.1\0 (cmblensing.jl_amd jld2_writer, 64-bit LE)"
            self.buf = bytearray(hdr.ljust(BASE, b"\0")) + bytearray(48)
            self._commit("Core.DataType", DT_DATATYPE, name="Core.DataType", params=[])     # always the first committed datatype
        else:
            raise ValueError(mode)

    # ---- low level ----------------------------------------------------------------------------------------------------------------
    def _addr(self):
        return self.off + len(self.buf) - BASE

    def _pos(self, addr):
        """index into buf of relative file address `addr`"""
        return BASE + addr - self.off

    @staticmethod
    def _object_bytes(msgs):
        body = b"".join(msgs)
        hdr = b"OHDR" + (bytes([2, 0x00, len(body)]) if len(body) < 256 else bytes([2, 0x01]) + struct.pack("<H", len(body)) if len(body) < 65536
                         else bytes([2, 0x02]) + struct.pack("<I", len(body))) + body
        return hdr + struct.pack("<I", lookup3(hdr))

    def _object(self, msgs):
        a = self._addr()
        self.buf += self._object_bytes(msgs)
        return a

    def _heap_put(self, data):
        """store one object in a global heap collection -> (collection address, index)"""
        need = 16 + ((len(data) + 7) & ~7)
        if self._heap is None or self._heap[1] + need + 16 > self._heap[2]:
            self._heap_close()
            cap = max(HEAP_MIN, (16 + need + 16 + 7) & ~7)
            a = self._addr()
            self.buf += b"GCOL" + bytes([1, 0, 0, 0]) + struct.pack("<Q", cap) + bytes(cap - 16)
            self._heap = [a, 16, cap, 1]
        a, used, cap, idx = self._heap
        p = self._pos(a) + used
        self.buf[p:p + 16] = struct.pack("<HHIQ", idx, 0, 0, len(data))
        self.buf[p + 16:p + 16 + len(data)] = data
        self._heap[1] = used + need
        self._heap[3] = idx + 1
        return a, idx

    def _heap_close(self):
        if self._heap is not None:
            a, used, cap, _ = self._heap
            if cap - used >= 16:                                         # free-space object: index 0, its size covers the rest
                self.buf[self._pos(a) + used:self._pos(a) + used + 16] = struct.pack("<HHIQ", 0, 0, 0, cap - used)
            self._heap = None

    def _vlen(self, data, n=None):
        """the 16-byte variable-length element {length, heap collection address, index}"""
        if len(data) == 0:
            return struct.pack("<IQI", 0, 0, 0)
        a, i = self._heap_put(data)
        return struct.pack("<IQI", len(data) if n is None else n, a, i)

    def _dataset(self, dt_body, shared, dims, raw, attrs=()):
        layout = bytes([4, 0]) + struct.pack("<H", len(raw)) + raw if len(raw) <= COMPACT_MAX else None
        if layout is None:
            a = self._addr()
            self.buf += raw
            layout = bytes([4, 1]) + struct.pack("<QQ", a, len(raw))
        return self._object([_msg(0x05, bytes([3, 9])), _msg(0x01, _space(dims)), _msg(0x03, dt_body, 0x03 if shared else 0x01), _msg(0x08, layout)]
                            + [_msg(0x0C, a) for a in attrs])

    # ---- Julia type descriptions ------------------------------------------------------------------------------------------------------
    def _datatype_value(self, name, params):
        """32 bytes: an instance of the DataType compound {name, parameters}"""
        refs = b"".join(struct.pack("<Q", r) for r in params)
        return self._vlen(name.encode()) + self._vlen(refs, len(params))

    def _type_attr(self, attr, name, params):
        an = attr.encode() + b"\0"
        dt = _shared(self.type_order[0] if self.type_order else 0)
        return bytes([2, 1]) + struct.pack("<HHH", len(an), len(dt), len(SPACE_SCALAR)) + an + dt + SPACE_SCALAR + self._datatype_value(name, params)

    def _commit(self, key, dt, name, params, written=None, field_types=None):
        """commit datatype `dt` as the Julia type `key` = name{params}; params = addresses (committed datatypes or parameter datasets)"""
        if key in self.types:
            return self.types[key]
        first = not self.type_order
        if first:
            # the DataType datatype describes itself: its julia_type attribute is an instance of the datatype being committed.  Like
            # JLD2 the object header comes first (address 48, straight after the superblock) and the heap that holds its name after it:
            # reserve the header's bytes, fill them in once the heap objects exist
            a0 = self._addr()
            size = len(self._object_bytes([_msg(0x03, dt, 0x40), _msg(0x0C, bytes(8 + 11 + 10 + 4 + 32))]))
            self.buf += bytes(size)
            self.type_order.append(a0)
            raw = self._object_bytes([_msg(0x03, dt, 0x40), _msg(0x0C, self._type_attr("julia_type", name, params))])
            assert len(raw) == size
            self.buf[self._pos(a0):self._pos(a0) + size] = raw
            self.types[key] = a0
            return a0
        msgs = [_msg(0x03, dt, 0x40), _msg(0x0C, self._type_attr("julia_type", name, params))]
        if written is not None:
            msgs.append(_msg(0x0C, self._type_attr("written_type", written[0], written[1])))
        if field_types is not None:
            an = b"field_datatypes\0"
            sp = _space((len(field_types),))
            msgs.append(_msg(0x0C, bytes([2, 0]) + struct.pack("<HHH", len(an), len(DT_REF), len(sp)) + an + DT_REF + sp
                             + b"".join(struct.pack("<Q", r) for r in field_types)))
        a = self._object(msgs)
        self.type_order.append(a)
        self.types[key] = a
        return a

    def _type_param(self, name):
        """a type without a committed datatype of its own (Core.Float64, Core.Any, ...) as a parameter: a dataset holding a DataType value"""
        if ("T", name) not in self.values:
            self.values[("T", name)] = self._dataset(_shared(self.type_order[0]), True, None, self._datatype_value(name, []))
        return self.values[("T", name)]

    def _int_param(self, n):
        if ("I", n) not in self.values:
            self.values[("I", n)] = self._dataset(_dt_int(8), False, None, struct.pack("<q", n))
        return self.values[("I", n)]

    def _t_symbol(self):
        return self._commit("Core.Symbol", DT_VLEN_STR, "Core.Symbol", [])

    def _t_complex(self, fsize):
        f = "Core.Float32" if fsize == 4 else "Core.Float64"
        dt = _dt_compound([("re", 0, _dt_float(fsize)), ("im", fsize, _dt_float(fsize))], 2 * fsize)
        return self._commit(f"Base.Complex{{{f}}}", dt, "Base.Complex", [self._type_param(f)])

    def _t_pair(self):
        sym, any_ = self._t_symbol(), self._type_param("Core.Any")
        dt = _dt_compound([("first", 0, DT_VLEN_STR), ("second", 16, DT_REF)], 24)
        return self._commit("Base.Pair{Core.Symbol,Core.Any}", dt, "Base.Pair", [sym, any_], field_types=[sym, 0])

    def _t_dict(self):
        sym, any_, pair = self._t_symbol(), self._type_param("Core.Any"), self._t_pair()
        return self._commit("Base.Dict{Core.Symbol,Core.Any}", DT_REF, "Base.Dict", [sym, any_], written=("Core.Array", [pair, self._int_param(1)]))

    # ---- values ---------------------------------------------------------------------------------------------------------------------
    def _value(self, v):
        """-> address of the dataset holding `v`"""
        if isinstance(v, (bool, np.bool_)):
            return self._dataset(DT_BOOL, False, None, bytes([bool(v)]))
        if isinstance(v, (int, np.integer)):
            return self._dataset(_dt_int(8), False, None, struct.pack("<q", int(v)))
        if isinstance(v, (float, np.floating)):
            return self._dataset(_dt_float(8), False, None, struct.pack("<d", float(v)))
        if isinstance(v, Symbol):
            return self._dataset(_shared(self._t_symbol()), True, None, self._vlen(str(v).encode()))
        if isinstance(v, str):
            return self._dataset(DT_VLEN_STR, False, None, self._vlen(v.encode()))
        if isinstance(v, np.ndarray):
            if v.ndim == 0:
                return self._value(v.item())
            a = np.ascontiguousarray(v)
            if a.dtype == np.bool_:
                return self._dataset(DT_BOOL, False, a.shape, a.astype(np.uint8).tobytes())
            if a.dtype.kind == "c":
                return self._dataset(_shared(self._t_complex(a.dtype.itemsize // 2)), True, a.shape, a.tobytes())
            if a.dtype.kind == "i" and a.dtype not in _NP:
                a = a.astype(np.int64)
            if a.dtype.kind == "f" and a.dtype not in _NP:
                a = a.astype(np.float64)
            if a.dtype not in _NP:
                raise TypeError(f"cannot write arrays of {a.dtype}")
            return self._dataset(_NP[a.dtype][1](), False, a.shape, a.astype(a.dtype.newbyteorder("<")).tobytes())
        if isinstance(v, (list, tuple)):
            refs = [self._value(e) for e in v]
            return self._dataset(DT_REF, False, (len(refs),), b"".join(struct.pack("<Q", r) for r in refs))
        if isinstance(v, dict):
            items = [(k, x) for k, x in v.items() if x is not None]
            refs = [self._value(x) for _, x in items]
            raw = b"".join(self._vlen(str(k).encode()) + struct.pack("<Q", r) for (k, _), r in zip(items, refs))
            pairs = self._dataset(_shared(self._t_pair()), True, (len(items),), raw)
            return self._dataset(_shared(self._t_dict()), True, None, struct.pack("<Q", pairs))
        if hasattr(v, "detach"):                                         # a torch tensor
            return self._value(v.detach().cpu().numpy())
        raise TypeError(f"cannot write {type(v)} to a JLD2 file")

    def write(self, name, v):
        if name in self.links:
            raise KeyError(f"{name} exists")                             # as JLD2: datasets are never overwritten
        self.links[name] = self._value(v)

    def __contains__(self, name):
        return name in self.links

    def close(self):
        self._heap_close()
        link = lambda k, a: _msg(0x06, bytes([1, 0x10, 1, len(k.encode())]) + k.encode() + struct.pack("<Q", a))
        ginfo = [_msg(0x02, bytes([0, 0]) + struct.pack("<QQ", UNDEF, UNDEF)), _msg(0x0A, bytes([0, 0]))]
        types = self._object(ginfo + [link(f"{i + 1:08d}", a) for i, a in enumerate(self.type_order)])
        root = self._object(ginfo + [link(k, a) for k, a in self.links.items()] + [link("_types", types)])
        sb = b"\x89HDF\r\n\x1a\n" + bytes([2, 8, 8, 0]) + struct.pack("<QQQQ", BASE, UNDEF, self.off + len(self.buf), root)
        sb += struct.pack("<I", lookup3(sb))
        if self.off == 0:
            self.buf[BASE:BASE + len(sb)] = sb
            tmp = self.path + ".tmp"
            with open(tmp, "wb") as fh:
                fh.write(bytes(self.buf))
            os.replace(tmp, self.path)                                    # a reader never sees a half-written file
        else:
            # append: new objects behind the old end of file, then the superblock switches to the new root group in one 48-byte
            # write (the old root group and `_types` group stay behind as unreferenced bytes, as after JLD2's own "a+")
            with open(self.path, "r+b") as fh:
                fh.seek(self.off)
                fh.write(bytes(self.buf))
                fh.flush()
                os.fsync(fh.fileno())
                fh.seek(BASE)
                fh.write(sb)

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
