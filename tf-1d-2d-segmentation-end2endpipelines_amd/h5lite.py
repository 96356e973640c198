"""h5lite - the part of the HDF5 file format that Keras checkpoints use, in plain Python + numpy (no libhdf5, no h5py).

Why it exists: the reference checkpoints `tf.keras` models (`2D/Train.py:375-379`,
`ModelCheckpoint(... .keras)`; `model.load_weights` to resume) - a `.keras` archive holds `model.weights.h5`, the older formats are
HDF5 files outright - and neither TensorFlow nor h5py can be installed next to this engine.  A user who switches over brings such
files along; `keras_file.py` maps them onto the structural parameter names, this module gets the bytes out (and writes them back).

Reader (`H5File`): superblock versions 0 - 3; object headers version 1 and 2 (continuation blocks); groups as symbol tables
(B-tree v1 + local heap + symbol nodes: what libhdf5 writes unless `libver='latest'`) or as compact link messages; datasets in
compact, contiguous and chunked (B-tree v1; single-chunk and implicit index of layout version 4) layout with the deflate / shuffle /
fletcher32 filters; fixed-point, floating-point, fixed-length string, variable-length string (global heap), enum (h5py booleans)
datatypes, either byte order; attributes (message versions 1 - 3) stored in the object header.  NOT read, and refused with an
H5Error that says so rather than returning wrong data: dense link / attribute storage (fractal heaps: groups of more than eight
links written with `libver='latest'` or `track_order=True`), the extensible / fixed array and B-tree v2 chunk indexes, compound /
reference / array datatypes, shared (committed) datatype messages, external files, virtual datasets.

Writer (`H5Writer`): superblock version 0, version-1 object headers, symbol-table groups, contiguous little-endian datasets,
numeric / fixed-length string attributes, long text attributes as variable-length strings in a global heap collection - enough for a
weights file that libhdf5 (h5py, Keras) opens.

Verified against files written by the real library (h5py 3.3 / HDF5 1.10.6, `tests/golden/make_h5_files.py`) and, the other way,
by letting that library read what `H5Writer` wrote (`tests/test_h5lite.py`).  Format facts follow the published "HDF5 File Format
Specification Version 3.0"; nothing here is derived from libhdf5 source."""
from __future__ import annotations

import mmap
import os
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5Error(ValueError):
    pass


def _guard(fn):
    """A damaged file (a truncated download, a flipped bit in an offset) must surface as H5Error, whatever the parser tripped over."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except (H5Error, KeyError):
            raise
        except (struct.error, IndexError, TypeError, ValueError, OverflowError, UnicodeDecodeError, zlib.error, MemoryError, RecursionError) as e:
            raise H5Error(f"corrupt or truncated HDF5 file ({type(e).__name__}: {str(e)[:160]})") from None
    return wrapped


# ------------------------------------------------------------------------------------------------------------------ datatypes
class _Type:
    """A parsed datatype message: `kind` in {"num", "str", "vstr", "vlen"}; `dtype` the numpy dtype of one stored element."""

    def __init__(self, kind: str, dtype: np.dtype, size: int, base: Optional["_Type"] = None, utf8: bool = False):
        self.kind, self.dtype, self.size, self.base, self.utf8 = kind, dtype, size, base, utf8


def _parse_type(d: bytes, O: int) -> _Type:
    cls, ver = d[0] & 0x0F, d[0] >> 4
    b0 = d[1]
    size = struct.unpack_from("<I", d, 4)[0]
    if not 1 <= ver <= 4:
        raise H5Error(f"datatype message version {ver}")
    if cls == 0:                                        # fixed point
        order = ">" if b0 & 1 else "<"
        return _Type("num", np.dtype(f"{order}{'i' if b0 & 0x08 else 'u'}{size}"), size)
    if cls == 1:                                        # floating point (IEEE layouts only: 2, 4, 8 bytes)
        if size not in (2, 4, 8):
            raise H5Error(f"floating-point type of {size} bytes")
        return _Type("num", np.dtype(f"{'>' if b0 & 1 else '<'}f{size}"), size)
    if cls == 3:                                        # fixed-length string
        return _Type("str", np.dtype(f"S{size}"), size, utf8=bool((b0 >> 4) & 1))
    if cls == 9:                                        # variable length: 4-byte length + global heap address + 4-byte index
        base = _parse_type(d[8:], O)
        is_str = (b0 & 0x0F) == 1
        return _Type("vstr" if is_str else "vlen", np.dtype(f"V{size}"), size, base, utf8=bool(d[2] & 1))
    if cls == 8:                                        # enum (h5py stores numpy booleans this way): the base type's values
        return _parse_type(d[8:], O)
    names = {2: "time", 4: "bitfield", 5: "opaque", 6: "compound", 7: "reference", 10: "array"}
    raise H5Error(f"datatype class {cls} ({names.get(cls, '?')}) is outside the subset h5lite reads")


# ------------------------------------------------------------------------------------------------------------------ the file
class H5File:
    """Read-only view of an HDF5 file: `H5File(path)` (memory-mapped), `H5File(bytes)` or `H5File(file-like)`.  Behaves like the root
    group: `f["layers/conv2d/vars/0"].read()`, `f.keys()`, `f.attrs`, `"x" in f`, `f.visit_datasets()`."""

    @_guard
    def __init__(self, src: Union[str, os.PathLike, bytes, bytearray, memoryview, object]):
        self._fh = self._mm = None
        if isinstance(src, (str, os.PathLike)):
            self._fh = open(src, "rb")
            try:
                self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
            except ValueError as e:                     # empty file
                self._fh.close()
                raise H5Error(f"{src}: not an HDF5 file ({e})") from None
            self.buf = memoryview(self._mm)
        elif isinstance(src, (bytes, bytearray, memoryview)):
            self.buf = memoryview(src)
        else:
            self.buf = memoryview(src.read())
        self._gheaps: Dict[int, Dict[int, bytes]] = {}
        self._superblock()
        self.root = Group(self, self._root_addr, "/")

    # -- context manager / group protocol, delegated to the root group
    def close(self):
        self.buf.release() if hasattr(self.buf, "release") else None
        if self._mm is not None:
            try:
                self._mm.close()
            except BufferError:                         # arrays handed out still view the map: it goes with them
                pass
            self._fh.close()
            self._mm = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __getitem__(self, k):
        return self.root[k]

    def __contains__(self, k):
        return k in self.root

    def keys(self):
        return self.root.keys()

    def items(self):
        return self.root.items()

    @property
    def attrs(self):
        return self.root.attrs

    def visit_datasets(self):
        return self.root.visit_datasets()

    # -- primitives
    def _u(self, off: int, n: int) -> int:
        if off < 0 or off + n > len(self.buf):
            raise H5Error(f"read of {n} bytes at offset {off} past the end of the file ({len(self.buf)} bytes): truncated or corrupt")
        return int.from_bytes(self.buf[off:off + n], "little")

    def _bytes(self, off: int, n: int) -> bytes:
        if off < 0 or n < 0 or off + n > len(self.buf):
            raise H5Error(f"read of {n} bytes at offset {off} past the end of the file ({len(self.buf)} bytes): truncated or corrupt")
        return bytes(self.buf[off:off + n])

    def _addr(self, off: int) -> int:
        a = self._u(off, self.O)
        return -1 if a == (1 << (8 * self.O)) - 1 else a + self.base

    def _superblock(self):
        off = 0
        while True:
            if off + 8 > len(self.buf):
                raise H5Error("no HDF5 signature: not an HDF5 file")
            if bytes(self.buf[off:off + 8]) == SIGNATURE:
                break
            off = 512 if off == 0 else off * 2
        ver = self._u(off + 8, 1)
        self.base = 0
        if ver in (0, 1):
            self.O, self.L = self._u(off + 13, 1), self._u(off + 14, 1)
            p = off + 24 + (4 if ver == 1 else 0)
            base = self._u(p, self.O)
            self.base = base
            p += 4 * self.O                             # base, free-space info, end of file, driver info
            self._root_addr = self._addr(p + self.O)    # root symbol table entry: link name offset, object header address
        elif ver in (2, 3):
            self.O, self.L = self._u(off + 9, 1), self._u(off + 10, 1)
            p = off + 12
            self.base = self._u(p, self.O)
            self._root_addr = self._addr(p + 3 * self.O)
        else:
            raise H5Error(f"superblock version {ver}")
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise H5Error(f"offset / length sizes {self.O} / {self.L}")

    # -- object headers
    def _messages(self, addr: int) -> List[Tuple[int, int, int, int]]:
        """[(type, flags, data offset, data size)] of the object header at `addr`, continuation blocks followed."""
        out = []
        if self._bytes(addr, 4) == b"OHDR":
            if self._u(addr + 4, 1) != 2:
                raise H5Error("object header version")
            hflags = self._u(addr + 5, 1)
            p = addr + 6 + (16 if hflags & 0x20 else 0) + (4 if hflags & 0x10 else 0)
            w = 1 << (hflags & 3)
            size0 = self._u(p, w)
            blocks = [(p + w, size0)]
            extra = 2 if hflags & 0x04 else 0
            nblocks = 0
            while blocks:
                p, n = blocks.pop(0)
                nblocks += 1
                if nblocks > 4096:
                    raise H5Error("object header continuation blocks form a cycle")
                end = p + n
                while p + 4 + extra <= end:
                    t, sz, fl = self._u(p, 1), self._u(p + 1, 2), self._u(p + 3, 1)
                    d = p + 4 + extra
                    if d + sz > end:
                        break                           # (a gap too small for a message)
                    if t == 0x10:
                        ca, cl = self._addr(d), self._u(d + self.O, self.L)
                        if self._bytes(ca, 4) != b"OCHK":
                            raise H5Error("object header continuation block without its signature")
                        blocks.append((ca + 4, cl - 8))          # (signature in front, checksum behind)
                    elif t != 0:
                        out.append((t, fl, d, sz))
                    p = d + sz
            return out
        if self._u(addr, 1) != 1:
            raise H5Error(f"no object header at offset {addr}")
        nmsg = self._u(addr + 2, 2)
        blocks = [(addr + 16, self._u(addr + 8, 4))]
        seen = 0
        nblocks = 0
        while blocks and seen < nmsg:
            p, n = blocks.pop(0)
            nblocks += 1
            if nblocks > 4096:
                raise H5Error("object header continuation blocks form a cycle")
            end = p + n
            while p + 8 <= end and seen < nmsg:
                t, sz, fl = self._u(p, 2), self._u(p + 2, 2), self._u(p + 4, 1)
                d = p + 8
                seen += 1
                if t == 0x10:
                    blocks.append((self._addr(d), self._u(d + self.O, self.L)))
                elif t != 0:
                    out.append((t, fl, d, sz))
                p = d + sz
        return out

    # -- global heap (variable-length data)
    def _gheap_object(self, addr: int, index: int) -> bytes:
        col = self._gheaps.get(addr)
        if col is None:
            if self._bytes(addr, 4) != b"GCOL":
                raise H5Error("global heap collection without its signature")
            size = self._u(addr + 8, self.L)
            col, p, end = {}, addr + 8 + self.L, addr + size
            while p + 8 + self.L <= end:
                idx = self._u(p, 2)
                n = self._u(p + 8, self.L)
                if idx == 0:
                    break
                col[idx] = (p + 8 + self.L, n)
                p += 8 + self.L + ((n + 7) & ~7)
            self._gheaps[addr] = col
        if index not in col:
            raise H5Error(f"global heap object {index} missing from the collection at {addr}")
        o, n = col[index]
        return self._bytes(o, n)

    # -- raw element bytes -> python / numpy value
    def _decode(self, raw, t: _Type, shape: Tuple[int, ...]):
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if t.kind in ("num", "str"):
            a = np.frombuffer(raw, dtype=t.dtype, count=count).reshape(shape)
            if t.kind == "num" and not a.dtype.isnative:
                a = a.astype(a.dtype.newbyteorder("="))
            return a[()] if not shape else a.copy()
        out = []
        step = 8 + self.O
        for i in range(count):
            n = int.from_bytes(raw[i * step:i * step + 4], "little")
            ga = int.from_bytes(raw[i * step + 4:i * step + 4 + self.O], "little")
            idx = int.from_bytes(raw[i * step + 4 + self.O:i * step + step], "little")
            data = b"" if (n == 0 or ga == 0) else self._gheap_object(ga + self.base, idx)
            if t.kind == "vstr":
                out.append(data[:n].decode("utf-8", "replace"))
            else:
                out.append(np.frombuffer(data, dtype=t.base.dtype, count=n).copy())
        if not shape:
            return out[0]
        a = np.empty(count, dtype=object)
        a[:] = out
        return a.reshape(shape)


def _parse_space(d: bytes, L: int) -> Optional[Tuple[int, ...]]:
    """Dataspace message -> shape (() for a scalar, None for a null dataspace)."""
    ver, rank = d[0], d[1]
    if ver == 1:
        p = 8
    elif ver == 2:
        if d[3] == 2:
            return None
        p = 4
    else:
        raise H5Error(f"dataspace message version {ver}")
    return tuple(int.from_bytes(d[p + i * L:p + (i + 1) * L], "little") for i in range(rank))


class _Node:
    def __init__(self, f: H5File, addr: int, name: str):
        self.file, self.addr, self.name = f, addr, name
        self._msgs = f._messages(addr)
        self._attrs: Optional[Dict[str, object]] = None

    def _first(self, t: int) -> Optional[Tuple[int, int, int]]:
        for mt, fl, d, sz in self._msgs:
            if mt == t:
                return fl, d, sz
        return None

    @property
    @_guard
    def attrs(self) -> Dict[str, object]:
        if self._attrs is None:
            f, out = self.file, {}
            for mt, fl, d, sz in self._msgs:
                if mt == 0x15:                          # attribute info: dense storage when the heap address is defined
                    flags = f._u(d + 1, 1)
                    p = d + 2 + (2 if flags & 1 else 0)
                    if f._addr(p) != -1:
                        raise H5Error(f"{self.name}: attributes in dense storage (fractal heap) are outside the subset h5lite reads")
                if mt != 0x0C:
                    continue
                if fl & 0x02:
                    raise H5Error(f"{self.name}: shared attribute messages are outside the subset h5lite reads")
                m = f._bytes(d, sz)
                ver = m[0]
                nlen, tlen, slen = struct.unpack_from("<HHH", m, 2)
                if ver == 1:
                    pad = lambda n: (n + 7) & ~7
                    p = 8
                elif ver in (2, 3):
                    if m[1] & 0x03:
                        raise H5Error(f"{self.name}: attribute with a shared datatype / dataspace")
                    pad = lambda n: n
                    p = 8 + (1 if ver == 3 else 0)
                else:
                    raise H5Error(f"attribute message version {ver}")
                name = m[p:p + nlen].split(b"\0", 1)[0].decode("utf-8")
                p += pad(nlen)
                t = _parse_type(m[p:p + tlen], f.O)
                p += pad(tlen)
                shape = _parse_space(m[p:p + slen], f.L)
                p += pad(slen)
                out[name] = None if shape is None else f._decode(m[p:], t, shape)
            self._attrs = out
        return self._attrs


class Dataset(_Node):
    @_guard
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        if self._first(0x03) is None or self._first(0x01) is None or self._first(0x08) is None:
            raise H5Error(f"{name}: a dataset without its datatype / dataspace / layout message")
        fl, d, sz = self._first(0x03)
        if fl & 0x02:
            raise H5Error(f"{name}: committed (shared) datatypes are outside the subset h5lite reads")
        self._type = _parse_type(f._bytes(d, sz), f.O)
        _fl, d, sz = self._first(0x01)
        self.shape = _parse_space(f._bytes(d, sz), f.L)
        self.dtype = self._type.dtype.newbyteorder("=") if self._type.kind == "num" else (np.dtype(object) if self._type.kind in ("vstr", "vlen") else self._type.dtype)

    def __getitem__(self, key):
        a = self.read()
        return a if key == () or key is Ellipsis else a[key]

    @_guard
    def read(self):
        """The whole dataset as a numpy array (a scalar dataspace gives a numpy scalar / str)."""
        f, t = self.file, self._type
        if self.shape is None:
            return None
        count = 1
        for s_ in self.shape:
            count *= int(s_)
        if count * t.size > 1100 * len(f.buf) + (1 << 20):          # (deflate expands at most ~1030 x: no file of this size holds that)
            raise H5Error(f"{self.name}: shape {self.shape} x {t.size} bytes is more than this {len(f.buf)}-byte file can hold: corrupt dataspace")
        _fl, d, sz = self._first(0x08)
        ver = f._u(d, 1)
        if ver in (1, 2):
            ndim, cls = f._u(d + 1, 1), f._u(d + 2, 1)
            p = d + 8
            addr = -1
            if cls != 0:
                addr = f._addr(p)
                p += f.O
            dims = [f._u(p + 4 * i, 4) for i in range(ndim)]
            p += 4 * ndim
            if cls == 0:
                return f._decode(f._bytes(p + 4, f._u(p, 4)), t, self.shape)
            if cls == 1:
                return self._contiguous(addr, count)
            return self._chunked_btree(addr, dims[:-1])
        if ver not in (3, 4):
            raise H5Error(f"{self.name}: data layout message version {ver}")
        cls = f._u(d + 1, 1)
        if cls == 0:
            n = f._u(d + 2, 2)
            return f._decode(f._bytes(d + 4, n), t, self.shape)
        if cls == 1:
            return self._contiguous(f._addr(d + 2), count)
        if cls == 2 and ver == 3:
            ndim = f._u(d + 2, 1)
            addr = f._addr(d + 3)
            dims = [f._u(d + 3 + f.O + 4 * i, 4) for i in range(ndim)]
            return self._chunked_btree(addr, dims[:-1])
        if cls == 2:
            flags, ndim, enc = f._u(d + 2, 1), f._u(d + 3, 1), f._u(d + 4, 1)
            dims = [f._u(d + 5 + enc * i, enc) for i in range(ndim)]
            p = d + 5 + enc * ndim
            index = f._u(p, 1)
            p += 1
            if index == 1:                              # single chunk
                csize, mask = None, 0
                if flags & 0x02:
                    csize, mask = f._u(p, f.L), f._u(p + f.L, 4)
                    p += f.L + 4
                addr = f._addr(p)
                out = np.zeros(self.shape, dtype=t.dtype)
                if addr != -1:
                    n = csize if csize is not None else int(np.prod(dims[:-1], dtype=np.int64)) * t.size
                    self._place(out, f._bytes(addr, n), mask, [0] * len(self.shape), dims[:-1])
                return self._finish(out)
            if index == 2:                              # implicit: every chunk allocated, in order, unfiltered
                addr = f._addr(p)
                cd = dims[:-1]
                out = np.zeros(self.shape, dtype=t.dtype)
                n = int(np.prod(cd, dtype=np.int64)) * t.size
                grid = [-(-s // c) for s, c in zip(self.shape, cd)]
                for i, idx in enumerate(np.ndindex(*grid)):
                    self._place(out, f._bytes(addr + i * n, n), 0, [a * c for a, c in zip(idx, cd)], cd, filters=())
                return self._finish(out)
            names = {3: "fixed array", 4: "extensible array", 5: "B-tree v2"}
            raise H5Error(f"{self.name}: chunk index type {index} ({names.get(index, '?')}) is outside the subset h5lite reads")
        raise H5Error(f"{self.name}: data layout class {cls} (virtual / external storage) is outside the subset h5lite reads")

    def _contiguous(self, addr, count):
        f, t = self.file, self._type
        if addr == -1:                                  # never written: the fill value (zero / empty)
            if t.kind in ("vstr", "vlen"):
                raise H5Error(f"{self.name}: unallocated variable-length dataset")
            return np.zeros(self.shape, dtype=self.dtype)
        n = count * t.size
        if addr + n > len(f.buf):
            raise H5Error(f"{self.name}: data at {addr} + {n} bytes runs past the end of the file ({len(f.buf)} bytes): truncated file")
        return f._decode(f.buf[addr:addr + n], t, self.shape)

    def _filters(self):
        m = self._first(0x0B)
        if m is None:
            return []
        f = self.file
        _fl, d, _sz = m
        ver, n = f._u(d, 1), f._u(d + 1, 1)
        p = d + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = f._u(p, 2)
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = f._u(p, 2)
                p += 2
            ncd = f._u(p + 2, 2)
            p += 4
            p += ((nlen + 7) & ~7) if ver == 1 else nlen
            cd = [f._u(p + 4 * i, 4) for i in range(ncd)]
            p += 4 * ncd + (4 if (ver == 1 and ncd & 1) else 0)
            out.append((fid, cd))
        return out

    def _place(self, out, raw, mask, offset, cdims, filters=None):
        t = self._type
        filters = self._filters() if filters is None else filters
        for i in range(len(filters) - 1, -1, -1):
            if mask & (1 << i):
                continue
            fid, cd = filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cd[0] if cd else t.size
                n = len(raw) // es
                raw = np.frombuffer(raw[:n * es], np.uint8).reshape(es, n).T.tobytes() + bytes(raw[n * es:])
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise H5Error(f"{self.name}: filter {fid} is outside the subset h5lite reads (deflate, shuffle, fletcher32)")
        chunk = np.frombuffer(raw, dtype=t.dtype, count=int(np.prod(cdims, dtype=np.int64))).reshape(cdims)
        sel_o, sel_c = [], []
        for o, c, s in zip(offset, cdims, self.shape):
            n = min(c, s - o)
            if n <= 0:
                return
            sel_o.append(slice(o, o + n))
            sel_c.append(slice(0, n))
        out[tuple(sel_o)] = chunk[tuple(sel_c)]

    def _finish(self, out):
        if not out.dtype.isnative:
            out = out.astype(out.dtype.newbyteorder("="))
        return out

    def _chunked_btree(self, addr, cdims):
        f, t = self.file, self._type
        if t.kind not in ("num", "str"):
            raise H5Error(f"{self.name}: chunked variable-length data is outside the subset h5lite reads")
        out = np.zeros(self.shape, dtype=t.dtype)
        if addr == -1:
            return self._finish(out)
        nd = len(cdims) + 1
        filters = self._filters()

        def walk(a, above=256):
            if f._bytes(a, 4) != b"TREE" or f._u(a + 4, 1) != 1:
                raise H5Error(f"{self.name}: chunk B-tree node expected at {a}")
            level, n = f._u(a + 5, 1), f._u(a + 6, 2)
            if level >= above:
                raise H5Error(f"{self.name}: B-tree levels do not fall towards the leaves (a cycle)")
            p = a + 8 + 2 * f.O
            ksz = 8 + 8 * nd
            for _ in range(n):
                size, mask = f._u(p, 4), f._u(p + 4, 4)
                off = [f._u(p + 8 + 8 * i, 8) for i in range(nd - 1)]
                child = f._addr(p + ksz)
                p += ksz + f.O
                if level:
                    walk(child, level)
                else:
                    self._place(out, f._bytes(child, size), mask, off, cdims, filters)

        walk(addr)
        return self._finish(out)


class Group(_Node):
    @_guard
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        self._links: Optional[Dict[str, int]] = None

    @_guard
    def _load(self) -> Dict[str, int]:
        if self._links is not None:
            return self._links
        f, links = self.file, {}
        st = self._first(0x11)
        if st is not None:
            _fl, d, _sz = st
            btree, heap = f._addr(d), f._addr(d + f.O)
            if f._bytes(heap, 4) != b"HEAP":
                raise H5Error(f"{self.name}: local heap expected at {heap}")
            hdata = f._addr(heap + 8 + 2 * f.L)
            hsize = f._u(heap + 8, f.L)

            def name_at(off):
                raw = f._bytes(hdata + off, min(hsize - off, 4096))
                while b"\0" not in raw and len(raw) < hsize - off:
                    raw = f._bytes(hdata + off, min(hsize - off, 2 * len(raw)))
                return raw.split(b"\0", 1)[0].decode("utf-8")

            def walk(a, above=256):
                if f._bytes(a, 4) != b"TREE" or f._u(a + 4, 1) != 0:
                    raise H5Error(f"{self.name}: group B-tree node expected at {a}")
                level, n = f._u(a + 5, 1), f._u(a + 6, 2)
                if level >= above:
                    raise H5Error(f"{self.name}: B-tree levels do not fall towards the leaves (a cycle)")
                p = a + 8 + 2 * f.O + f.L
                for _ in range(n):
                    child = f._addr(p)
                    p += f.O + f.L
                    if level:
                        walk(child, level)
                        continue
                    if f._bytes(child, 4) != b"SNOD":
                        raise H5Error(f"{self.name}: symbol table node expected at {child}")
                    q = child + 8
                    for _s in range(f._u(child + 6, 2)):
                        if f._u(q + 2 * f.O, 4) != 2:              # (cache type 2: a soft link - not followed)
                            links[name_at(f._u(q, f.O))] = f._addr(q + f.O)
                        q += 2 * f.O + 24

            if btree != -1:
                walk(btree)
        else:
            for mt, fl, d, sz in self._msgs:
                if mt == 0x02:                          # link info: dense storage when the fractal heap address is defined
                    flags = f._u(d + 1, 1)
                    p = d + 2 + (8 if flags & 1 else 0)
                    if f._addr(p) != -1:
                        raise H5Error(f"{self.name}: links in dense storage (fractal heap: libver='latest' / track_order groups of "
                                      "more than eight links) are outside the subset h5lite reads")
                if mt != 0x06:
                    continue
                flags = f._u(d + 1, 1)
                p = d + 2
                ltype = 0
                if flags & 0x08:
                    ltype = f._u(p, 1)
                    p += 1
                if flags & 0x04:
                    p += 8
                if flags & 0x10:
                    p += 1
                w = 1 << (flags & 3)
                n = f._u(p, w)
                p += w
                nm = f._bytes(p, n).decode("utf-8")
                p += n
                if ltype == 0:
                    links[nm] = f._addr(p)
        if any(not k or "/" in k for k in links):
            raise H5Error(f"{self.name}: a link with an empty name or a '/' in it: corrupt group")
        self._links = links
        return links

    def keys(self) -> List[str]:
        return sorted(self._load())

    def __contains__(self, key: str) -> bool:
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __iter__(self) -> Iterator[str]:
        return iter(self.keys())

    def __len__(self):
        return len(self._load())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    @_guard
    def __getitem__(self, key: str):
        node = self
        if node.name.count("/") > 64:
            raise H5Error(f"{node.name[:80]}...: groups nested more than 64 deep (a link that points back up the tree)")
        for part in [p for p in key.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(f"{key!r}: {node.name!r} is a dataset")
            links = node._load()
            if part not in links:
                raise KeyError(f"{key!r}: no {part!r} in {node.name!r}")
            cache = node.__dict__.setdefault("_children", {})
            if part not in cache:
                addr = links[part]
                path = (node.name.rstrip("/") + "/" + part)
                kinds = {m[0] for m in node.file._messages(addr)}
                cache[part] = Dataset(node.file, addr, path) if 0x08 in kinds else Group(node.file, addr, path)
            node = cache[part]
        return node

    def visit_datasets(self) -> List[Tuple[str, Dataset]]:
        """[(path relative to this group, dataset)] of everything below, depth first in name order."""
        out = []

        def rec(g, prefix):
            for k in g.keys():
                x = g[k]
                if isinstance(x, Group):
                    rec(x, prefix + k + "/")
                else:
                    out.append((prefix + k, x))

        rec(self, "")
        return out


# ------------------------------------------------------------------------------------------------------------------ the writer
def _c_order(x) -> np.ndarray:
    a = np.asarray(x)
    return a if a.ndim == 0 else np.ascontiguousarray(a)          # (ascontiguousarray alone makes a scalar one-dimensional)


class H5Writer:
    """Writes a tree of groups, numeric datasets and small attributes as an HDF5 file that libhdf5 reads.

        w = H5Writer()
        w.create_dataset("layers/conv2d/vars/0", kernel)          # intermediate groups are created
        w.create_group("layers/activation/vars")                  # (an empty group)
        w.set_attr("/", "keras_version", "2.13.1")                # str -> fixed-length string, numbers / arrays as they are,
        w.set_attr("/", "layer_names", [b"conv2d", b"bn"])        # a list of bytes -> an array of fixed-length strings
        w.save(path)            # or w.tobytes()

    Layout: superblock version 0 with 8-byte offsets; every group a symbol table (one local heap, ONE B-tree leaf level whose symbol
    nodes hold up to 2 x 512 entries - the group leaf K the superblock declares -, 32 such nodes per group: 32 768 entries);
    datasets contiguous, little-endian, 8-byte aligned; attributes as version-1 messages in the object header (64 KB per message)."""

    LEAF_K, INTERNAL_K = 512, 16

    def __init__(self):
        self._root: Dict = {"kind": "group", "children": {}, "attrs": {}}

    def _find(self, path: str, create: bool):
        node = self._root
        for part in [p for p in path.split("/") if p]:
            if node["kind"] != "group":
                raise H5Error(f"{path!r}: a dataset is in the way")
            ch = node["children"]
            if part not in ch:
                if not create:
                    raise KeyError(path)
                ch[part] = {"kind": "group", "children": {}, "attrs": {}}
            node = ch[part]
        return node

    def create_group(self, path: str):
        self._find(path, True)

    def create_dataset(self, path: str, data):
        a = _c_order(data)
        if a.dtype.kind not in "fiub":
            raise H5Error(f"{path!r}: dtype {a.dtype} (numeric arrays only)")
        if a.dtype.kind == "b":
            a = a.astype(np.uint8)
        parts = [p for p in path.split("/") if p]
        parent = self._find("/".join(parts[:-1]), True)
        if parent["kind"] != "group":
            raise H5Error(f"{path!r}: a dataset is in the way")
        if parts[-1] in parent["children"]:
            raise H5Error(f"{path!r} exists")
        parent["children"][parts[-1]] = {"kind": "dataset", "data": a.astype(a.dtype.newbyteorder("<")), "attrs": {}}

    def set_attr(self, path: str, name: str, value):
        self._find(path, False)["attrs"][name] = value

    # -- message encoders
    @staticmethod
    def _type_msg(dt: np.dtype) -> bytes:
        if dt.kind == "S":
            return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, dt.itemsize)                       # null-terminated ASCII
        if dt.kind == "f":
            sz = dt.itemsize
            exp_loc, exp_sz, man_sz, bias = {2: (10, 5, 10, 15), 4: (23, 8, 23, 127), 8: (52, 11, 52, 1023)}[sz]
            return (struct.pack("<BBBBI", 0x11, 0x20, 8 * sz - 1, 0, sz)                       # LE, implied-one mantissa, sign position
                    + struct.pack("<HHBBBBI", 0, 8 * sz, exp_loc, exp_sz, 0, man_sz, bias))
        if dt.kind in "iu":
            return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
        raise H5Error(f"dtype {dt}")

    @staticmethod
    def _space_msg(shape) -> bytes:
        return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(struct.pack("<Q", s) for s in shape)

    @classmethod
    def _attr_msg(cls, name: str, value, alloc=None) -> bytes:
        pad = lambda b: b + b"\0" * (-len(b) % 8)
        nm = name.encode("utf-8") + b"\0"
        if isinstance(value, str):
            value = value.encode("utf-8")
        if isinstance(value, bytes) and len(value) > 60000 and alloc is not None:
            # more than an object-header message holds (a legacy file's model_config): a variable-length string - the text in a
            # global heap collection of its own (exactly filled: header, object 1, no free-space object), the attribute a reference
            n = (len(value) + 7) & ~7
            gaddr = alloc(b"GCOL" + struct.pack("<B3xQ", 1, 32 + n) + struct.pack("<HHIQ", 1, 0, 0, len(value)) + value + b"\0" * (n - len(value)))
            t = struct.pack("<BBBBI", 0x19, 0x01, 0x01, 0, 16) + struct.pack("<BBBBI", 0x10, 0, 0, 0, 1) + struct.pack("<HH", 0, 8)
            sp = struct.pack("<BBBB4x", 1, 0, 0, 0)
            return (struct.pack("<BBHHH", 1, 0, len(nm), len(t), len(sp)) + pad(nm) + pad(t) + pad(sp)
                    + struct.pack("<IQI", len(value), gaddr, 1))
        if isinstance(value, bytes):
            a = np.array(value + b"\0", dtype=f"S{len(value) + 1}")
        elif isinstance(value, (list, tuple)) and value and isinstance(value[0], (bytes, str)):
            vals = [v.encode("utf-8") if isinstance(v, str) else v for v in value]
            a = np.array(vals, dtype=f"S{max(len(v) for v in vals) + 1}")
        else:
            a = _c_order(value)
            if a.dtype.kind not in "fiuS":
                raise H5Error(f"attribute {name!r}: dtype {a.dtype}")
            if a.dtype.kind != "S":
                a = a.astype(a.dtype.newbyteorder("<"))
        t, s = cls._type_msg(a.dtype), cls._space_msg(a.shape)
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(t), len(s)) + pad(nm) + pad(t) + pad(s) + a.tobytes()
        if len(body) > 65000:
            raise H5Error(f"attribute {name!r}: {len(body)} bytes (an object-header message holds 64 KB; split the list as Keras does)")
        return body

    @staticmethod
    def _header(msgs: List[Tuple[int, bytes]]) -> bytes:
        body = b""
        for t, m in msgs:
            m = m + b"\0" * (-len(m) % 8)
            body += struct.pack("<HHB3x", t, len(m), 0) + m
        return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body

    def tobytes(self) -> bytes:
        out = bytearray(b"\0" * 96)                      # the superblock (56 bytes + the 40-byte root symbol table entry)

        def alloc(b: bytes) -> int:
            out.extend(b"\0" * (-len(out) % 8))
            a = len(out)
            out.extend(b)
            return a

        def emit(node) -> Tuple[int, Optional[Tuple[int, int]]]:
            attrs = [(0x0C, self._attr_msg(k, v, alloc)) for k, v in node["attrs"].items()]
            if node["kind"] == "dataset":
                a = node["data"]
                daddr = alloc(a.tobytes()) if a.size else 0xFFFFFFFFFFFFFFFF
                msgs = [(0x01, self._space_msg(a.shape)), (0x03, self._type_msg(a.dtype)),
                        (0x05, struct.pack("<BBBBI", 2, 2, 2, 1, 0)),                  # fill value as libhdf5 writes it: version 2, allocate late, write if set, default
                        (0x08, struct.pack("<BBQQ", 3, 1, daddr, a.nbytes))] + attrs
                return alloc(self._header(msgs)), None
            names = sorted(node["children"], key=lambda s: s.encode("utf-8"))
            if len(names) > 2 * self.LEAF_K * 2 * self.INTERNAL_K:
                raise H5Error(f"a group of {len(names)} entries (H5Writer holds {2 * self.LEAF_K * 2 * self.INTERNAL_K})")
            entries = [(nm,) + emit(node["children"][nm]) for nm in names]
            heap = bytearray(b"\0" * 8)                                             # offset 0: the empty name
            offs = []
            for nm in names:
                offs.append(len(heap))
                b = nm.encode("utf-8") + b"\0"
                heap.extend(b + b"\0" * (-len(b) % 8))
            free = len(heap)                                                        # one free block closes the heap (next = 1: end of list)
            heap.extend(struct.pack("<QQ", 1, 32) + b"\0" * 16)
            hdata = alloc(bytes(heap))
            haddr = alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free, hdata))
            per = 2 * self.LEAF_K
            snods, keys = [], [0]
            for s in range(0, len(entries), per):
                part = entries[s:s + per]
                body = b"SNOD" + struct.pack("<BBH", 1, 0, len(part))
                for (nm, oaddr, st), off in zip(part, offs[s:s + per]):
                    if st is not None:
                        body += struct.pack("<QQII", off, oaddr, 1, 0) + struct.pack("<QQ", st[0], st[1])
                    else:
                        body += struct.pack("<QQII16x", off, oaddr, 0, 0)
                body += b"\0" * ((per - len(part)) * 40)
                snods.append(alloc(body))
                keys.append(offs[s + len(part) - 1])
            node_body = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF)
            node_body += struct.pack("<Q", keys[0])
            for a, k in zip(snods, keys[1:]):
                node_body += struct.pack("<QQ", a, k)
            node_body += b"\0" * ((2 * self.INTERNAL_K - len(snods)) * 16)
            baddr = alloc(node_body)
            return alloc(self._header([(0x11, struct.pack("<QQ", baddr, haddr))] + attrs)), (baddr, haddr)

        raddr, rst = emit(self._root)
        out.extend(b"\0" * (-len(out) % 8))
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, 0xFFFFFFFFFFFFFFFF, len(out), 0xFFFFFFFFFFFFFFFF)
        sb += struct.pack("<QQII", 0, raddr, 1, 0) + struct.pack("<QQ", rst[0], rst[1])
        out[:len(sb)] = sb
        return bytes(out)

    def save(self, path: str):
        with open(path, "wb") as fh:
            fh.write(self.tobytes())
