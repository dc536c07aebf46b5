"""Dependency-free reader (and test-data writer) for the ONE HDF5 layout this project consumes: the files written by the
reference's `external/dataset_tool_h5.py:104-111` with h5py defaults --

    /shapes   int32   [N, 3]   contiguous          (3, h, w) per image
    /images   vlen<uint8> [N]  contiguous          raw CHW bytes of image i, in global-heap collections

h5py / libhdf5 are not installed in the build image, so this module parses the file format directly
(HDF5 File Format Specification 3.0: version-0 superblock, version-1 object headers, symbol-table groups with a v1 B-tree +
local heap, layout message v3 contiguous, variable-length sequences as {length, global-heap address, index} descriptors,
"GCOL" global heap collections).  Scope: exactly that subset; anything else raises `H5LiteError` loudly.

VALIDATION STATUS: pinned against a file written by the real library -- tests/golden/g_libhdf5_dataset.h5 comes from libhdf5
1.10 (oracle/h5gen/make_fixture.c: the reference tool's layout, one element per write like h5py's `dset[idx] = ...`) and
tests/test_data_layer.py::test_h5lite_reads_a_libhdf5_written_file reads it back bit for bit; the reader is also exercised
against files produced by `write_dataset_file` below (same subset, written from the specification).  Not covered: chunked or
compressed datasets, new-style (v2) groups -- none of which the reference's converter produces.
"""
import os
import struct
from typing import List, Tuple

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"


class H5LiteError(RuntimeError):
    pass


class _File:
    """Positioned reads only (`os.pread`): the descriptor carries no file offset that forked DataLoader workers could race on
    (a seek()+read() pair on an inherited descriptor did: corrupted images)."""

    def __init__(self, path: str):
        self.fd = os.open(path, os.O_RDONLY)
        self.base = 0

    def read(self, addr: int, n: int) -> bytes:
        b = os.pread(self.fd, n, self.base + addr)
        while 0 < len(b) < n:                       # (pread may return short on some filesystems)
            more = os.pread(self.fd, n - len(b), self.base + addr + len(b))
            if not more:
                break
            b += more
        if len(b) != n:
            raise H5LiteError("short read at %d (+%d)" % (addr, n))
        return b

    def close(self):
        if self.fd >= 0:
            os.close(self.fd)
            self.fd = -1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _superblock(fh: _File) -> int:
    """-> address of the root group's object header"""
    head = fh.read(0, 8 + 16)
    if head[:8] != SIG:
        raise H5LiteError("not an HDF5 file (signature at offset 0 missing)")
    ver = head[8]
    if ver not in (0, 1):
        raise H5LiteError("superblock version %d not supported (only 0/1: h5py default libver='earliest')" % ver)
    size_off, size_len = head[13], head[14]
    if size_off != 8 or size_len != 8:
        raise H5LiteError("only 8-byte offsets / lengths supported")
    p = 8 + 16 + (4 if ver == 1 else 0)
    body = fh.read(p, 32 + 40)
    base, _free, _eof, _drv = struct.unpack_from("<QQQQ", body, 0)
    fh.base = base
    # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
    _lno, ohdr, _cache = struct.unpack_from("<QQI", body, 32)
    return ohdr


def _messages(fh: _File, addr: int) -> List[Tuple[int, bytes]]:
    """All header messages of a version-1 object header (following continuation blocks)."""
    h = fh.read(addr, 16)
    if h[0] != 1:
        raise H5LiteError("object header version %d not supported" % h[0])
    nmsg = struct.unpack_from("<H", h, 2)[0]
    size = struct.unpack_from("<I", h, 8)[0]
    blocks = [(addr + 16, size)]
    out = []
    while blocks and len(out) < nmsg:
        a, n = blocks.pop(0)
        buf = fh.read(a, n)
        p = 0
        while p + 8 <= n and len(out) < nmsg:
            mtype, msize, _flags = struct.unpack_from("<HHB", buf, p)
            data = buf[p + 8:p + 8 + msize]
            p += 8 + msize
            if mtype == 0x0010:                    # continuation
                ca, cn = struct.unpack_from("<QQ", data, 0)
                blocks.append((ca, cn))
            out.append((mtype, data))
    return out


def _group_links(fh: _File, ohdr: int) -> dict:
    """name -> object header address for an old-style (symbol table) group"""
    st = [d for t, d in _messages(fh, ohdr) if t == 0x0011]
    if not st:
        raise H5LiteError("root group has no symbol-table message (new-style groups are not supported)")
    btree, heap = struct.unpack_from("<QQ", st[0], 0)
    hh = fh.read(heap, 32)
    if hh[:4] != b"HEAP":
        raise H5LiteError("local heap signature missing")
    heap_data = struct.unpack_from("<Q", hh, 24)[0]
    links = {}

    def name_at(off):
        s = b""
        while True:
            c = fh.read(heap_data + off + len(s), 1)
            if c == b"\0":
                return s.decode()
            s += c

    def walk(node):
        nh = fh.read(node, 24)
        if nh[:4] != b"TREE":
            raise H5LiteError("B-tree node signature missing")
        ntype, level, used = nh[4], nh[5], struct.unpack_from("<H", nh, 6)[0]
        if ntype != 0:
            raise H5LiteError("unexpected B-tree node type")
        body = fh.read(node + 24, (2 * used + 1) * 8)
        for i in range(used):
            child = struct.unpack_from("<Q", body, 8 + 16 * i)[0]
            if level > 0:
                walk(child)
                continue
            sh = fh.read(child, 8)
            if sh[:4] != b"SNOD":
                raise H5LiteError("symbol node signature missing")
            nsym = struct.unpack_from("<H", sh, 6)[0]
            ent = fh.read(child + 8, nsym * 40)
            for k in range(nsym):
                lno, oh = struct.unpack_from("<QQ", ent, 40 * k)
                links[name_at(lno)] = oh

    walk(btree)
    return links


def _dataset(fh: _File, ohdr: int):
    """-> (dims, datatype class, element size, data address, data size, raw datatype message)"""
    dims = dtcls = dtsize = daddr = dsize = dt = None
    for t, d in _messages(fh, ohdr):
        if t == 0x0001:                            # dataspace
            ver, rank, flags = d[0], d[1], d[2]
            off = 8 if ver == 1 else 4
            dims = list(struct.unpack_from("<%dQ" % rank, d, off))
        elif t == 0x0003:                          # datatype
            dtcls, dtsize, dt = d[0] & 0x0F, struct.unpack_from("<I", d, 4)[0], d
        elif t == 0x0008:                          # layout
            if d[0] != 3:
                raise H5LiteError("layout message version %d not supported" % d[0])
            if d[1] != 1:
                raise H5LiteError("only CONTIGUOUS datasets are supported (layout class %d)" % d[1])
            daddr, dsize = struct.unpack_from("<QQ", d, 2)
    if None in (dims, dtcls, daddr):
        raise H5LiteError("dataset header incomplete")
    return dims, dtcls, dtsize, daddr, dsize, dt


class ImageFile:
    """Random access to /images and /shapes of a dataset_tool_h5.py file."""

    def __init__(self, path: str):
        self.fh = _File(path)
        self.path = path
        self._map, self._map_pid = None, -1
        links = _group_links(self.fh, _superblock(self.fh))
        for name in ("images", "shapes"):
            if name not in links:
                raise H5LiteError("dataset '%s' not found (have: %s)" % (name, sorted(links)))
        sd, scls, ssize, saddr, _, _ = _dataset(self.fh, links["shapes"])
        if scls != 0 or ssize != 4 or len(sd) != 2 or sd[1] != 3:
            raise H5LiteError("/shapes must be int32 [N,3]")
        self.shapes = np.frombuffer(self.fh.read(saddr, sd[0] * 12), dtype="<i4").reshape(sd[0], 3).copy() if saddr != UNDEF \
            else np.zeros((sd[0], 3), np.int32)
        idims, icls, isize, iaddr, _, _ = _dataset(self.fh, links["images"])
        if icls != 9 or isize != 16 or len(idims) != 1 or idims[0] != sd[0]:
            raise H5LiteError("/images must be a variable-length uint8 sequence dataset [N]")
        self.n, self.iaddr = idims[0], iaddr
        self._heaps = {}

    def __len__(self) -> int:
        return self.n

    def _heap_object(self, addr: int, index: int) -> Tuple[int, int]:
        """(file offset, size) of object `index` in the global heap collection at `addr`"""
        col = self._heaps.get(addr)
        if col is None:
            h = self.fh.read(addr, 16)
            if h[:4] != b"GCOL":
                raise H5LiteError("global heap collection signature missing")
            size = struct.unpack_from("<Q", h, 8)[0]
            col, p = {}, 16
            while p + 16 <= size:
                oh = self.fh.read(addr + p, 16)
                idx, _ref, _res, osz = struct.unpack("<HHIQ", oh)
                if idx == 0:
                    break
                col[idx] = (addr + p + 16, osz)
                p += 16 + (osz + 7) // 8 * 8
            if len(self._heaps) > 64:
                self._heaps.clear()
            self._heaps[addr] = col
        if index not in col:
            raise H5LiteError("global heap object %d missing" % index)
        return col[index]

    def image_bytes(self, i: int) -> np.ndarray:
        ln, haddr, hidx = struct.unpack("<IQI", self.fh.read(self.iaddr + 16 * i, 16))
        off, osz = self._heap_object(haddr, hidx)
        if osz < ln:
            raise H5LiteError("heap object shorter than the sequence length")
        return np.frombuffer(self.fh.read(off, ln), dtype=np.uint8)

    def image(self, i: int) -> np.ndarray:
        """uint8 array with the stored shape (3, h, w)"""
        return self.image_bytes(i).reshape(tuple(int(v) for v in self.shapes[i]))

    def image_location(self, i: int) -> Tuple[int, int]:
        """(absolute file offset, byte length) of image i's raw CHW payload"""
        loc = self._locs.get(i) if hasattr(self, "_locs") else None
        if loc is None:
            if not hasattr(self, "_locs"):
                self._locs = {}
            ln, haddr, hidx = struct.unpack("<IQI", self.fh.read(self.iaddr + 16 * i, 16))
            off, osz = self._heap_object(haddr, hidx)
            if osz < ln:
                raise H5LiteError("heap object shorter than the sequence length")
            loc = self._locs[i] = (self.fh.base + off, ln)
        return loc

    def view(self, i: int) -> np.ndarray:
        """Image i as a read-only (3, h, w) VIEW of the memory-mapped file: slicing a window out of it touches only the pages
        of that window (the training loader crops 64x64 patches out of ~0.5 MB images).  The mapping is per process and
        read-only: nothing a forked DataLoader worker could race on."""
        if getattr(self, "_map", None) is None or self._map_pid != os.getpid():
            self._map = np.memmap(self.path, dtype=np.uint8, mode="r")
            self._map_pid = os.getpid()
        off, ln = self.image_location(i)
        return self._map[off:off + ln].reshape(tuple(int(v) for v in self.shapes[i]))

    def close(self):
        self.fh.close()


# ------------------------------------------------------------------------------------------------------------------------------
# writer of the same subset (test fixtures / synthetic datasets; NOT a general HDF5 writer)
# ------------------------------------------------------------------------------------------------------------------------------
def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHBBBB", mtype, len(data), flags, 0, 0, 0) + data


def _ohdr(msgs: List[bytes]) -> bytes:
    body = b"".join(msgs)
    return struct.pack("<BBHII", 1, 0, len(msgs), 1, len(body)) + b"\0" * 4 + body


def write_dataset_file(path: str, images: List[np.ndarray]):
    """images: list of uint8 arrays [3, h, w] -> a file with /shapes and /images laid out as dataset_tool_h5.py does."""
    n = len(images)
    blob = bytearray(b"\0" * 96)                                   # superblock (v0) goes here at the end

    def put(b: bytes) -> int:
        a = len(blob)
        blob.extend(_pad8(b))
        return a

    # global heap: one collection per image (simple, valid)
    desc = b""
    for i, im in enumerate(images):
        raw = np.ascontiguousarray(im, dtype=np.uint8).tobytes()
        obj = struct.pack("<HHIQ", 1, 0, 0, len(raw)) + _pad8(raw)
        free = struct.pack("<HHIQ", 0, 0, 0, 0)
        size = 16 + len(obj) + len(free)
        addr = put(b"GCOL" + struct.pack("<B3xQ", 1, size) + obj + free)
        desc += struct.pack("<IQI", len(raw), addr, 1)
    images_data = put(desc) if n else UNDEF
    shapes = np.array([im.shape for im in images], dtype="<i4").reshape(n, 3)
    shapes_data = put(shapes.tobytes()) if n else UNDEF

    def dataspace(dims):
        return _msg(0x0001, struct.pack("<BBB5x", 1, len(dims), 0) + b"".join(struct.pack("<Q", d) for d in dims))

    int32 = struct.pack("<BBBBI", 0x10 | 0, 0x08, 0, 0, 4) + struct.pack("<HH", 0, 32)         # fixed point, LE, signed
    uint8 = struct.pack("<BBBBI", 0x10 | 0, 0x00, 0, 0, 1) + struct.pack("<HH", 0, 8)
    vlen = struct.pack("<BBBBI", 0x10 | 9, 0x00, 0, 0, 16) + uint8                              # vlen sequence of uint8

    def layout(addr, size):
        return _msg(0x0008, struct.pack("<BBQQ", 3, 1, addr, size))

    oh_shapes = put(_ohdr([dataspace([n, 3]), _msg(0x0003, int32, 1), layout(shapes_data, n * 12)]))
    oh_images = put(_ohdr([dataspace([n]), _msg(0x0003, vlen, 1), layout(images_data, n * 16)]))
    # root group: local heap with the two names, one symbol node, one B-tree leaf
    names = b"\0" * 8 + b"images\0\0" + b"shapes\0\0"                 # offsets: images 8, shapes 16
    heap_data = put(names + b"\0" * 64)
    heap = put(b"HEAP" + struct.pack("<B3xQQQ", 0, len(names) + 64, UNDEF, heap_data))

    def sym(lno, oh):
        return struct.pack("<QQII16x", lno, oh, 0, 0)

    snod = put(b"SNOD" + struct.pack("<BBH", 1, 0, 2) + sym(8, oh_images) + sym(16, oh_shapes) + b"\0" * 40 * 14)
    btree = put(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod, 16))
    root = put(_ohdr([_msg(0x0011, struct.pack("<QQ", btree, heap))]))
    sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, len(blob), UNDEF)
    sb += struct.pack("<QQII16x", 0, root, 0, 0)
    blob[0:len(sb)] = sb
    with open(path, "wb") as f:
        f.write(bytes(blob))
