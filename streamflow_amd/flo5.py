"""``.flo5``: Spring's flow container -- an HDF5 file with ONE dataset ``flow`` (float32 [H, W, 2], gzip level 5), as
the reference writes and reads it through h5py (core/utils/frame_utils.py:31-47 ``writeFlo5File``, :130-135
``readFlo5Flow``).  h5py / libhdf5 are not importable from this package's interpreter, so this is a small codec for exactly that subset of the
published HDF5 File Format Specification (version 1.1 structures, what libhdf5 emits with ``libver='earliest'``,
h5py's default):

* superblock version 0 or 1, 8-byte offsets and lengths;
* old-style groups: symbol-table message -> version-1 B-tree (node type 0) -> symbol-table nodes + local heap;
* version-1 object headers (continuation blocks followed);
* dataspace v1 / v2, datatype class 0 (integers) and 1 (IEEE floats) of 1 / 2 / 4 / 8 bytes, either byte order;
* data layout v3: compact, contiguous, or chunked through a version-1 B-tree (node type 1, any depth);
* filter pipeline v1 / v2 with deflate (id 1) and shuffle (id 2); other filters raise.

``write_flo5`` emits the same structures (one chunked, deflate-compressed dataset under the root group).
Parity is pinned both ways against the real library (h5py 3.3.0 / HDF5 1.10.6 of the build container's conda interpreter,
which is not this package's python): the reader decodes, bit for bit, the files under tests/golden/flo5 that h5py wrote with
the reference's own call (tests/golden/make_flo5_golden.py), and files from ``write_flo5`` open in h5py with the same values,
gzip level 5 (tests/test_flow_io_cpu.py); plus the round trips and byte-level structure checks.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, List, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
GROUP_LEAF_K, GROUP_INTERNAL_K, CHUNK_K = 4, 16, 32      # libhdf5 defaults (superblock v0 has no field for the last)
HEAP_FREE_NULL = 1                                       # "end of free list" on disk (H5HL)

MSG_DATASPACE, MSG_DATATYPE, MSG_FILL_OLD, MSG_FILL, MSG_LAYOUT, MSG_FILTERS = 0x1, 0x3, 0x4, 0x5, 0x8, 0xB
MSG_CONTINUATION, MSG_SYMBOL_TABLE = 0x10, 0x11


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


# ------------------------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------------------------
def _message(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(messages: List[bytes]) -> bytes:
    body = b"".join(messages)
    # version, reserved, number of messages, reference count, header size; 4 bytes of padding align the first message
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body


def _float32_datatype() -> bytes:
    # class 1 (floating point), version 1; bit field: little-endian, mantissa normalisation 2 (msb implied), sign bit 31
    head = struct.pack("<B3BI", 0x11, 0x20, 0x1F, 0x00, 4)
    props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)   # bit offset, precision, exp loc/size, mant loc/size, bias
    return head + props


def write_flo5(path: str, flow: np.ndarray, compression_level: int = 5, rows_per_chunk: int = 0) -> None:
    """flow [H, W, 2] -> HDF5 file with dataset 'flow' (float32, chunked, gzip), like frame_utils.py:46-47."""
    flow = np.ascontiguousarray(flow, dtype="<f4")
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError(f"writeFlo5File {path}: expected shape height x width x 2 but received {flow.shape}")
    H, W, _ = flow.shape
    if rows_per_chunk <= 0:                                 # <= 64 chunks: the chunk index is ONE B-tree leaf
        rows_per_chunk = max(1, -(-H // 32))
    n_chunks = -(-H // rows_per_chunk)
    if n_chunks > 2 * CHUNK_K:
        raise ValueError("rows_per_chunk too small: more than 64 chunks")
    cdims = (rows_per_chunk, W, 2)

    # ---- layout of the file: fixed-size metadata first, chunk data last
    off_super, off_root = 0, 96
    root_hdr_size = 16 + 8 + 16                              # prefix + one symbol-table message
    off_btree = off_root + root_hdr_size
    btree_size = 24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8
    off_heap = off_btree + btree_size
    heap_data_size = 32                                      # "" | "flow" | one 16-byte free block
    off_heap_data = off_heap + 32
    off_snod = off_heap_data + heap_data_size
    snod_size = 8 + 2 * GROUP_LEAF_K * 40
    off_dset = off_snod + snod_size

    dataspace = struct.pack("<BBB5x", 1, 3, 0) + struct.pack("<3Q", H, W, 2)
    fill = struct.pack("<BBBB", 2, 3, 2, 0)                  # v2: allocate incrementally, write fill if set, undefined
    filters = (struct.pack("<BB6x", 1, 1) +                  # v1 pipeline, one filter
               struct.pack("<HHHH", 1, 8, 1, 1) + b"deflate\0" + struct.pack("<I4x", int(compression_level)))
    layout_size = 2 + 1 + 8 + 4 * 4
    dset_msgs_wo_layout = [_message(MSG_DATASPACE, dataspace), _message(MSG_DATATYPE, _float32_datatype(), 1),
                           _message(MSG_FILL, fill), _message(MSG_FILTERS, filters, 1)]
    dset_hdr_size = 16 + sum(len(m) for m in dset_msgs_wo_layout) + 8 + (layout_size + 7) // 8 * 8
    off_cbtree = off_dset + dset_hdr_size
    key_size = 8 + 8 * 4
    cbtree_size = 24 + 2 * CHUNK_K * 8 + (2 * CHUNK_K + 1) * key_size
    off_chunks = off_cbtree + cbtree_size

    # ---- chunks (edge chunks are stored full-size, zero padded, as libhdf5 does)
    blobs, addrs = [], []
    pos = off_chunks
    for c in range(n_chunks):
        buf = np.zeros(cdims, "<f4")
        rows = flow[c * rows_per_chunk:(c + 1) * rows_per_chunk]
        buf[: rows.shape[0]] = rows
        z = zlib.compress(buf.tobytes(), int(compression_level))
        blobs.append(z)
        addrs.append(pos)
        pos += len(z)
    eof = pos

    layout = struct.pack("<BBB", 3, 2, 4) + struct.pack("<Q", off_cbtree) + struct.pack("<4I", *cdims, 4)
    dset_hdr = _object_header(dset_msgs_wo_layout + [_message(MSG_LAYOUT, layout)])
    assert len(dset_hdr) == dset_hdr_size

    cb = struct.pack("<4sBBHQQ", b"TREE", 1, 0, n_chunks, UNDEF, UNDEF)
    for c in range(n_chunks):
        cb += struct.pack("<II4Q", len(blobs[c]), 0, c * rows_per_chunk, 0, 0, 0) + struct.pack("<Q", addrs[c])
    cb += struct.pack("<II4Q", 0, 0, n_chunks * rows_per_chunk, 0, 0, 0)              # final key: one past the last chunk
    cb += b"\0" * (cbtree_size - len(cb))

    sup = (SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", GROUP_LEAF_K, GROUP_INTERNAL_K, 0) +
           struct.pack("<4Q", 0, UNDEF, eof, UNDEF) +
           struct.pack("<QQII", 0, off_root, 1, 0) + struct.pack("<QQ", off_btree, off_heap))
    assert len(sup) == 96
    root_hdr = _object_header([_message(MSG_SYMBOL_TABLE, struct.pack("<QQ", off_btree, off_heap))])
    assert len(root_hdr) == root_hdr_size
    gb = struct.pack("<4sBBHQQ", b"TREE", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, off_snod, 8)
    gb += b"\0" * (btree_size - len(gb))
    heap = struct.pack("<4sB3xQQQ", b"HEAP", 0, heap_data_size, 16, off_heap_data)
    heap_data = _pad8(b"\0") + _pad8(b"flow\0") + struct.pack("<QQ", HEAP_FREE_NULL, 16)
    assert len(heap) == 32 and len(heap_data) == heap_data_size
    snod = struct.pack("<4sBxH", b"SNOD", 1, 1) + struct.pack("<QQII16x", 8, off_dset, 0, 0)
    snod += b"\0" * (snod_size - len(snod))

    with open(path, "wb") as f:
        for part in (sup, root_hdr, gb, heap, heap_data, snod, dset_hdr, cb, *blobs):
            f.write(part)
        assert f.tell() == eof


# ------------------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------------------
class _File:
    def __init__(self, data: bytes, path: str):
        self.d, self.path = data, path
        start = 0
        while True:                                           # the superblock may sit at 0, 512, 1024, ...
            if data[start:start + 8] == SIGNATURE:
                break
            start = 512 if start == 0 else start * 2
            if start >= len(data):
                raise IOError(f"{path}: not an HDF5 file")
        ver = data[start + 8]
        if ver > 1:
            raise IOError(f"{path}: superblock version {ver} (libver='latest' files) is not supported by this reader")
        so, sl = data[start + 13], data[start + 14]
        if (so, sl) != (8, 8):
            raise IOError(f"{path}: only 8-byte offsets / lengths are supported (file has {so}/{sl})")
        p = start + 24 + (4 if ver == 1 else 0)
        self.base, _, self.eof, _ = struct.unpack_from("<4Q", data, p)
        p += 32
        _, self.root_header, _cache_type, _ = struct.unpack_from("<QQII", data, p)

    def at(self, addr: int, n: int) -> bytes:
        a = self.base + addr
        if addr == UNDEF or a + n > len(self.d):
            raise IOError(f"{self.path}: address {addr:#x} (+{n}) outside the file")
        return self.d[a:a + n]

    # -- object headers -----------------------------------------------------------------------------------------
    def messages(self, addr: int) -> List[Tuple[int, bytes]]:
        ver, nmsg, _, size = struct.unpack_from("<BxHII", self.at(addr, 12))
        if ver != 1:
            raise IOError(f"{self.path}: object header version {ver} is not supported")
        out: List[Tuple[int, bytes]] = []
        blocks = [(addr + 16, size)]
        while blocks and len(out) < nmsg:
            a, n = blocks.pop(0)
            blk = self.at(a, n)
            p = 0
            while p + 8 <= n and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", blk, p)
                body = blk[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == MSG_CONTINUATION:
                    blocks.append(struct.unpack_from("<QQ", body))
                out.append((mtype, body))
        return out

    # -- old-style groups ---------------------------------------------------------------------------------------
    def group_entries(self, header_addr: int) -> Dict[str, int]:
        st = [b for t, b in self.messages(header_addr) if t == MSG_SYMBOL_TABLE]
        if not st:
            raise IOError(f"{self.path}: root group without a symbol table (new-style links are not supported)")
        btree, heap = struct.unpack_from("<QQ", st[0])
        sig, _, seg_size, _, seg_addr = struct.unpack_from("<4sB3xQQQ", self.at(heap, 32))
        if sig != b"HEAP":
            raise IOError(f"{self.path}: bad local heap")
        names = self.at(seg_addr, seg_size)
        out: Dict[str, int] = {}

        def walk(addr: int) -> None:
            sig, ntype, level, used = struct.unpack_from("<4sBBH", self.at(addr, 8))
            if sig != b"TREE" or ntype != 0:
                raise IOError(f"{self.path}: bad group B-tree node")
            body = self.at(addr + 24, (2 * used + 1) * 8)
            for i in range(used):
                child = struct.unpack_from("<Q", body, (2 * i + 1) * 8)[0]
                if level > 0:
                    walk(child)
                    continue
                sig, _, nsym = struct.unpack_from("<4sBxH", self.at(child, 8))
                if sig != b"SNOD":
                    raise IOError(f"{self.path}: bad symbol table node")
                ent = self.at(child + 8, nsym * 40)
                for j in range(nsym):
                    noff, ohdr = struct.unpack_from("<QQ", ent, j * 40)
                    end = names.index(b"\0", noff)
                    out[names[noff:end].decode()] = ohdr

        walk(btree)
        return out

    # -- datasets -----------------------------------------------------------------------------------------------
    def dataset(self, header_addr: int) -> np.ndarray:
        msgs = self.messages(header_addr)
        get = lambda t: next((b for mt, b in msgs if mt == t), None)
        sp, dt, lay, flt = get(MSG_DATASPACE), get(MSG_DATATYPE), get(MSG_LAYOUT), get(MSG_FILTERS)
        if sp is None or dt is None or lay is None:
            raise IOError(f"{self.path}: dataset header lacks dataspace / datatype / layout")
        sver, rank = sp[0], sp[1]
        shape = struct.unpack_from(f"<{rank}Q", sp, 8 if sver == 1 else 4)
        dtype = self._dtype(dt)
        filters = self._filters(flt) if flt is not None else []
        lver, lclass = lay[0], lay[1]
        if lver != 3:
            raise IOError(f"{self.path}: data layout version {lver} is not supported")
        count = int(np.prod(shape)) if rank else 1
        if lclass == 0:                                       # compact
            n = struct.unpack_from("<H", lay, 2)[0]
            return np.frombuffer(lay[4:4 + n], dtype, count).reshape(shape).copy()
        if lclass == 1:                                       # contiguous
            addr, n = struct.unpack_from("<QQ", lay, 2)
            return np.frombuffer(self.at(addr, n), dtype, count).reshape(shape).copy()
        if lclass != 2:
            raise IOError(f"{self.path}: unknown layout class {lclass}")
        nd = lay[2]
        btree = struct.unpack_from("<Q", lay, 3)[0]
        cdims = struct.unpack_from(f"<{nd}I", lay, 11)
        if nd != rank + 1 or cdims[-1] != dtype.itemsize:
            raise IOError(f"{self.path}: chunk dimensionality {nd} does not match rank {rank}")
        cshape = cdims[:-1]
        out = np.zeros(shape, dtype)
        if btree != UNDEF:
            self._chunks(btree, nd, cshape, dtype, filters, out)
        return out

    def _chunks(self, addr, nd, cshape, dtype, filters, out) -> None:
        sig, ntype, level, used = struct.unpack_from("<4sBBH", self.at(addr, 8))
        if sig != b"TREE" or ntype != 1:
            raise IOError(f"{self.path}: bad chunk B-tree node")
        ksz = 8 + 8 * nd
        body = self.at(addr + 24, used * (ksz + 8) + ksz)
        for i in range(used):
            p = i * (ksz + 8)
            nbytes, mask = struct.unpack_from("<II", body, p)
            offs = struct.unpack_from(f"<{nd}Q", body, p + 8)[:-1]
            child = struct.unpack_from("<Q", body, p + ksz)[0]
            if level > 0:
                self._chunks(child, nd, cshape, dtype, filters, out)
                continue
            raw = self.at(child, nbytes)
            for k in range(len(filters) - 1, -1, -1):         # decode in reverse pipeline order; mask bit k = skipped
                if mask >> k & 1:
                    continue
                fid = filters[k]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    n = len(raw) // dtype.itemsize
                    raw = np.frombuffer(raw, np.uint8).reshape(dtype.itemsize, n).T.tobytes()
                else:
                    raise IOError(f"{self.path}: HDF5 filter {fid} is not supported (deflate and shuffle are)")
            chunk = np.frombuffer(raw, dtype, int(np.prod(cshape))).reshape(cshape)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, out.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]

    def _dtype(self, dt: bytes) -> np.dtype:
        cls, b0 = dt[0] & 0x0F, dt[1]
        size = struct.unpack_from("<I", dt, 4)[0]
        order = ">" if b0 & 1 else "<"
        if cls == 1 and size in (2, 4, 8):
            return np.dtype(f"{order}f{size}")
        if cls == 0 and size in (1, 2, 4, 8):
            return np.dtype(f"{order}{'i' if b0 & 8 else 'u'}{size}")
        raise IOError(f"{self.path}: datatype class {cls} size {size} is not supported")

    def _filters(self, b: bytes) -> List[int]:
        ver, n = b[0], b[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from("<H", b, p)[0]
            if ver == 1 or fid >= 256:
                nlen, _flags, ncd = struct.unpack_from("<HHH", b, p + 2)
                p += 8 + (nlen + 7) // 8 * 8 if ver == 1 else 8 + nlen
            else:
                _flags, ncd = struct.unpack_from("<HH", b, p + 2)
                p += 6
            p += 4 * ncd + (4 if (ver == 1 and ncd % 2) else 0)
            out.append(fid)
        return out


def read_hdf5_dataset(path: str, name: str) -> np.ndarray:
    with open(path, "rb") as f:
        hf = _File(f.read(), path)
    entries = hf.group_entries(hf.root_header)
    if name not in entries:
        raise IOError(f"File {path} does not have a '{name}' key. Is this a valid flo5 file?")
    return hf.dataset(entries[name])


def read_flo5(path: str) -> np.ndarray:
    """frame_utils.py:130-135 readFlo5Flow: f['flow'][()]"""
    return read_hdf5_dataset(path, "flow")
