"""A small pure-Python reader (and fixture writer) for the corner of HDF5 the reference's feature stores use
(VLN-HAMT/finetune_src/r2r/data_utils.py:15-47: one 2-D float dataset per key in the root group, read whole with `f[key][...]`).

h5py is not part of this image, and the run-time loaders (formats.py) must not depend on it; the feature files written by the
reference's preprocessing are "classic" HDF5: superblock version 0/1, the root group as a symbol table (v1 B-tree + local heap),
version-1 object headers, contiguous or chunked (+ deflate / shuffle) dataset layouts. That subset is restated here from the HDF5
File Format Specification (version 3.0, sections III.A-C, IV.A.1, IV.A.2.b/d/f/i/l); anything else (superblock v2/v3, link-message
groups, virtual / external storage, compound types, szip) raises NotImplementedError with the feature's name. When h5py is importable
formats.py uses it instead.

write_store() emits the same subset (what `h5py.File(...).create_dataset(key, data=..., compression='gzip')` produces structurally)
and exists to build the test fixtures (tests/golden/make_hdf5_fixture.py); it is not a general HDF5 writer."""
import struct
import zlib

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


# =====================================================================================================================
#  reader
# =====================================================================================================================
class Hdf5File:
    """Read-only view of a classic HDF5 file: keys() of the root group and dataset(key) -> numpy array."""

    def __init__(self, path):
        import mmap
        with open(path, "rb") as f:                             # mapped, not read: the view-feature stores are > 1 GB and every rank opens them
            self.buf = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        b = self.buf
        if b[:8] != SIG:
            raise ValueError(f"{path}: not an HDF5 file")
        ver = b[8]
        if ver not in (0, 1):
            raise NotImplementedError(f"{path}: superblock version {ver} (only the classic versions 0 / 1 are read)")
        self.O, self.L = b[13], b[14]
        if (self.O, self.L) != (8, 8):
            raise NotImplementedError(f"{path}: {self.O}-byte offsets / {self.L}-byte lengths")
        p = 24 + (4 if ver == 1 else 0)
        self.base = self._u64(p)
        root = p + 32                                         # root group symbol table entry
        self._links = {}
        hdr = self._u64(root + 8)
        cache = struct.unpack_from("<I", b, root + 16)[0]
        if cache == 1:
            btree, heap = self._u64(root + 24), self._u64(root + 32)
        else:
            msgs = self._messages(hdr)
            st = [m for m in msgs if m[0] == 0x11]
            if not st:
                raise NotImplementedError("root group without a symbol table (new-style link messages)")
            btree, heap = struct.unpack_from("<QQ", st[0][1])
        self._walk_group(btree, self._heap_data(heap))

    # ---- primitives ----
    def _u64(self, p):
        return struct.unpack_from("<Q", self.buf, p)[0]

    def _heap_data(self, addr):
        a = self.base + addr
        if self.buf[a:a + 4] != b"HEAP":
            raise ValueError("bad local heap signature")
        return self.base + self._u64(a + 24)

    def _name(self, heap_data, off):
        e = self.buf.find(b"\0", heap_data + off)
        if e < 0:
            raise ValueError("hdf5_lite: a link name in the local heap has no terminator (truncated or corrupt file)")
        return self.buf[heap_data + off:e].decode()

    def _walk_group(self, addr, heap_data):
        a = self.base + addr
        sig = self.buf[a:a + 4]
        if sig == b"TREE":
            ntype, level, used = self.buf[a + 4], self.buf[a + 5], struct.unpack_from("<H", self.buf, a + 6)[0]
            if ntype != 0:
                raise ValueError("group B-tree expected")
            p = a + 8 + 16                                    # past the sibling pointers
            for i in range(used):
                child = self._u64(p + 8 + i * 16)             # key_i (8), child_i (8), ...
                self._walk_group(child, heap_data)
        elif sig == b"SNOD":
            n = struct.unpack_from("<H", self.buf, a + 6)[0]
            for i in range(n):
                e = a + 8 + 40 * i
                self._links[self._name(heap_data, self._u64(e))] = self._u64(e + 8)
        else:
            raise ValueError(f"unexpected group node signature {sig!r}")

    def _messages(self, addr):
        """[(type, payload bytes)] of a version-1 object header, following continuation blocks."""
        a = self.base + addr
        if self.buf[a] != 1:
            raise NotImplementedError(f"object header version {self.buf[a]} (only version 1)")
        nmsg = struct.unpack_from("<H", self.buf, a + 2)[0]
        size = struct.unpack_from("<I", self.buf, a + 8)[0]
        blocks, out = [(a + 16, size)], []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = struct.unpack_from("<HH", self.buf, p)
                data = self.buf[p + 8:p + 8 + msize]
                if mtype == 0x10:
                    off, ln = struct.unpack_from("<QQ", data)
                    blocks.append((self.base + off, ln))
                out.append((mtype, data))
                p += 8 + msize
        return out

    # ---- public ----
    def keys(self):
        return sorted(self._links)

    def __contains__(self, k):
        return k in self._links

    def dataset(self, key):
        msgs = dict()
        for t, d in self._messages(self._links[key]):
            msgs.setdefault(t, d)
        if 0x1 not in msgs or 0x3 not in msgs or 0x8 not in msgs:
            raise NotImplementedError(f"{key}: not a simple dataset")
        shape = self._dataspace(msgs[0x1])
        dt = self._datatype(msgs[0x3])
        lay = msgs[0x8]
        if lay[0] != 3:
            raise NotImplementedError(f"{key}: data layout message version {lay[0]}")
        cls = lay[1]
        n = int(np.prod(shape)) if shape else 1
        if cls == 0:                                          # compact
            size = struct.unpack_from("<H", lay, 2)[0]
            return np.frombuffer(lay[4:4 + size], dt, n).reshape(shape).copy()
        if cls == 1:                                          # contiguous
            addr = struct.unpack_from("<Q", lay, 2)[0]
            if addr == UNDEF:
                return np.zeros(shape, dt)
            return np.frombuffer(self.buf, dt, n, self.base + addr).reshape(shape).copy()
        if cls == 2:                                          # chunked
            nd = lay[2]
            btree = struct.unpack_from("<Q", lay, 3)[0]
            cdims = struct.unpack_from("<" + "I" * nd, lay, 11)
            filters = self._filters(msgs[0xB]) if 0xB in msgs else []
            out = np.zeros(shape, dt)
            if btree != UNDEF:
                self._walk_chunks(btree, nd, cdims[:-1], filters, dt, out)
            return out
        raise NotImplementedError(f"{key}: layout class {cls}")

    def __getitem__(self, key):
        return self.dataset(key)

    @staticmethod
    def _dataspace(d):
        ver, rank, flags = d[0], d[1], d[2]
        if ver == 1:
            return struct.unpack_from("<" + "Q" * rank, d, 8)
        if ver == 2:
            return struct.unpack_from("<" + "Q" * rank, d, 4)
        raise NotImplementedError(f"dataspace message version {ver}")

    @staticmethod
    def _datatype(d):
        cls, bits0, size = d[0] & 0x0F, d[1], struct.unpack_from("<I", d, 4)[0]
        order = ">" if (bits0 & 1) else "<"
        if cls == 1:
            if size not in (2, 4, 8):
                raise NotImplementedError(f"{size}-byte float")
            return np.dtype(f"{order}f{size}")
        if cls == 0:
            return np.dtype(f"{order}{'i' if (bits0 & 8) else 'u'}{size}")
        raise NotImplementedError(f"datatype class {cls} (only fixed-point and floating-point)")

    @staticmethod
    def _filters(d):
        ver, n = d[0], d[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from("<H", d, p)[0]
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", d, p + 2)[0]
                flags, ncv = struct.unpack_from("<HH", d, p + 4)
                p += 8
                p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            else:
                flags, ncv = struct.unpack_from("<HH", d, p + 2)
                p += 6
            cv = struct.unpack_from("<" + "I" * ncv, d, p)
            p += 4 * ncv
            if ver == 1 and ncv % 2:
                p += 4
            out.append((fid, cv))
        return out

    def _walk_chunks(self, addr, nd, cshape, filters, dt, out):
        a = self.base + addr
        if self.buf[a:a + 4] != b"TREE" or self.buf[a + 4] != 1:
            raise ValueError("chunk B-tree expected")
        level, used = self.buf[a + 5], struct.unpack_from("<H", self.buf, a + 6)[0]
        ksize = 8 + 8 * nd
        p = a + 24
        for i in range(used):
            k = p + i * (ksize + 8)
            nbytes, mask = struct.unpack_from("<II", self.buf, k)
            offs = struct.unpack_from("<" + "Q" * nd, self.buf, k + 8)[:-1]
            child = self._u64(k + ksize)
            if level > 0:
                self._walk_chunks(child, nd, cshape, filters, dt, out)
                continue
            raw = self.buf[self.base + child:self.base + child + nbytes]
            for j, (fid, cv) in reversed(list(enumerate(filters))):
                if mask & (1 << j):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                                # shuffle: bytes of each element were de-interleaved
                    es = cv[0] if cv else dt.itemsize
                    raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
                elif fid == 3:                                # fletcher32: checksum behind the data
                    raw = raw[:-4]
                else:
                    raise NotImplementedError(f"HDF5 filter {fid}")
            chunk = np.frombuffer(raw, dt, int(np.prod(cshape))).reshape(cshape)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, out.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]


def read_store(path):
    """{key: array} of every dataset in the root group."""
    f = Hdf5File(path)
    return {k: f.dataset(k) for k in f.keys()}


# =====================================================================================================================
#  fixture writer (classic layout: superblock v0, symbol-table root group, v1 object headers)
# =====================================================================================================================
GROUP_K, CHUNK_K = 16, 32                                     # library defaults: children per B-tree node = 2 K


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(msgs):
    body = b"".join(msgs)
    return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        # class 1 version 1; bit fields: little-endian, IEEE pads, mantissa normalisation 2 (implied msb), sign bit position
        sign = dt.itemsize * 8 - 1
        prop = {4: struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127), 8: struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)}[dt.itemsize]
        return struct.pack("<BBBBI", 0x11, 0x20, sign, 0, dt.itemsize) + prop
    if dt.kind in "iu":
        return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    raise NotImplementedError(dt)


def write_store(path, arrays, chunks=None, compress=True, leaf_k=4):
    """Writes {key: 2-D array} as datasets of the root group. chunks=(r, c): chunked layout (+ deflate if `compress`), None:
    contiguous. leaf_k: symbols per group leaf = 2 * leaf_k (4 is the library default: more than 8 keys span several leaves)."""
    out = bytearray(b"\0" * 96)                               # superblock (24 + 32 + 40), filled in last

    def put(b):
        off = len(out)
        out.extend(_pad8(b))
        return off

    names = sorted(arrays)
    heap_data, name_off = bytearray(b"\0" * 8), {}
    for k in names:
        name_off[k] = len(heap_data)
        heap_data.extend(_pad8(k.encode() + b"\0"))
    hdr_addr = {}
    for k in names:
        a = np.ascontiguousarray(arrays[k])
        assert a.ndim == 2
        space = struct.pack("<BBB5x", 1, 2, 0) + struct.pack("<QQ", *a.shape)
        msgs = [_msg(0x1, space), _msg(0x3, _dtype_msg(a.dtype), 1), _msg(0x5, struct.pack("<BBBB", 2, 2, 2, 0))]
        if chunks is None:
            addr = put(a.tobytes())
            msgs.append(_msg(0x8, struct.pack("<BBQQ", 3, 1, addr, a.nbytes)))
        else:
            cr, cc = chunks
            keys = []
            for r0 in range(0, a.shape[0], cr):
                for c0 in range(0, a.shape[1], cc):
                    blk = np.zeros((cr, cc), a.dtype)
                    sub = a[r0:r0 + cr, c0:c0 + cc]
                    blk[:sub.shape[0], :sub.shape[1]] = sub
                    raw = zlib.compress(blk.tobytes(), 4) if compress else blk.tobytes()
                    keys.append((len(raw), r0, c0, put(raw)))
            node = b"TREE" + struct.pack("<BBHQQ", 1, 0, len(keys), UNDEF, UNDEF)
            for nbytes, r0, c0, addr in keys:
                node += struct.pack("<IIQQQ", nbytes, 0, r0, c0, 0) + struct.pack("<Q", addr)
            node += struct.pack("<IIQQQ", 0, 0, -(-a.shape[0] // cr) * cr, -(-a.shape[1] // cc) * cc, 0)   # the final key: one chunk past the last
            assert len(keys) <= 2 * CHUNK_K, "fixture writer: one chunk B-tree node"
            node += b"\0" * (24 + 2 * CHUNK_K * 8 + (2 * CHUNK_K + 1) * 32 - len(node))   # nodes are allocated at full size
            bt = put(node)
            if compress:
                msgs.append(_msg(0xB, struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<II", 4, 0)))
            msgs.append(_msg(0x8, struct.pack("<BBBQIII", 3, 2, 3, bt, cr, cc, a.dtype.itemsize)))
        hdr_addr[k] = put(_object_header(msgs))
    heap_seg = put(bytes(heap_data))
    heap = put(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, heap_seg))       # free-list head 1 = no free block (H5HL_FREE_NULL)
    per = 2 * leaf_k
    leaves = []
    for i in range(0, len(names), per):
        grp = names[i:i + per]
        node = b"SNOD" + struct.pack("<BBH", 1, 0, len(grp))
        for k in grp:
            node += struct.pack("<QQII16x", name_off[k], hdr_addr[k], 0, 0)
        node += b"\0" * (40 * (per - len(grp)))
        leaves.append((put(node), name_off[grp[-1]]))
    tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(leaves), UNDEF, UNDEF) + struct.pack("<Q", 0)
    for addr, last in leaves:
        tree += struct.pack("<QQ", addr, last)
    assert len(leaves) <= 2 * GROUP_K, "fixture writer: one group B-tree node"
    tree += b"\0" * (24 + 2 * GROUP_K * 8 + (2 * GROUP_K + 1) * 8 - len(tree))
    btree = put(tree)
    root = put(_object_header([_msg(0x11, struct.pack("<QQ", btree, heap))]))
    sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, leaf_k, GROUP_K, 0) + struct.pack("<QQQQ", 0, UNDEF, len(out), UNDEF)
    sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", btree, heap)
    out[:len(sb)] = sb
    with open(path, "wb") as f:
        f.write(bytes(out))
