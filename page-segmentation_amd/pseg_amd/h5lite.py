"""Minimal HDF5 reader / writer for Keras weight files (`model.h5`, lib/network.py:59,106-107,177-183).

h5py is not installed beside the product interpreter, and a Keras model file is the format trained
ocr4all models travel in, so this module reads and writes the small subset of HDF5 (file format
specification 1.8/1.10, "earliest" library version -- what h5py/Keras produce by default) that such
files use:

  * superblock version 0/1 (and 2/3 for reading), 8-byte offsets and lengths;
  * old-style groups: symbol-table message -> v1 B-tree ("TREE") -> symbol nodes ("SNOD") with names
    in a local heap ("HEAP"); version-1 object headers with continuation blocks (and version-2
    "OHDR" headers / compact link messages when reading);
  * datasets with contiguous or compact layout (and unfiltered chunked layout when reading), IEEE
    little-endian float / integer element types;
  * attributes holding numeric arrays, fixed-length string arrays or variable-length strings (global
    heap) -- h5py writes `layer_names` / `weight_names` either way depending on its version.

Keras layout (tensorflow/python/keras/saving/hdf5_format.py): the root group -- or `/model_weights`
in a full-model file written by ModelCheckpoint -- has the attribute `layer_names`; every layer is a
group with the attribute `weight_names` (e.g. b"conv2d/kernel:0") naming datasets relative to it.
`load_weights(by_name=False)` matches layers that have weights *by order*, which is what
`read_keras_weights` returns and `ocr4all_pixel_classifier.lib.network.Network` applies.
Nothing from the file is executed; unsupported constructs raise H5Error.
"""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(Exception):
    pass


# ------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------
class _Obj:
    def __init__(self):
        self.links = None      # name -> object header address (groups)
        self.attrs = {}
        self.shape = None
        self.dtype = None
        self.layout = None     # ("contiguous", addr, size) | ("compact", bytes) | ("chunked", btree, chunk_dims)


class H5File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.b = f.read()
        self.base = 0
        off = 0
        while True:
            if self.b[off:off + 8] == SIG:
                break
            off = 512 if off == 0 else off * 2
            if off + 8 > len(self.b):
                raise H5Error("not an HDF5 file: %s" % path)
        self.sb = off
        ver = self.b[off + 8]
        if ver in (0, 1):
            so, sl = self.b[off + 13], self.b[off + 14]
            if (so, sl) != (8, 8):
                raise H5Error("only 8-byte offsets/lengths are supported")
            p = off + 24 + (4 if ver == 1 else 0)
            self.base = self.u64(p)
            root_entry = p + 32
            self.root = self.u64(root_entry + 8) + self.base
        elif ver in (2, 3):
            if (self.b[off + 9], self.b[off + 10]) != (8, 8):
                raise H5Error("only 8-byte offsets/lengths are supported")
            self.base = self.u64(off + 12)
            self.root = self.u64(off + 12 + 24) + self.base
        else:
            raise H5Error("unsupported superblock version %d" % ver)
        self._cache = {}

    # -- primitives
    def u8(self, p): return self.b[p]
    def u16(self, p): return struct.unpack_from("<H", self.b, p)[0]
    def u32(self, p): return struct.unpack_from("<I", self.b, p)[0]
    def u64(self, p): return struct.unpack_from("<Q", self.b, p)[0]

    # -- object headers
    def obj(self, addr):
        if addr in self._cache:
            return self._cache[addr]
        o = _Obj()
        msgs = []
        if self.b[addr:addr + 4] == b"OHDR":
            self._read_v2_header(addr, msgs)
        else:
            self._read_v1_header(addr, msgs)
        for mtype, p, size in msgs:
            if mtype == 0x11:                                   # symbol table
                o.links = o.links or {}
                self._read_btree_group(self.u64(p) + self.base, self.u64(p + 8) + self.base, o.links)
            elif mtype == 0x06:                                 # link (new-style compact group)
                o.links = o.links or {}
                name, target = self._read_link(p)
                if target is not None:
                    o.links[name] = target
            elif mtype == 0x02:
                o.links = o.links if o.links is not None else {}
                if self.u8(p) != 0:
                    raise H5Error("unsupported link-info message")
                flags = self.u8(p + 1)
                q = p + 2 + (8 if flags & 1 else 0)
                if self.u64(q) != UNDEF:
                    raise H5Error("dense link storage (fractal heap) is not supported")
            elif mtype == 0x01:
                o.shape = self._read_dataspace(p)
            elif mtype == 0x03:
                o.dtype = self._read_datatype(p)
            elif mtype == 0x08:
                o.layout = self._read_layout(p)
            elif mtype == 0x0B:
                raise H5Error("filtered (compressed) datasets are not supported")
            elif mtype == 0x0C:
                name, val = self._read_attribute(p)
                o.attrs[name] = val
        self._cache[addr] = o
        return o

    def _read_v1_header(self, addr, msgs):
        if self.u8(addr) != 1:
            raise H5Error("unsupported object header version %d at %d" % (self.u8(addr), addr))
        nmsg = self.u16(addr + 2)
        size = self.u32(addr + 8)
        blocks = [(addr + 16, size)]
        while blocks and len(msgs) < nmsg + 64:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and nmsg > 0:
                mtype, msize, flags = self.u16(p), self.u16(p + 2), self.u8(p + 4)
                body = p + 8
                nmsg -= 1
                if mtype == 0x10:
                    blocks.append((self.u64(body) + self.base, self.u64(body + 8)))
                elif flags & 0x02:
                    raise H5Error("shared header messages are not supported")
                else:
                    msgs.append((mtype, body, msize))
                p = body + msize

    def _read_v2_header(self, addr, msgs):
        flags = self.u8(addr + 5)
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nsz = 1 << (flags & 3)
        chunk0 = int.from_bytes(self.b[p:p + nsz], "little")
        p += nsz
        blocks = [(p, chunk0)]
        track = bool(flags & 0x04)
        while blocks:
            p, left = blocks.pop(0)
            end = p + left
            while p + 4 + (2 if track else 0) <= end:
                mtype, msize, mflags = self.u8(p), self.u16(p + 1), self.u8(p + 3)
                body = p + 4 + (2 if track else 0)
                if mtype == 0x10:
                    caddr, clen = self.u64(body) + self.base, self.u64(body + 8)
                    blocks.append((caddr + 4, clen - 8))        # "OCHK" signature, trailing checksum
                elif mflags & 0x02:
                    raise H5Error("shared header messages are not supported")
                elif mtype != 0:
                    msgs.append((mtype, body, msize))
                p = body + msize

    def _read_link(self, p):
        ver, flags = self.u8(p), self.u8(p + 1)
        if ver != 1:
            raise H5Error("unsupported link message version")
        q = p + 2
        ltype = 0
        if flags & 0x08:
            ltype = self.u8(q); q += 1
        if flags & 0x04:
            q += 8
        if flags & 0x10:
            q += 1
        nsz = 1 << (flags & 3)
        nlen = int.from_bytes(self.b[q:q + nsz], "little"); q += nsz
        name = self.b[q:q + nlen].decode("utf8"); q += nlen
        if ltype != 0:
            return name, None                                   # soft / external links are ignored
        return name, self.u64(q) + self.base

    def _read_btree_group(self, btree, heap, links):
        if self.b[heap:heap + 4] != b"HEAP":
            raise H5Error("bad local heap")
        hdata = self.u64(heap + 24) + self.base

        def name_at(off):
            e = self.b.index(b"\0", hdata + off)
            return self.b[hdata + off:e].decode("utf8")

        def walk(node):
            if self.b[node:node + 4] == b"SNOD":
                n = self.u16(node + 6)
                for i in range(n):
                    e = node + 8 + i * 40
                    links[name_at(self.u64(e))] = self.u64(e + 8) + self.base
                return
            if self.b[node:node + 4] != b"TREE" or self.u8(node + 4) != 0:
                raise H5Error("bad group B-tree node")
            n = self.u16(node + 6)
            p = node + 24
            for i in range(n):
                walk(self.u64(p + 8 + i * 16) + self.base)      # key_i (8), child_i (8), ...
        walk(btree)

    def _read_dataspace(self, p):
        ver, rank = self.u8(p), self.u8(p + 1)
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if self.u8(p + 3) == 2:
                return None                                     # null dataspace
            q = p + 4
        else:
            raise H5Error("unsupported dataspace version")
        return tuple(self.u64(q + 8 * i) for i in range(rank))

    def _read_datatype(self, p):
        cls = self.u8(p) & 0x0F
        bits0 = self.u8(p + 1)
        size = self.u32(p + 4)
        if cls == 1:                                            # floating point
            if bits0 & 1:
                raise H5Error("big-endian floats are not supported")
            return np.dtype("<f%d" % size)
        if cls == 0:                                            # fixed point
            if bits0 & 1:
                raise H5Error("big-endian integers are not supported")
            return np.dtype("<%s%d" % ("i" if bits0 & 0x08 else "u", size))
        if cls == 3:                                            # fixed-length string
            return np.dtype("S%d" % size)
        if cls == 9:
            return "vlen"
        raise H5Error("unsupported datatype class %d" % cls)

    def _read_layout(self, p):
        ver = self.u8(p)
        if ver == 3:
            cls = self.u8(p + 1)
            if cls == 0:
                n = self.u16(p + 2)
                return ("compact", self.b[p + 4:p + 4 + n])
            if cls == 1:
                return ("contiguous", self.u64(p + 2), self.u64(p + 10))
            if cls == 2:
                nd = self.u8(p + 2)
                bt = self.u64(p + 3)
                dims = tuple(self.u32(p + 11 + 4 * i) for i in range(nd))
                return ("chunked", bt, dims)
        elif ver in (1, 2):
            nd, cls = self.u8(p + 1), self.u8(p + 2)
            q = p + 8
            if cls == 1:
                addr = self.u64(q)
                return ("contiguous", addr, None)
            if cls == 0:
                q += 4 * nd
                n = self.u32(q)
                return ("compact", self.b[q + 4:q + 4 + n])
        raise H5Error("unsupported data layout (version %d)" % ver)

    def _read_attribute(self, p):
        ver = self.u8(p)
        nsz, dsz, ssz = self.u16(p + 2), self.u16(p + 4), self.u16(p + 6)
        q = p + 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = self.b[q:q + nsz].split(b"\0")[0].decode("utf8"); q += pad(nsz)
        dt = self._read_datatype(q); q += pad(dsz)
        shape = self._read_dataspace(q); q += pad(ssz)
        if shape is None:
            return name, None
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if dt == "vlen":                                        # variable-length strings -> global heap
            vals = [self._read_gheap(self.u64(q + 16 * i + 4) + self.base, self.u32(q + 16 * i + 12), self.u32(q + 16 * i))
                    for i in range(n)]
            w = max([len(v) for v in vals] + [1])
            return name, np.array(vals, dtype="S%d" % w).reshape(shape)
        arr = np.frombuffer(self.b, dtype=dt, count=n, offset=q).reshape(shape)
        return name, arr.copy()

    def _read_gheap(self, coll, index, length):
        if self.b[coll:coll + 4] != b"GCOL":
            raise H5Error("bad global heap collection")
        end = coll + self.u64(coll + 8)
        p = coll + 16
        while p + 16 <= end:
            idx, size = self.u16(p), self.u64(p + 8)
            if idx == index:
                return bytes(self.b[p + 16:p + 16 + min(size, length)])
            if idx == 0:
                break
            p += 16 + ((size + 7) & ~7)
        raise H5Error("global heap object %d not found" % index)

    # -- public
    def group(self, path="/"):
        addr = self.root
        for part in [s for s in path.split("/") if s]:
            o = self.obj(addr)
            if not o.links or part not in o.links:
                raise KeyError(path)
            addr = o.links[part]
        return addr

    def dataset(self, addr):
        o = self.obj(addr)
        if o.shape is None or o.dtype is None or o.layout is None or o.dtype == "vlen":
            raise H5Error("object is not a readable dataset")
        n = int(np.prod(o.shape, dtype=np.int64)) if o.shape else 1
        kind = o.layout[0]
        if kind == "compact":
            return np.frombuffer(o.layout[1], dtype=o.dtype, count=n).reshape(o.shape).copy()
        if kind == "contiguous":
            if o.layout[1] == UNDEF:
                return np.zeros(o.shape, o.dtype)
            return np.frombuffer(self.b, dtype=o.dtype, count=n, offset=o.layout[1] + self.base).reshape(o.shape).copy()
        return self._read_chunked(o)

    def _read_chunked(self, o):
        _, bt, cdims = o.layout
        nd = len(o.shape)
        out = np.zeros(o.shape, o.dtype)
        csh = cdims[:nd]

        def walk(node):
            if self.b[node:node + 4] != b"TREE" or self.u8(node + 4) != 1:
                raise H5Error("bad chunk B-tree node")
            level, n = self.u8(node + 5), self.u16(node + 6)
            p = node + 24
            ksz = 8 + 8 * (nd + 1)
            for i in range(n):
                key = p + i * (ksz + 8)
                size, mask = self.u32(key), self.u32(key + 4)
                offs = tuple(self.u64(key + 8 + 8 * d) for d in range(nd))
                child = self.u64(key + ksz) + self.base
                if level > 0:
                    walk(child)
                else:
                    if mask:
                        raise H5Error("filtered chunks are not supported")
                    c = np.frombuffer(self.b, dtype=o.dtype, count=int(np.prod(csh)), offset=child).reshape(csh)
                    sl = tuple(slice(a, min(a + s, lim)) for a, s, lim in zip(offs, csh, o.shape))
                    out[sl] = c[tuple(slice(0, s.stop - s.start) for s in sl)]
        if bt != UNDEF:
            walk(bt + self.base)
        return out


def read_keras_weights(path):
    """-> [(layer_name, [(weight_name, ndarray), ...]), ...] in the file's layer order (layers without
    weights included with an empty list), from a Keras weights file or full-model file."""
    f = H5File(path)
    root = f.obj(f.root)
    base = "/"
    if "layer_names" not in root.attrs and root.links and "model_weights" in root.links:
        base = "/model_weights"
        root = f.obj(f.group(base))
    if root.attrs.get("layer_names") is None:
        raise H5Error("%s has no Keras 'layer_names' attribute" % path)
    out = []
    for ln in root.attrs["layer_names"].ravel():
        lname = ln.decode("utf8")
        gaddr = f.group(base + "/" + lname)
        g = f.obj(gaddr)
        names = g.attrs.get("weight_names")
        ws = []
        if names is not None:
            for wn in names.ravel():
                wname = wn.decode("utf8")
                addr = gaddr
                for part in wname.split("/"):
                    addr = f.obj(addr).links[part]
                ws.append((wname, f.dataset(addr)))
        out.append((lname, ws))
    return out


def read_keras_model_name(path):
    """Name of the model a full-model Keras file was saved from (`model_config` attribute -> config.name, e.g.
    'fcn_skip', 'res_unet', or Keras' default 'model'); None for a weights-only file."""
    import json
    f = H5File(path)
    cfg = f.obj(f.root).attrs.get("model_config")
    if cfg is None:
        return None
    if isinstance(cfg, np.ndarray):
        cfg = cfg.ravel()[0]
    if isinstance(cfg, bytes):
        cfg = cfg.decode("utf8", "replace")
    try:
        return json.loads(cfg).get("config", {}).get("name")
    except (ValueError, AttributeError):
        return None


# ------------------------------------------------------------------------------------------------
# writer (weights-only Keras file: what model.save_weights('x.h5') produces)
# ------------------------------------------------------------------------------------------------
class _Writer:
    def __init__(self):
        self.buf = bytearray(b"\0" * 96)        # superblock v0 (56 + 40-byte root entry)

    def align(self, n=8):
        while len(self.buf) % n:
            self.buf.append(0)

    def put(self, data):
        self.align()
        off = len(self.buf)
        self.buf += data
        return off

    # messages
    @staticmethod
    def _msg(mtype, body):
        body = bytes(body)
        pad = (-len(body)) % 8
        return struct.pack("<HHBBBB", mtype, len(body) + pad, 0, 0, 0, 0) + body + b"\0" * pad

    @staticmethod
    def _dataspace(shape):
        return struct.pack("<BBBBI", 1, len(shape), 0, 0, 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)

    @staticmethod
    def _datatype(dt):
        dt = np.dtype(dt)
        if dt.kind == "f" and dt.itemsize == 4:
            return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        if dt.kind == "f" and dt.itemsize == 8:
            return struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        if dt.kind == "S":
            return struct.pack("<BBBBI", 0x13, 0x00, 0x00, 0x00, dt.itemsize)   # null-terminated ASCII
        raise H5Error("cannot write dtype %s" % dt)

    def _attr(self, name, arr):
        arr = np.ascontiguousarray(arr)
        nm = name.encode("utf8") + b"\0"
        dt, ds = self._datatype(arr.dtype), self._dataspace(arr.shape)
        p8 = lambda b: b + b"\0" * ((-len(b)) % 8)
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + p8(nm) + p8(dt) + p8(ds) + arr.tobytes()
        return self._msg(0x0C, body)

    def _header(self, msgs):
        body = b"".join(msgs)
        return self.put(struct.pack("<BBHII", 1, 0, len(msgs), 1, len(body)) + b"\0" * 4 + body)

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr)
        data = self.put(arr.tobytes()) if arr.size else UNDEF
        layout = struct.pack("<BBQQ", 3, 1, data, arr.nbytes)
        fill = struct.pack("<BBBB", 2, 2, 2, 0)                 # late allocation, never written, undefined
        return self._header([self._msg(0x01, self._dataspace(arr.shape)), self._msg(0x03, self._datatype(arr.dtype)),
                             self._msg(0x05, fill), self._msg(0x08, layout)])

    def group(self, children, attrs=()):
        """children: {name: object header address}.  One leaf B-tree node + one symbol node (<= 2K entries
        would need a split; Keras groups here hold a handful), names in a local heap."""
        names = sorted(children)                                # B-tree order = strcmp order of names
        if len(names) > 32:
            # split into several symbol nodes under one B-tree node (node K = 16 -> 32 entries per SNOD)
            pass
        heap_data = bytearray(b"\0" * 8)                        # offset 0 = empty string
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += n.encode("utf8") + b"\0"
            while len(heap_data) % 8:
                heap_data.append(0)
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)                  # free block: next = 1 (none), size 16
        hd = self.put(bytes(heap_data))
        heap = self.put(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap_data), free_off, hd))
        # symbol nodes of up to 32 entries each (2 * leaf K, K = 16)
        snods = []
        for i in range(0, max(len(names), 1), 32):
            part = names[i:i + 32]
            ent = b"".join(struct.pack("<QQII16s", offs[n], children[n], 0, 0, b"") for n in part)
            ent += b"\0" * (40 * (32 - len(part)))
            snods.append((self.put(b"SNOD" + struct.pack("<BBH", 1, 0, len(part)) + ent), part))
        if len(snods) > 32:
            raise H5Error("group too large for the single-level B-tree this writer produces")
        keys = [0] + [offs[p[-1]] if p else 0 for _, p in snods]
        node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF)
        for i, (addr, _) in enumerate(snods):
            node += struct.pack("<QQ", keys[i], addr)
        node += struct.pack("<Q", keys[-1])
        node += b"\0" * (24 + (2 * 16 + 1) * 8 + 2 * 16 * 8 - len(node))   # full-size node (internal K = 16)
        bt = self.put(node)
        msgs = [self._msg(0x11, struct.pack("<QQ", bt, heap))] + [self._attr(k, v) for k, v in attrs]
        return self._header(msgs), bt, heap

    def finish(self, root_hdr, bt, heap, path):
        sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 16, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", bt, heap)
        self.buf[:len(sb)] = sb
        with open(path, "wb") as f:
            f.write(self.buf)


def write_keras_weights(path, layers, backend=b"tensorflow", keras_version=b"2.5.0"):
    """layers: [(layer_name, [(weight_name, ndarray), ...])] -> a file `model.load_weights(path)` accepts
    (and read_keras_weights reads back).  Weight names follow Keras ("conv2d/kernel:0")."""
    w = _Writer()
    layer_hdrs = {}
    for lname, ws in layers:
        # nested groups for the "scope/var:0" path below the layer group
        tree = {}
        for wname, arr in ws:
            parts = wname.split("/")
            d = tree
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = np.asarray(arr, dtype=np.float32)

        def emit(d, attrs=()):
            ch = {}
            for k, v in d.items():
                ch[k] = emit(v)[0] if isinstance(v, dict) else w.dataset(v)
            return w.group(ch, attrs)

        names = np.array([n.encode("utf8") for n, _ in ws], dtype="S") if ws else np.zeros((0,), "S1")
        layer_hdrs[lname] = emit(tree, [("weight_names", names)])[0]
    lnames = np.array([n.encode("utf8") for n, _ in layers], dtype="S")
    root, bt, heap = w.group(layer_hdrs, [("layer_names", lnames),
                                         ("backend", np.array(backend, dtype="S")),
                                         ("keras_version", np.array(keras_version, dtype="S"))])
    w.finish(root, bt, heap, path)
    return path
