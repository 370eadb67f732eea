"""Keras HDF5 weight files without h5py: a pure-Python reader (and a writer of the same subset) for the files
`model.save_weights('....h5')` / `ModelCheckpoint` produce in the reference
(mycode/given_others_gt_mean_var_seq2seq.py:484,570; FoV_seq2seq.py:108; convlstm_seq2seq.py:444).

What Keras 2.1-2.2 + h5py write (keras/engine/saving.py `save_weights_to_hdf5_group`), and all this module understands:
  * HDF5 superblock version 0, "old style" groups (symbol-table message -> v1 B-tree + local heap + symbol-table
    nodes), version-1 object headers with continuation blocks;
  * root attributes `layer_names` (fixed-length byte strings), `backend`, `keras_version`; one group per layer with
    the attribute `weight_names`; one CONTIGUOUS little-endian float32 / float64 dataset per weight, addressed by its
    name relative to the layer group (`lstm_1/kernel:0` -> nested group `lstm_1`);
  * a full-model file (`model.save`) keeps the same tree under `model_weights/`.
Not understood (raises): chunked / compressed / compact datasets, new-style (link-message) groups, superblock >= 2,
variable-length strings.  VALIDATION: the reference ships no `.h5` file and h5py is not installed on either box, but the
authoring container carries the HDF5 C library itself (1.10.6, the library h5py wraps).  The READER is pinned by
tests/golden/keras_real_weights.h5 / keras_real_model.h5 - written by that library through its C API in exactly the
Keras layout (tests/golden/make_keras_h5_real.c) - and the WRITER's output is listed and dumped correctly by that
library's own h5ls / h5dump (groups, attributes, dataset values).  Not validated against a file written by Keras itself.
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


# =====================================================================================================================
# reader
# =====================================================================================================================
class _File:
    def __init__(self, data):
        self.d = data
        self.skipped_attributes = []   # (name, reason) of attributes whose datatype is not decoded
        if data[:8] != SIGNATURE:
            raise H5Error("not an HDF5 file (signature)")
        ver = data[8]
        if ver != 0:
            raise H5Error("HDF5 superblock version %d not supported (only 0: what h5py writes by default)" % ver)
        self.so, self.sl = data[13], data[14]          # size of offsets / lengths
        if self.so != 8 or self.sl != 8:
            raise H5Error("only 8-byte offsets and lengths are supported")
        self.base = self.u64(24)
        # root group symbol-table entry sits behind base, free-space, eof, driver-info addresses
        self.root = self.symbol_entry(24 + 4 * 8)

    def u8(self, o): return self.d[o]
    def u16(self, o): return struct.unpack_from("<H", self.d, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.d, o)[0]
    def u64(self, o): return struct.unpack_from("<Q", self.d, o)[0]

    def symbol_entry(self, o):
        """-> dict(name_off, header, cache, btree, heap)"""
        e = {"name_off": self.u64(o), "header": self.u64(o + 8), "cache": self.u32(o + 16), "btree": None, "heap": None}
        if e["cache"] == 1:
            e["btree"], e["heap"] = self.u64(o + 24), self.u64(o + 32)
        return e

    # ---- object headers (version 1) ----
    def messages(self, addr):
        """[(type, flags, offset of the data, size)] of the object header at addr, continuation blocks followed."""
        a = self.base + addr
        if self.u8(a) != 1:
            raise H5Error("object header version %d not supported (only 1)" % self.u8(a))
        nmsg, size = self.u16(a + 2), self.u32(a + 8)
        blocks = [(a + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            o, left = blocks.pop(0)
            while left >= 8 and len(out) < nmsg:
                mtype, msize, flags = self.u16(o), self.u16(o + 2), self.u8(o + 4)
                body = o + 8
                if mtype == 0x0010:                      # continuation: offset, length
                    blocks.append((self.base + self.u64(body), self.u64(body + 8)))
                out.append((mtype, flags, body, msize))
                o += 8 + msize
                left -= 8 + msize
        return out

    # ---- groups ----
    def heap_string(self, heap_addr, off):
        h = self.base + heap_addr
        if self.d[h:h + 4] != b"HEAP":
            raise H5Error("local heap signature")
        data = self.base + self.u64(h + 24)
        end = self.d.index(b"\x00", data + off)
        return self.d[data + off:end].decode("utf-8")

    def group_entries(self, btree, heap):
        """name -> symbol-table entry of every member of an old-style group."""
        out = {}

        def node(addr):
            a = self.base + addr
            if self.d[a:a + 4] != b"TREE":
                raise H5Error("B-tree signature")
            ntype, level, used = self.u8(a + 4), self.u8(a + 5), self.u16(a + 6)
            if ntype != 0:
                raise H5Error("B-tree node type %d in a group" % ntype)
            o = a + 8 + 16                                # behind the sibling addresses: key0 child0 key1 ...
            for i in range(used):
                child = self.u64(o + 8 + i * 16)
                if level > 0:
                    node(child)
                else:
                    snod(child)

        def snod(addr):
            a = self.base + addr
            if self.d[a:a + 4] != b"SNOD":
                raise H5Error("symbol-table node signature")
            n = self.u16(a + 6)
            for i in range(n):
                e = self.symbol_entry(a + 8 + 40 * i)
                out[self.heap_string(heap, e["name_off"])] = e

        node(btree)
        return out

    def open_group(self, entry):
        """entry -> (btree, heap) of a group object (from the cached scratch pad or its symbol-table message)."""
        if entry.get("btree") is not None:
            return entry["btree"], entry["heap"]
        for mtype, _, body, _ in self.messages(entry["header"]):
            if mtype == 0x0011:
                return self.u64(body), self.u64(body + 8)
            if mtype in (0x0002, 0x0006):
                raise H5Error("new-style (link message) groups are not supported")
        raise H5Error("object is not a group")

    def lookup(self, entry, path):
        for part in [p for p in path.split("/") if p]:
            members = self.group_entries(*self.open_group(entry))
            if part not in members:
                raise KeyError("%r not found (members: %s)" % (part, sorted(members)))
            entry = members[part]
        return entry

    # ---- datatype / dataspace ----
    def datatype(self, o):
        """-> (kind, size): kind 'f' float, 'S' fixed string, 'i'/'u' integer"""
        cv = self.u8(o)
        cls, bits0, size = cv & 0x0F, self.u8(o + 1), self.u32(o + 4)
        if cls == 1:
            if bits0 & 1:
                raise H5Error("big-endian floats are not supported")
            return "f", size
        if cls == 0:
            if bits0 & 1:
                raise H5Error("big-endian integers are not supported")
            return ("i" if bits0 & 8 else "u"), size
        if cls == 3:
            return "S", size
        if cls == 9:
            # variable-length: h5py stores a Python bytes / str SCALAR this way (Keras: backend, keras_version, model_config,
            # training_config); each element is a 16-byte global-heap reference {length u32, collection address u64, index u32}
            if (bits0 & 0x0F) != 1:
                raise H5Error("variable-length sequences (non-string) are not supported")
            return "V", 16
        raise H5Error("datatype class %d not supported" % cls)

    def dataspace(self, o):
        ver, rank, flags = self.u8(o), self.u8(o + 1), self.u8(o + 2)
        if ver == 1:
            start = o + 8
        elif ver == 2:
            if self.u8(o + 3) == 2:      # null dataspace
                return (0,)
            start = o + 4
        else:
            raise H5Error("dataspace version %d" % ver)
        _ = flags
        return tuple(self.u64(start + 8 * i) for i in range(rank))

    def global_heap_object(self, addr, index):
        """Bytes of object `index` of the global heap collection at `addr` (HDF5 file format III.E)."""
        if self.d[addr:addr + 4] != b"GCOL":
            raise H5Error("bad global heap collection signature")
        size = self.u64(addr + 8)
        o, end = addr + 16, addr + size
        while o + 16 <= end:
            idx, osz = self.u16(o), self.u64(o + 8)
            if idx == 0:      # free space: end of the used part
                break
            if idx == index:
                return bytes(self.d[o + 16:o + 16 + osz])
            o += 16 + ((osz + 7) & ~7)
        raise H5Error("global heap object %d not found" % index)

    def _array(self, kind, size, shape, raw_off):
        n = int(np.prod(shape)) if shape else 1
        if kind == "V":
            vals = []
            for i in range(n):
                o = raw_off + 16 * i
                ln, addr, idx = self.u32(o), self.u64(o + 4), self.u32(o + 12)
                vals.append(b"" if (ln == 0 or addr == 0) else self.global_heap_object(addr, idx)[:ln])
            return np.array(vals, dtype=object).reshape(shape)
        if kind == "S":
            raw = self.d[raw_off:raw_off + n * size]
            vals = [raw[i * size:(i + 1) * size].split(b"\x00", 1)[0] for i in range(n)]
            return np.array(vals, dtype="S%d" % max(size, 1)).reshape(shape)
        dt = np.dtype("<%s%d" % (kind, size))
        return np.frombuffer(self.d, dtype=dt, count=n, offset=raw_off).reshape(shape).copy()

    # ---- attributes and datasets ----
    def attributes(self, entry):
        """name -> array.  An attribute whose datatype this reader does not decode (compound, non-string variable length,
        ...) maps to None instead of failing the whole object: Keras files carry attributes the weights do not need."""
        out = {}
        for mtype, _, body, _ in self.messages(entry["header"]):
            if mtype != 0x000C:
                continue
            try:
                name, val = self._attribute(body)
            except H5Error as e:
                name, val = self._attribute_name(body), None
                self.skipped_attributes.append((name, str(e)))
            out[name] = val
        return out

    def _attribute_name(self, body):
        ver, nsz = self.u8(body), self.u16(body + 2)
        o = body + 8 + (1 if ver == 3 else 0)
        return self.d[o:o + nsz].split(b"\x00", 1)[0].decode("utf-8", "replace")

    def _attribute(self, body):
        ver = self.u8(body)
        nsz, tsz, ssz = self.u16(body + 2), self.u16(body + 4), self.u16(body + 6)
        if ver == 1:
            pad = lambda v: (v + 7) & ~7
            o = body + 8
            name = self.d[o:o + nsz].split(b"\x00", 1)[0].decode("utf-8")
            o += pad(nsz)
            kind, size = self.datatype(o)
            o += pad(tsz)
            shape = self.dataspace(o)
            o += pad(ssz)
        elif ver in (2, 3):
            o = body + 8 + (1 if ver == 3 else 0)
            name = self.d[o:o + nsz].split(b"\x00", 1)[0].decode("utf-8")
            o += nsz
            kind, size = self.datatype(o)
            o += tsz
            shape = self.dataspace(o)
            o += ssz
        else:
            raise H5Error("attribute message version %d" % ver)
        return name, self._array(kind, size, shape, o)

    def dataset(self, entry):
        kind = size = shape = addr = None
        for mtype, _, body, _ in self.messages(entry["header"]):
            if mtype == 0x0001:
                shape = self.dataspace(body)
            elif mtype == 0x0003:
                kind, size = self.datatype(body)
            elif mtype == 0x0008:
                ver = self.u8(body)
                if ver == 3:
                    cls = self.u8(body + 1)
                    if cls != 1:
                        raise H5Error("only contiguous datasets are supported (layout class %d: %s)"
                                      % (cls, {0: "compact", 2: "chunked"}.get(cls, "?")))
                    addr = self.u64(body + 2)
                elif ver in (1, 2):
                    if self.u8(body + 2) != 1:
                        raise H5Error("only contiguous datasets are supported")
                    addr = self.u64(body + 8)
                else:
                    raise H5Error("data layout version %d" % ver)
            elif mtype == 0x000B:
                raise H5Error("filtered (compressed) datasets are not supported")
        if None in (kind, size, shape, addr):
            raise H5Error("object is not a simple dataset")
        if addr == UNDEF:
            return np.zeros(shape, dtype="<%s%d" % (kind, size))
        return self._array(kind, size, shape, self.base + addr)


def _decode(a):
    return [x.decode("utf-8") if isinstance(x, bytes) else str(x) for x in np.asarray(a).ravel()]


def read_keras_layers(path):
    """-> [(layer_name, [(weight_name, ndarray), ...]), ...] in the file's own order (layers without weights included)."""
    with open(path, "rb") as fh:
        f = _File(fh.read())
    root = f.root
    attrs = f.attributes(root)
    if "layer_names" not in attrs:                     # model.save(): the weights live under model_weights/
        try:
            root = f.lookup(root, "model_weights")
        except KeyError:
            raise H5Error("no layer_names attribute and no model_weights group: not a Keras weight file")
        attrs = f.attributes(root)
    layers = []
    for lname in _decode(attrs["layer_names"]):
        g = f.lookup(root, lname)
        wnames = _decode(f.attributes(g).get("weight_names", np.array([], dtype="S1")))
        layers.append((lname, [(wn, f.dataset(f.lookup(g, wn))) for wn in wnames]))
    return layers


def read_keras_weights(path, expected_shapes=None):
    """Flat weight list of a Keras weight file in Keras's own order (layer by layer: kernel, recurrent_kernel, bias /
    kernel, bias) as float32 arrays.  `expected_shapes`: the receiving model's shapes - checked one to one, so a file
    of a different topology is refused instead of being loaded into the wrong tensors."""
    flat = [np.asarray(a, dtype=np.float32) for _, ws in read_keras_layers(path) for _, a in ws]
    if expected_shapes is not None:
        got = [tuple(a.shape) for a in flat]
        want = [tuple(s) for s in expected_shapes]
        if got != want:
            raise ValueError("weight file does not match the model: file has %s, model expects %s" % (got, want))
    return flat


# =====================================================================================================================
# writer (the same subset: what save_weights('x.h5', h5=True) and the test fixtures use)
# =====================================================================================================================
class _Writer:
    def __init__(self):
        self.buf = bytearray()

    def alloc(self, n, align=8):
        while len(self.buf) % align:
            self.buf.append(0)
        off = len(self.buf)
        self.buf.extend(b"\x00" * n)
        return off

    def put(self, off, data):
        self.buf[off:off + len(data)] = data


def _dt_float(size):
    if size == 4:
        props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + props
    props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    return struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0x00, 8) + props


def _dt_string(size):
    return struct.pack("<BBBBI", 0x13, 0x00, 0x00, 0x00, size)     # class 3, null-terminated, ASCII


def _dataspace(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)


def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _message(mtype, body):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _attribute(name, arr):
    arr = np.asarray(arr)
    nm = name.encode("utf-8") + b"\x00"
    if arr.dtype.kind == "S":
        dt, raw = _dt_string(arr.dtype.itemsize), arr.tobytes()
    else:
        arr = arr.astype("<f4")
        dt, raw = _dt_float(4), arr.tobytes()
    ds = _dataspace(arr.shape)
    body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + raw
    return _message(0x000C, body)


def _object_header(w, messages):
    body = b"".join(messages)
    off = w.alloc(16 + len(body))
    w.put(off, struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body)
    return off


INTERNAL_K = 16      # superblock: group internal node K (B-tree nodes are allocated at their full size)


def _write_group(w, members, attrs, leaf_k):
    """members: name -> object header address.  -> header address of the group object (old-style symbol table).
    Nodes are written at the full size their K implies - the library reads whole nodes."""
    names = sorted(members)                      # symbol-table entries are ordered by name
    heap_data = bytearray(b"\x00" * 8)           # offset 0: the empty string (key 0 of the B-tree)
    name_off = {}
    for nme in names:
        name_off[nme] = len(heap_data)
        heap_data.extend(_pad8(nme.encode("utf-8") + b"\x00"))
    free_off = len(heap_data)
    heap_data.extend(struct.pack("<QQ", 1, 16))   # one free block: next = 1 (none), size 16
    data_addr = w.alloc(len(heap_data))
    w.put(data_addr, bytes(heap_data))
    heap = w.alloc(32)
    w.put(heap, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, data_addr))
    # one symbol-table node (capacity 2 * leaf K; K is set so that everything fits one node - see write_keras_layers)
    assert len(names) <= 2 * leaf_k
    snod = w.alloc(8 + 40 * 2 * leaf_k)
    ent = b"".join(struct.pack("<QQII16x", name_off[nme], members[nme], 0, 0) for nme in names)
    w.put(snod, b"SNOD" + struct.pack("<BxH", 1, len(names)) + ent)
    btree = w.alloc(24 + 2 * INTERNAL_K * 16 + 8)
    last = name_off[names[-1]] if names else 0
    w.put(btree, b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod, last))
    msgs = [_message(0x0011, struct.pack("<QQ", btree, heap))] + [_attribute(k, v) for k, v in attrs.items()]
    return _object_header(w, msgs), btree, heap


def _write_dataset(w, arr):
    arr = np.ascontiguousarray(arr)
    if arr.dtype not in (np.dtype("<f4"), np.dtype("<f8")):
        arr = arr.astype("<f4")
    raw = w.alloc(arr.nbytes)
    w.put(raw, arr.tobytes())
    msgs = [_message(0x0001, _dataspace(arr.shape)), _message(0x0003, _dt_float(arr.dtype.itemsize)),
            _message(0x0008, struct.pack("<BBQQ", 3, 1, raw, arr.nbytes))]
    return _object_header(w, msgs)


def write_keras_layers(path, layers, backend="tensorflow", keras_version="2.2.4"):
    """layers: [(layer_name, [(weight_name, ndarray), ...]), ...] -> a Keras-layout HDF5 weight file (see module doc)."""
    w = _Writer()
    w.alloc(8 + 16 + 4 * 8 + 40)                    # superblock v0 with the root symbol-table entry
    # every group fits ONE symbol-table node: leaf K = half the largest member count (at least the library's default 4)
    max_members = max([len(layers)] + [len(ws) for _, ws in layers] + [1])
    leaf_k = max(4, (max_members + 1) // 2)
    layer_addr = {}
    for lname, ws in layers:
        # weight names are paths relative to the layer group: build the nested groups
        tree = {}
        for wn, arr in ws:
            parts = wn.split("/")
            node = tree
            for prt in parts[:-1]:
                node = node.setdefault(prt, {})
            node[parts[-1]] = _write_dataset(w, arr)

        def emit(node, attrs):
            members = {k: (emit(v, {}) if isinstance(v, dict) else v) for k, v in node.items()}
            return _write_group(w, members, attrs, leaf_k)[0]

        names = np.array([wn.encode("utf-8") for wn, _ in ws], dtype="S") if ws else np.zeros((0,), dtype="S1")
        layer_addr[lname] = emit(tree, {"weight_names": names})
    root_attrs = {"layer_names": np.array([n.encode("utf-8") for n, _ in layers], dtype="S"),
                  "backend": np.array(backend.encode("utf-8"), dtype="S"),
                  "keras_version": np.array(keras_version.encode("utf-8"), dtype="S")}
    root, btree, heap = _write_group(w, layer_addr, root_attrs, leaf_k)
    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, leaf_k, INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, len(w.buf), UNDEF)
    sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", btree, heap)
    w.put(0, sb)
    with open(path, "wb") as fh:
        fh.write(bytes(w.buf))
