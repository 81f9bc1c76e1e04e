"""Minimal HDF5 writer for test files (classic structures only: superblock v0, version-1 object headers, symbol-table
groups with v1 B-tree / SNOD / local heap, contiguous or chunked datasets -- chunks indexed by a version-1 chunk B-tree
of one or two levels, optionally through the gzip / shuffle / fletcher32 filters h5py offers -- fixed- and
variable-length string attributes).

Test infrastructure: it produces Keras-shaped `converted.hdf5` files for the dependency-free reader in
ipu_path_trace_amd/host/Hdf5Reader.cpp (there is no h5py here).  The reader is additionally checked against a file
written by the real HDF5 library (scipy's MATLAB v7.3 fixture), see tests/test_hdf5.py.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b):
    return b + b"\0" * ((-len(b)) % 8)


class H5Writer:
    def __init__(self, user_block=0, snod_capacity=8):
        self.buf = bytearray(b"\0" * user_block)
        self.base = user_block
        self.snod_capacity = snod_capacity
        self.sb_at = len(self.buf)
        self.buf += b"\0" * 96                      # superblock placeholder (24 + 4*8 + 40 = 96)
        self.gheap = []                              # (index, bytes) of variable-length strings
        self.gheap_addr = None

    # ---- low level
    def _alloc(self, data):
        self.buf += b"\0" * ((-len(self.buf)) % 8)
        at = len(self.buf) - self.base
        self.buf += data
        return at

    def _patch(self, at, data):
        self.buf[self.base + at: self.base + at + len(data)] = data

    @staticmethod
    def _msg(mtype, data, flags=0):
        data = _pad8(data)
        return struct.pack("<HHB3x", mtype, len(data), flags) + data

    def _object_header(self, messages):
        body = b"".join(messages)
        return self._alloc(struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body)

    # ---- messages
    @staticmethod
    def _dataspace(shape):
        return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", s) for s in shape)

    @staticmethod
    def _float_type(size):
        sign, exploc, expsize, mansize, bias = {2: (15, 10, 5, 10, 15), 4: (31, 23, 8, 23, 127), 8: (63, 52, 11, 52, 1023)}[size]
        return struct.pack("<BBBBI", 0x11, 0x20, sign, 0, size) + struct.pack("<HHBBBBI", 0, 8 * size, exploc, expsize, 0, mansize, bias)

    @staticmethod
    def _string_type(n):
        return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, n)       # class 3, null-terminated, ASCII

    def _attribute(self, name, value, vlen=False):
        nm = name.encode() + b"\0"
        val = value.encode() if isinstance(value, str) else value
        if vlen:
            idx = len(self.gheap) + 1
            self.gheap.append((idx, val))
            dt = struct.pack("<BBBBI", 0x19, 0x01, 0, 0, 16) + self._string_type(1)   # vlen of 1-byte strings
            data = struct.pack("<IQI", len(val), 0, idx)                                # collection address patched later
            self._vlen_refs = getattr(self, "_vlen_refs", [])
        else:
            dt = self._string_type(len(val) + 1)
            data = val + b"\0"
        ds = struct.pack("<BBB5x", 1, 0, 0)                      # scalar
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data
        return self._msg(0x000C, body), (len(struct.pack("<BBHHH", 1, 0, 0, 0, 0)) + len(_pad8(nm)) + len(_pad8(dt)) + len(_pad8(ds)) + 4 if vlen else None)

    # ---- objects
    def dataset(self, array, attrs=None, chunks=None, compression=None, shuffle=False, fletcher32=False, filter_ids=None,
                layout_version=3, leaf_fan=None, skip_chunks=(), fletcher_first=False, cyclic_index=False):
        """chunks: chunk shape -> chunked layout (edge chunks stored full size, as libhdf5 does).  compression="gzip",
        shuffle, fletcher32: the filter pipeline h5py writes for them, in h5py's order (shuffle, gzip, fletcher32).
        filter_ids: extra filter ids appended to the pipeline message WITHOUT being applied (for rejection tests).
        leaf_fan: entries per B-tree leaf (a second tree level appears when there are more chunks).  skip_chunks: chunk
        indices left unwritten (they read back as zeros).  fletcher_first: the checksum filter at the head of the pipeline
        (h5repack -f FLET -f GZIP order) instead of h5py's tail.  cyclic_index: a MALFORMED chunk index whose internal node
        points at itself through every entry (the reader must refuse it quickly)."""
        a = np.ascontiguousarray(array)
        es = a.dtype.itemsize
        msgs = [self._msg(0x0001, self._dataspace(a.shape)), self._msg(0x0003, self._float_type(es), flags=1)]
        if chunks is None:
            raw = self._alloc(a.tobytes())
            msgs.append(self._msg(0x0008, struct.pack("<BBQQ", 3, 1, raw, a.nbytes)))
            return self._finish_object(msgs, attrs)
        import itertools
        import zlib
        filters = []                                  # (id, client values)
        if fletcher32 and fletcher_first:
            filters.append((3, []))
        if shuffle:
            filters.append((2, [es]))
        if compression == "gzip":
            filters.append((1, [4]))
        if fletcher32 and not fletcher_first:
            filters.append((3, []))
        entries = []                                  # (offset tuple, address, stored bytes)
        grid = [range(0, s, c) for s, c in zip(a.shape, chunks)]
        for ci, off in enumerate(itertools.product(*grid)):
            if ci in skip_chunks:
                continue
            block = np.zeros(chunks, dtype=a.dtype)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, chunks, a.shape))
            block[tuple(slice(0, x.stop - x.start) for x in sl)] = a[sl]
            data = block.tobytes()
            for fid, vals in filters:
                if fid == 2:
                    data = np.frombuffer(data, dtype=np.uint8).reshape(-1, es).T.tobytes()
                elif fid == 1:
                    data = zlib.compress(data, vals[0])
                elif fid == 3:
                    data = data + b"\0\0\0\0"
            entries.append((off, self._alloc(data), len(data)))
        nd = len(chunks) + 1

        def key(off, nbytes):
            return struct.pack("<II", nbytes, 0) + b"".join(struct.pack("<Q", o) for o in off) + struct.pack("<Q", 0)

        def node(level, items):                        # items: (first offset, child address, stored bytes)
            body = b"TREE" + struct.pack("<BBHQQ", 1, level, len(items), UNDEF, UNDEF)
            for off, child, nbytes in items:
                body += key(off, nbytes) + struct.pack("<Q", child)
            body += key(tuple(a.shape), 0)             # the final key
            return self._alloc(body)

        if cyclic_index:
            fan = 64
            at = self._alloc(b"\0" * (24 + fan * (len(key(entries[0][0], 0)) + 8) + len(key(entries[0][0], 0))))
            body = b"TREE" + struct.pack("<BBHQQ", 1, 3, fan, UNDEF, UNDEF)
            for _ in range(fan):
                body += key(entries[0][0], 0) + struct.pack("<Q", at)
            body += key(tuple(a.shape), 0)
            self._patch(at, body)
            tree = at
        elif not entries:
            tree = UNDEF
        elif leaf_fan and len(entries) > leaf_fan:
            leaves = [entries[i:i + leaf_fan] for i in range(0, len(entries), leaf_fan)]
            tree = node(1, [(part[0][0], node(0, part), 0) for part in leaves])
        else:
            tree = node(0, entries)
        dims = b"".join(struct.pack("<I", c) for c in chunks) + struct.pack("<I", es)
        if layout_version == 3:
            msgs.append(self._msg(0x0008, struct.pack("<BBB", 3, 2, nd) + struct.pack("<Q", tree) + dims))
        else:
            msgs.append(self._msg(0x0008, struct.pack("<BBB5x", layout_version, nd, 2) + struct.pack("<Q", tree) + dims))
        ids = filters + [(i, []) for i in (filter_ids or [])]
        if ids:
            body = struct.pack("<BB6x", 1, len(ids))
            for fid, vals in ids:
                name = b"" if fid < 256 else b"thirdparty\0"
                body += struct.pack("<HHHH", fid, len(_pad8(name)), 1, len(vals)) + _pad8(name)
                body += b"".join(struct.pack("<I", v) for v in vals) + (b"\0\0\0\0" if len(vals) & 1 else b"")
            msgs.append(self._msg(0x000B, body))
        return self._finish_object(msgs, attrs)

    def _finish_object(self, msgs, attrs):
        fixups = []
        for name, value in (attrs or {}).items():
            vlen = isinstance(value, tuple)
            m, off = self._attribute(name, value[0] if vlen else value, vlen=vlen)
            if vlen:
                fixups.append((sum(len(x) for x in msgs) + 8 + off, None))
            msgs.append(m)
        at = self._object_header(msgs)
        for off, _ in fixups:
            self._pending = getattr(self, "_pending", [])
            self._pending.append(at + 16 + off)     # position of the 8-byte collection address inside the header
        return at

    def group(self, children, attrs=None):
        """children: dict name -> object header address."""
        names = sorted(children)
        heap_data = bytearray(b"\0" * 8)
        offsets = {}
        for n in names:
            offsets[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b"\0")
        data_at = self._alloc(bytes(heap_data))
        heap_at = self._alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), UNDEF, data_at))
        # symbol table nodes of at most snod_capacity entries, one B-tree leaf level
        snods = []
        for i in range(0, max(len(names), 1), self.snod_capacity):
            part = names[i:i + self.snod_capacity]
            body = b"SNOD" + struct.pack("<BBH", 1, 0, len(part))
            for n in part:
                body += struct.pack("<QQII16x", offsets[n], children[n], 0, 0)
            snods.append((self._alloc(body), offsets[part[-1]] if part else 0))
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF) + struct.pack("<Q", 0)
        for at, last_key in snods:
            tree += struct.pack("<QQ", at, last_key)
        tree_at = self._alloc(tree)
        msgs = [self._msg(0x0011, struct.pack("<QQ", tree_at, heap_at))]
        at = self._finish_object(msgs, attrs)
        self._last_group = (tree_at, heap_at)
        return at

    def finish(self, root_header):
        tree_at, heap_at = self._last_group
        if self.gheap:
            body = b""
            for idx, val in self.gheap:
                body += struct.pack("<HH4xQ", idx, 1, len(val)) + _pad8(val)
            body += struct.pack("<HH4xQ", 0, 0, 0)
            size = 16 + len(body)
            self.gheap_addr = self._alloc(b"GCOL" + struct.pack("<B3xQ", 1, size) + body)
            for pos in getattr(self, "_pending", []):
                self._patch(pos, struct.pack("<Q", self.gheap_addr))
        eof = len(self.buf) - self.base
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
        sb += struct.pack("<QQQQ", 0 if self.base == 0 else self.base, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_header, 1, 0) + struct.pack("<QQ", tree_at, heap_at)
        assert len(sb) == 96
        self.buf[self.sb_at:self.sb_at + 96] = sb
        return bytes(self.buf)


def write_keras_h5(path, layers, vlen_config=False, user_block=0, snod_capacity=8, with_concat=True, **dataset_options):
    """Keras "Functional" H5 of a NIF: layers = [(kernel [in,out], bias | None, relu)], dataset paths
    /model_weights/<name>/<name>/{kernel:0,bias:0} (reference src/keras/Hdf5Model.cpp:71-82)."""
    import json
    w = H5Writer(user_block=user_block, snod_capacity=snod_capacity)
    names = ["dense" if i == 0 else "dense_%d" % i for i in range(len(layers))]
    cfg_layers = [{"class_name": "InputLayer", "config": {"name": "input_1", "dtype": "float32"}}]
    layer_groups = {}
    for name, (k, b, relu) in zip(names, layers):
        if with_concat and k.shape[0] not in (layers[0][0].shape[0], layers[0][0].shape[1]):
            cfg_layers.append({"class_name": "Concatenate", "config": {"name": "concatenate", "axis": -1}})
        dt = {"float16": "float16", "float32": "float32"}[str(k.dtype)]
        cfg_layers.append({"class_name": "Dense", "config": {"name": name, "dtype": dt, "units": int(k.shape[1]),
                                                              "activation": "relu" if relu else "linear",
                                                              "use_bias": b is not None}})
        opts = dict(dataset_options)
        if "chunks" in opts:                           # (rows, cols) for the kernels; the biases take the column count
            opts["chunks"] = tuple(min(c, s) for c, s in zip(opts["chunks"], k.shape))
        inner = {"kernel:0": w.dataset(k, **opts)}
        if b is not None:
            if "chunks" in opts:
                opts["chunks"] = (min(dataset_options["chunks"][1], b.shape[0]),)
            inner["bias:0"] = w.dataset(b, **opts)
        layer_groups[name] = w.group({name: w.group(inner)}, attrs={"weight_names": name})
    config = json.dumps({"class_name": "Functional", "config": {"name": "model", "layers": cfg_layers}})
    weights = w.group(layer_groups, attrs={"backend": "tensorflow", "keras_version": "2.8.0"})
    root = w.group({"model_weights": weights},
                   attrs={"keras_version": "2.8.0", "backend": "tensorflow",
                          "model_config": (config,) if vlen_config else config})
    with open(path, "wb") as f:
        f.write(w.finish(root))
