"""A small reader -- and, at the end of the file, writer -- for Keras HDF5 weight files (``model.save_weights('x.h5')``),
in pure Python + NumPy.

The reference loads its models through Keras (``model.load_weights(path.h5)``: api.py:87,
networks/inceptionv3.py:79-82), so a drop-in has to read those files; ``h5py`` is used when it is importable, and this
module when it is not (the MI355X image has no h5py in its interpreter).  It implements the part of the HDF5 file
format that h5py writes for such files with its default settings -- and nothing else:

  * superblock version 0 or 1, 8-byte offsets and lengths;
  * version-1 object headers (with continuation blocks);
  * old-style groups: symbol-table message -> version-1 B-tree of symbol-table nodes + local heap;
  * datasets with contiguous or compact layout (Keras writes neither chunks nor filters), little-endian IEEE floats
    and fixed-point integers;
  * attributes (message versions 1-3) holding fixed-length strings, variable-length strings (global heap) or numbers --
    Keras keeps ``layer_names`` / ``weight_names`` there.

Anything outside that (new-style groups, chunked or filtered datasets, big-endian data) raises ``H5Error`` naming the
feature, rather than guessing.  tests/test_h5lite.py checks it against files written by the real HDF5 library
(h5py 3.3 / libhdf5 1.10 under /opt/conda, where present) and against a committed fixture written the same way.
"""
import struct

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


def _u(buf, off, n):
    return int.from_bytes(buf[off:off + n], 'little')


class _Datatype:
    def __init__(self, buf, off):
        cv = buf[off]
        self.cls, self.version = cv & 0x0F, cv >> 4
        self.bits = buf[off + 1:off + 4]
        self.size = _u(buf, off + 4, 4)
        self.base = None
        if self.cls == 9:                                   # variable length: base type follows
            self.base = _Datatype(buf, off + 8)
            self.is_vlen_string = (self.bits[0] & 0x0F) == 1
        if self.cls in (0, 1) and (self.bits[0] & 1):
            raise H5Error('big-endian data is not supported')

    def numpy_dtype(self):
        if self.cls == 1:
            return np.dtype({2: '<f2', 4: '<f4', 8: '<f8'}[self.size])
        if self.cls == 0:
            signed = bool(self.bits[0] & 0x08)
            return np.dtype('<%s%d' % ('i' if signed else 'u', self.size))
        if self.cls == 3:
            return np.dtype('S%d' % self.size)
        raise H5Error('datatype class %d is not supported' % self.cls)


def _dataspace(buf, off):
    version, rank, flags = buf[off], buf[off + 1], buf[off + 2]
    if version == 1:
        p = off + 8
    elif version == 2:
        if buf[off + 3] == 2:                               # null dataspace
            return None
        p = off + 4
    else:
        raise H5Error('dataspace message version %d' % version)
    return tuple(_u(buf, p + 8 * i, 8) for i in range(rank))


class H5File:
    """``H5File(path)``; ``.root`` is a group: ``group.keys()``, ``group[name]`` (sub-group or ndarray),
    ``group.attrs`` (dict)."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            self.buf = fh.read()
        base = 0
        while self.buf[base:base + 8] != SIGNATURE:
            base = 512 if base == 0 else base * 2
            if base >= len(self.buf):
                raise H5Error('not an HDF5 file: %s' % path)
        b = self.buf
        version = b[base + 8]
        if version not in (0, 1):
            raise H5Error('superblock version %d (written with libver="latest"?) is not supported; '
                          'h5py and Keras write version 0 by default' % version)
        if b[base + 13] != 8 or b[base + 14] != 8:
            raise H5Error('only 8-byte offsets and lengths are supported')
        p = base + 24 + (4 if version == 1 else 0)
        self.base_address = _u(b, p, 8)
        root_entry = p + 32                                  # base, free-space, end-of-file, driver-info addresses
        self.root = _Group(self, _u(b, root_entry + 8, 8))
        self._gcol = {}

    # ---- object headers -------------------------------------------------------------------------------------
    def messages(self, addr):
        b = self.buf
        addr += self.base_address
        if b[addr:addr + 4] == b'OHDR':
            raise H5Error('version-2 object headers (libver="latest") are not supported')
        if b[addr] != 1:
            raise H5Error('object header version %d at %d' % (b[addr], addr))
        nmsgs = _u(b, addr + 2, 2)
        blocks = [(addr + 16, _u(b, addr + 8, 4))]
        out = []
        while blocks and len(out) < nmsgs:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < nmsgs:
                mtype, msize, mflags = _u(b, p, 2), _u(b, p + 2, 2), b[p + 4]
                body = p + 8
                if mflags & 2:
                    raise H5Error('shared object-header messages are not supported')
                if mtype == 0x10:                            # continuation
                    blocks.append((self.base_address + _u(b, body, 8), _u(b, body + 8, 8)))
                out.append((mtype, body, msize))
                p = body + msize
        return out

    # ---- heaps ----------------------------------------------------------------------------------------------
    def local_heap_data(self, addr):
        b = self.buf
        addr += self.base_address
        if b[addr:addr + 4] != b'HEAP':
            raise H5Error('local heap signature missing at %d' % addr)
        return self.base_address + _u(b, addr + 24, 8)

    def global_heap_object(self, collection, index):
        b = self.buf
        if collection not in self._gcol:
            a = collection + self.base_address
            if b[a:a + 4] != b'GCOL':
                raise H5Error('global heap signature missing at %d' % a)
            size = _u(b, a + 8, 8)
            objs, p = {}, a + 16
            while p + 16 <= a + size:
                idx, osize = _u(b, p, 2), _u(b, p + 8, 8)
                if idx == 0:
                    break
                objs[idx] = b[p + 16:p + 16 + osize]
                p += 16 + (osize + 7) // 8 * 8
            self._gcol[collection] = objs
        return self._gcol[collection][index]

    # ---- attribute / dataset payloads -----------------------------------------------------------------------
    def decode(self, dtype, shape, raw_off):
        b = self.buf
        n = int(np.prod(shape)) if shape else 1
        if dtype.cls == 9:
            if not dtype.is_vlen_string:
                raise H5Error('variable-length sequences are not supported')
            vals = []
            for i in range(n):
                p = raw_off + 16 * i
                vals.append(self.global_heap_object(_u(b, p + 4, 8), _u(b, p + 12, 4)))
            arr = np.array(vals, dtype=object)
            return arr.reshape(shape) if shape else arr[0]
        dt = dtype.numpy_dtype()
        arr = np.frombuffer(b, dtype=dt, count=n, offset=raw_off)
        return arr.reshape(shape).copy() if shape else arr[0]

    def attribute(self, body):
        b = self.buf
        version = b[body]
        nsize, tsize, ssize = _u(b, body + 2, 2), _u(b, body + 4, 2), _u(b, body + 6, 2)
        p = body + 8 + (1 if version == 3 else 0)
        pad = (lambda v: (v + 7) // 8 * 8) if version == 1 else (lambda v: v)
        if version not in (1, 2, 3):
            raise H5Error('attribute message version %d' % version)
        name = b[p:p + nsize].split(b'\x00', 1)[0].decode('utf-8')
        p += pad(nsize)
        dtype = _Datatype(b, p)
        p += pad(tsize)
        shape = _dataspace(b, p)
        p += pad(ssize)
        return name, (None if shape is None else self.decode(dtype, shape, p))


class _Group:
    def __init__(self, f, header_addr):
        self.file = f
        self._links = None
        self._attrs = None
        self._header = header_addr

    def _scan(self):
        f, b = self.file, self.file.buf
        self._links, self._attrs = {}, {}
        btree = heap = None
        for mtype, body, _ in f.messages(self._header):
            if mtype == 0x11:
                btree, heap = _u(b, body, 8), _u(b, body + 8, 8)
            elif mtype == 0x0C:
                k, v = f.attribute(body)
                self._attrs[k] = v
            elif mtype in (0x02, 0x06):
                raise H5Error('new-style groups (link messages) are not supported')
        if btree is None:
            return
        heap_data = f.local_heap_data(heap)

        def walk(addr):
            a = addr + f.base_address
            if b[a:a + 4] == b'TREE':
                if b[a + 4] != 0:
                    raise H5Error('unexpected B-tree node type %d in a group' % b[a + 4])
                used = _u(b, a + 6, 2)
                p = a + 24 + 8                              # first key
                for _ in range(used):
                    walk(_u(b, p, 8))
                    p += 16
            elif b[a:a + 4] == b'SNOD':
                for i in range(_u(b, a + 6, 2)):
                    e = a + 8 + 40 * i
                    s = heap_data + _u(b, e, 8)
                    name = b[s:b.index(b'\x00', s)].decode('utf-8')
                    self._links[name] = _u(b, e + 8, 8)
            else:
                raise H5Error('unknown group node at %d' % a)

        walk(btree)

    def keys(self):
        if self._links is None:
            self._scan()
        return list(self._links)

    @property
    def attrs(self):
        if self._attrs is None:
            self._scan()
        return self._attrs

    def __contains__(self, name):
        return name.split('/', 1)[0] in self.keys()

    def __getitem__(self, name):
        head, _, rest = name.partition('/')
        if head not in self.keys():
            raise KeyError(name)
        obj = self._open(self._links[head])
        return obj[rest] if rest else obj

    def _open(self, addr):
        f, b = self.file, self.file.buf
        msgs = f.messages(addr)
        types = {m[0] for m in msgs}
        if 0x11 in types or not ({0x01, 0x03, 0x08} <= types):
            return _Group(f, addr)
        dtype = shape = None
        data_off = None
        for mtype, body, _ in msgs:
            if mtype == 0x03:
                dtype = _Datatype(b, body)
            elif mtype == 0x01:
                shape = _dataspace(b, body)
            elif mtype == 0x0B:
                raise H5Error('filtered (compressed) datasets are not supported')
            elif mtype == 0x08:
                version, cls = b[body], b[body + 1]
                if version != 3:
                    raise H5Error('data layout message version %d' % version)
                if cls == 1:
                    a = _u(b, body + 2, 8)
                    data_off = None if a == UNDEF else a + f.base_address
                elif cls == 0:
                    data_off = body + 4
                else:
                    raise H5Error('chunked datasets are not supported (Keras writes contiguous ones)')
        if data_off is None:
            return np.zeros(shape or (), dtype=dtype.numpy_dtype())
        return f.decode(dtype, shape or (), data_off)


def _text(v):
    return v.decode('utf-8') if isinstance(v, (bytes, np.bytes_)) else str(v)


def read_keras_weights(path):
    """{weight name without the ':0' suffix: float32 ndarray} from a Keras ``save_weights`` file, or from the
    ``model_weights`` group of a ``model.save`` file.  Layer order and names come from the ``layer_names`` /
    ``weight_names`` attributes, exactly as Keras' own loader reads them."""
    try:
        import h5py                                          # the usual way, where it exists
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(path, 'r') as f:
            g = f['model_weights'] if 'model_weights' in f else f
            return _collect(g, lambda grp, name: np.asarray(grp[name]))
    f = H5File(path)
    g = f.root['model_weights'] if 'model_weights' in f.root else f.root
    return _collect(g, lambda grp, name: grp[name])


def _names_attr(attrs, key):
    if key in attrs:
        return [_text(v) for v in np.atleast_1d(attrs[key])]
    out, i = [], 0                                           # Keras splits attributes beyond 64 KB into key0, key1, ...
    while '%s%d' % (key, i) in attrs:
        out += [_text(v) for v in np.atleast_1d(attrs['%s%d' % (key, i)])]
        i += 1
    if not out:
        raise H5Error("no '%s' attribute: not a Keras weight file" % key)
    return out


def _collect(g, read):
    out = {}
    for layer in _names_attr(g.attrs, 'layer_names'):
        lg = g[layer]
        for wname in _names_attr(lg.attrs, 'weight_names') if len(lg.attrs) else []:
            arr = np.ascontiguousarray(read(lg, wname), dtype=np.float32)
            out[wname.rsplit(':', 1)[0]] = arr
    return out


# ----------------------------------------------------------------------------------------------------- writer
# The counterpart for ``model.save_weights('x.h5')`` (networks/inceptionv3.py:84-88 of the reference) where h5py is
# not installed: the same subset of the format, laid out the way libhdf5 1.8/1.10 lays out a file written with
# default settings -- superblock 0, old-style groups (version-1 B-tree over symbol-table nodes of up to 2 * 4 entries,
# up to 2 * 16 children per B-tree node, names in a local heap), version-1 object headers, contiguous little-endian
# float32 datasets, attributes holding fixed-length NUL-padded strings.  tests/test_h5lite.py reads such files back
# with this module's reader AND with the real HDF5 library (h5py under /opt/conda, where present).
_LEAF_K, _NODE_K = 4, 16


def _pad8(b):
    return b + b'\x00' * (-len(b) % 8)


def _dt_float32():
    # class 1 (floating point) version 1; little-endian, msb-implied mantissa; sign bit 31; 32-bit precision,
    # exponent at 23 (8 bits, bias 127), mantissa at 0 (23 bits)
    return bytes([0x11, 0x20, 0x1F, 0x00]) + struct.pack('<I', 4) + struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)


def _dt_string(n):
    # class 3 (string) version 1; NUL-padded, ASCII
    return bytes([0x13, 0x01, 0x00, 0x00]) + struct.pack('<I', n)


def _dataspace_msg(shape):
    return bytes([1, len(shape), 0, 0, 0, 0, 0, 0]) + b''.join(struct.pack('<Q', int(d)) for d in shape)


def _message(mtype, body):
    body = _pad8(body)
    if len(body) > 0xFFFF:
        raise H5Error('object-header message of %d bytes exceeds the 64 KB limit of the format (Keras splits its name '
                      'lists into layer_names0, layer_names1, ... beyond it; not needed by this package\'s models)' % len(body))
    return struct.pack('<HHB3x', mtype, len(body), 0) + body


def _attribute_msg(name, values):
    """Attribute holding a fixed-length string (scalar, ``bytes``) or a 1-D array of them (list of ``bytes``)."""
    nm = name.encode('utf-8') + b'\x00'
    if isinstance(values, bytes):
        width, shape, data = max(len(values), 1), (), values
        data = data.ljust(width, b'\x00')
    else:
        width = max([len(v) for v in values] + [1])
        shape, data = (len(values),), b''.join(v.ljust(width, b'\x00') for v in values)
    dt, ds = _dt_string(width), _dataspace_msg(shape)
    body = struct.pack('<BxHHH', 1, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data
    return _message(0x0C, body)


def _object_header(messages):
    body = b''.join(messages)
    return struct.pack('<BxHII4x', 1, len(messages), 1, len(body)) + body


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)                             # the superblock goes here at the end

    def alloc(self, data):
        self.buf += b'\x00' * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def dataset(self, arr):
        shape = np.shape(arr)                                # (ascontiguousarray would turn a scalar into shape (1,))
        arr = np.ascontiguousarray(arr, dtype='<f4')
        raw = self.alloc(arr.tobytes()) if arr.size else UNDEF
        layout = struct.pack('<BBQQ', 3, 1, raw, arr.nbytes)             # version 3, contiguous
        return self.alloc(_object_header([_message(0x01, _dataspace_msg(shape)), _message(0x03, _dt_float32()),
                                          _message(0x08, layout)]))

    def group(self, links, attrs=()):
        """links: {name: object-header address}; returns (header address, B-tree address, heap address)."""
        names = sorted(links, key=lambda s: s.encode('utf-8'))
        heap, offs = bytearray(b'\x00' * 8), {}              # offset 0: the empty string (the B-tree's first key)
        for n in names:
            offs[n] = len(heap)
            heap += _pad8(n.encode('utf-8') + b'\x00')
        data_addr = self.alloc(bytes(heap))
        heap_addr = self.alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), 1, data_addr))   # free list: none (1)
        # symbol-table nodes of up to 2 * _LEAF_K entries
        children = []                                        # (address, heap offset of the largest name below it)
        per = 2 * _LEAF_K
        for i in range(0, max(len(names), 1), per):
            part = names[i:i + per]
            node = b'SNOD' + struct.pack('<BxH', 1, len(part))
            for n in part:
                node += struct.pack('<QQII16x', offs[n], links[n], 0, 0)
            node += b'\x00' * (40 * (per - len(part)))
            children.append((self.alloc(node), offs[part[-1]] if part else 0))
        level = 0
        while True:                                          # B-tree levels until one node is left
            nodes, fan = [], 2 * _NODE_K
            for i in range(0, len(children), fan):
                part = children[i:i + fan]
                first_key = 0 if i == 0 else children[i - 1][1]
                body = struct.pack('<Q', first_key)
                for addr, key in part:
                    body += struct.pack('<QQ', addr, key)
                body += b'\x00' * (16 * (fan - len(part)))
                nodes.append([b'TREE' + struct.pack('<BBH', 0, level, len(part)), body, part[-1][1]])
            addrs = []
            for j, (head, body, _) in enumerate(nodes):      # siblings are contiguous: their addresses are known up front
                self.buf += b'\x00' * (-len(self.buf) % 8)
                base, size = len(self.buf), len(head) + 16 + len(body)
                left = base - size if j > 0 else UNDEF
                right = base + size if j + 1 < len(nodes) else UNDEF
                addrs.append(self.alloc(head + struct.pack('<QQ', left, right) + body))
            children = [(a, n[2]) for a, n in zip(addrs, nodes)]
            if len(children) == 1:
                break
            level += 1
        btree = children[0][0]
        header = self.alloc(_object_header([_message(0x11, struct.pack('<QQ', btree, heap_addr))] + list(attrs)))
        return header, btree, heap_addr

    def finish(self, root):
        header, btree, heap = root
        sb = SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack('<HHI', _LEAF_K, _NODE_K, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack('<QQII', 0, header, 1, 0) + struct.pack('<QQ', btree, heap)   # root entry, cached B-tree + heap
        assert len(sb) == 96
        self.buf[0:96] = sb
        return bytes(self.buf)


def write_keras_weights(path, params):
    """Writes ``{'layer/.../weight': ndarray}`` as a Keras ``save_weights`` file: one group per layer in first-seen
    order (``layer_names``), inside it the datasets ``<layer>/<weight>:0`` (``weight_names``), float32."""
    by_layer = {}
    for k, v in params.items():
        layer, w = k.rsplit('/', 1)
        by_layer.setdefault(layer, []).append((w, v))
    w = _Writer()

    def tree(entries):
        """entries: {path tuple: ndarray} -> links of one group level, built bottom-up."""
        here, below = {}, {}
        for path, arr in entries.items():
            if len(path) == 1:
                here[path[0]] = w.dataset(arr)
            else:
                below.setdefault(path[0], {})[path[1:]] = arr
        for name, sub in below.items():
            here[name] = w.group(tree(sub))[0]
        return here

    layers = {}
    for layer, ws in by_layer.items():
        wnames = ['%s/%s:0' % (layer, n) for n, _ in ws]
        links = tree({tuple(p.split('/')): arr for p, (_, arr) in zip(wnames, ws)})
        layers[layer] = w.group(links, [_attribute_msg('weight_names', [n.encode('utf-8') for n in wnames])])[0]
    for layer in by_layer:
        if '/' in layer:
            raise H5Error("layer name '%s' contains '/': not representable as one Keras layer group" % layer)
    root = w.group(layers, [_attribute_msg('layer_names', [n.encode('utf-8') for n in by_layer]),
                            _attribute_msg('backend', b'tensorflow')])
    with open(path, 'wb') as fh:
        fh.write(w.finish(root))
