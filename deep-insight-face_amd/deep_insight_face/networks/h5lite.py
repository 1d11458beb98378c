"""A small reader for Keras HDF5 weight files (``model.save_weights('x.h5')``), in pure Python + NumPy.

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
