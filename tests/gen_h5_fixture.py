"""Writes Keras-layout HDF5 weight files with the REAL HDF5 library (h5py), for tests/test_h5lite.py.

    /opt/conda/bin/python3.9 tests/gen_h5_fixture.py tiny tests/golden/keras_tiny.h5
    /opt/conda/bin/python3.9 tests/gen_h5_fixture.py npz <weights.npz> <out.h5>
    /opt/conda/bin/python3.9 tests/gen_h5_fixture.py read <in.h5> <out.npz>     # the other way: the real library reads a
                                                                                 # file written by h5lite's writer

(h5py is absent from the image's main interpreter but present under /opt/conda; the product never needs it: it reads
these files with deep_insight_face/networks/h5lite.py.)  The layout is what Keras' ``save_weights`` writes
(keras/saving/hdf5_format.py: save_weights_to_hdf5_group): root attributes ``layer_names``, ``backend``,
``keras_version``; one group per layer with a ``weight_names`` attribute and one dataset per weight, named
``<layer>/<weight>:0`` below the layer's group."""
import sys

import h5py
import numpy as np


def write(path, layers, model_save=False):
    with h5py.File(path, 'w') as f:
        g = f.create_group('model_weights') if model_save else f
        g.attrs['layer_names'] = [n.encode('utf8') for n, _ in layers]
        g.attrs['backend'] = b'tensorflow'
        g.attrs['keras_version'] = '2.4.0'                      # a str: h5py stores it as a variable-length string
        for name, weights in layers:
            lg = g.create_group(name)
            lg.attrs['weight_names'] = [('%s/%s:0' % (name, w)).encode('utf8') for w, _ in weights]
            for w, val in weights:
                d = lg.create_dataset('%s/%s:0' % (name, w), val.shape, dtype=val.dtype)
                if val.shape:
                    d[:] = val
                else:
                    d[()] = val


def tiny(path):
    rng = np.random.default_rng(3)
    layers = [('conv1', [('kernel', rng.standard_normal((3, 3, 3, 8)).astype(np.float32)),
                         ('bias', rng.standard_normal(8).astype(np.float32))]),
              ('input_1', []),
              ('bn1', [(k, rng.standard_normal(8).astype(np.float32)) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')]),
              ('dense_layer', [('kernel', rng.standard_normal((40, 5)).astype(np.float64)),      # float64 on purpose
                               ('bias', np.arange(5, dtype=np.int32))]),
              ('scalar_layer', [('iterations', np.array(7, dtype=np.int64))])]
    layers += [('block_%d' % i, [('kernel', rng.standard_normal((1, 1, 4, 4)).astype(np.float32))]) for i in range(40)]
    write(path, layers, model_save=True)


def from_npz(npz, out):
    z = np.load(npz)
    by_layer = {}
    for k in z.files:
        layer, w = k.replace('::', '/').rsplit('/', 1)
        by_layer.setdefault(layer, []).append((w, z[k]))
    write(out, sorted(by_layer.items()))


def to_npz(h5, out):
    """Reads a Keras-layout file the way Keras' loader does (layer_names / weight_names attributes) with the real library."""
    got = {}
    with h5py.File(h5, 'r') as f:
        assert f.attrs['backend'] in (b'tensorflow', 'tensorflow')
        for layer in f.attrs['layer_names']:
            g = f[layer.decode() if isinstance(layer, bytes) else layer]
            for wn in g.attrs['weight_names']:
                wn = wn.decode() if isinstance(wn, bytes) else wn
                got[wn.rsplit(':', 1)[0].replace('/', '::')] = np.asarray(g[wn])
        n_links = []
        f.visit(n_links.append)                                     # walks every group through the library's B-tree code
    np.savez(out, n_links=np.array(len(n_links)), **got)


if __name__ == '__main__':
    if sys.argv[1] == 'tiny':
        tiny(sys.argv[2])
    elif sys.argv[1] == 'read':
        to_npz(sys.argv[2], sys.argv[3])
    else:
        from_npz(sys.argv[2], sys.argv[3])
