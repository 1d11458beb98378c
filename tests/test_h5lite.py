"""Keras HDF5 weight files through networks/h5lite.py (pure Python): against a committed file written by the real HDF5
library (tests/gen_h5_fixture.py under /opt/conda's h5py 3.3 / libhdf5 1.10) and, where that interpreter exists, against
a full NN4.small2 weight file written on the spot and loaded through the reference's call
(InceptionNetwork(weights='x.h5'), inceptionv3.py:68-71)."""
import os
import subprocess

import numpy as np
import pytest

from deep_insight_face.networks import h5lite

CONDA_PY = '/opt/conda/bin/python3.9'
HERE = os.path.dirname(os.path.abspath(__file__))


def test_tiny_fixture(golden_dir):
    path = os.path.join(golden_dir, 'keras_tiny.h5')
    f = h5lite.H5File(path)
    assert f.root.keys() == ['model_weights']
    g = f.root['model_weights']
    names = [h5lite._text(v) for v in g.attrs['layer_names']]
    assert names[:5] == ['conv1', 'input_1', 'bn1', 'dense_layer', 'scalar_layer'] and len(names) == 45
    assert h5lite._text(g.attrs['backend']) == 'tensorflow' and h5lite._text(g.attrs['keras_version']) == '2.4.0'
    assert sorted(g.keys()) == sorted(names)                       # 45 links: a B-tree with several leaves
    rng = np.random.default_rng(3)                                 # the generator's stream (tests/gen_h5_fixture.py: tiny)
    kernel = rng.standard_normal((3, 3, 3, 8)).astype(np.float32)
    bias = rng.standard_normal(8).astype(np.float32)
    assert np.array_equal(g['conv1/conv1/kernel:0'], kernel) and np.array_equal(g['conv1']['conv1/bias:0'], bias)
    bn = [rng.standard_normal(8).astype(np.float32) for _ in range(4)]
    assert np.array_equal(g['bn1/bn1/moving_variance:0'], bn[3])
    dense = rng.standard_normal((40, 5)).astype(np.float64)
    got = g['dense_layer/dense_layer/kernel:0']
    assert got.dtype == np.float64 and np.array_equal(got, dense)
    assert np.array_equal(g['dense_layer/dense_layer/bias:0'], np.arange(5)) and g['scalar_layer/scalar_layer/iterations:0'] == 7
    assert g['input_1'].keys() == [] and len(g['input_1'].attrs['weight_names']) == 0
    w = h5lite.read_keras_weights(path)
    assert len(w) == 2 + 4 + 2 + 1 + 40 and w['conv1/kernel'].dtype == np.float32
    assert np.array_equal(w['dense_layer/kernel'], dense.astype(np.float32))
    with pytest.raises(KeyError):
        g['nope']


def test_not_hdf5(tmp_path):
    p = tmp_path / 'x.h5'
    p.write_bytes(b'not an hdf5 file at all' * 100)
    with pytest.raises(h5lite.H5Error, match='not an HDF5 file'):
        h5lite.H5File(str(p))


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason='no interpreter with h5py to write the file')
def test_nn4_weights_written_by_hdf5_library(tmp_path):
    """NN4.small2's 224 weights saved in Keras layout by h5py, loaded by the product without h5py."""
    from deep_insight_face.networks.inceptionv3 import InceptionNetwork
    from deep_insight_face.networks.weights import save_npz, synth_params
    from oracle import nets
    p = synth_params(nets.model_spec('nn4', 128, 96))
    npz, h5 = str(tmp_path / 'w.npz'), str(tmp_path / 'nn4.h5')
    save_npz(npz, p)
    r = subprocess.run([CONDA_PY, os.path.join(HERE, 'gen_h5_fixture.py'), 'npz', npz, h5], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip('h5py not usable there: ' + r.stderr[-200:])
    net = InceptionNetwork((96, 96, 3), 128, weights=h5)             # the reference's constructor path
    got = net.get_weights()
    assert set(got) == set(p) and all(np.array_equal(got[k], p[k]) for k in p)


def _many_weights(n_layers=300, seed=5):
    rng = np.random.default_rng(seed)
    p = {}
    for i in range(n_layers):                                      # 300 layer groups: a two-level B-tree at the root
        p['block%03d_conv/kernel' % i] = rng.standard_normal((3, 3, 2, 4)).astype(np.float32)
        p['block%03d_conv/bias' % i] = rng.standard_normal(4).astype(np.float32)
    p['bn/moving_mean'] = np.zeros(5, np.float32)
    p['empty/kernel'] = np.zeros((0, 3), np.float32)               # a dataset without storage
    p['z_last/alpha'] = np.float32(0.25).reshape(())               # scalar dataspace
    return p


def test_writer_round_trip(tmp_path):
    """h5lite.write_keras_weights -> h5lite's own reader: names, order, shapes, bits."""
    p = _many_weights()
    path = str(tmp_path / 'w.h5')
    h5lite.write_keras_weights(path, p)
    f = h5lite.H5File(path)
    assert [h5lite._text(v) for v in f.root.attrs['layer_names']] == list(dict.fromkeys(k.rsplit('/', 1)[0] for k in p))
    assert h5lite._text(f.root.attrs['backend']) == 'tensorflow'
    assert sorted(f.root.keys()) == sorted(set(k.rsplit('/', 1)[0] for k in p))
    got = {}
    for layer in h5lite._names_attr(f.root.attrs, 'layer_names'):
        for wn in h5lite._names_attr(f.root[layer].attrs, 'weight_names'):
            got[wn.rsplit(':', 1)[0]] = f.root[layer][wn]
    assert set(got) == set(p)
    for k in p:
        assert got[k].shape == p[k].shape and got[k].dtype == np.float32 and np.array_equal(got[k], p[k]), k
    with pytest.raises(h5lite.H5Error, match="contains '/'"):
        h5lite.write_keras_weights(path, {'a/b/kernel': np.zeros(1, np.float32)})


def test_save_weights_h5_without_h5py(tmp_path, monkeypatch):
    """The reference's ``model.save_weights('x.h5')`` (inceptionv3.py:84-88) on an interpreter without h5py: the file goes
    through the pure-Python writer and loads back through ``load_weights`` (host logic only: no GPU involved)."""
    import builtins
    from deep_insight_face.networks import weights as W
    real_import = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == 'h5py':
            raise ImportError('h5py hidden by the test')
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, '__import__', no_h5py)
    p = {k: v for k, v in _many_weights(12).items() if v.ndim > 0}
    path = str(tmp_path / 'm.h5')
    W.save_keras_h5(path, p)
    got = h5lite.read_keras_weights(path)
    assert set(got) == set(p) and all(np.array_equal(got[k], p[k]) for k in p)


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason='no interpreter with h5py to read the file')
def test_writer_output_read_by_hdf5_library(tmp_path):
    """A file written by h5lite's writer, read by the REAL HDF5 library (h5py 3.3 / libhdf5 1.10) the way Keras reads it."""
    p = _many_weights()
    h5, npz = str(tmp_path / 'w.h5'), str(tmp_path / 'back.npz')
    h5lite.write_keras_weights(h5, p)
    r = subprocess.run([CONDA_PY, os.path.join(HERE, 'gen_h5_fixture.py'), 'read', h5, npz], capture_output=True, text=True)
    if r.returncode != 0 and 'No module named' in r.stderr:
        pytest.skip('h5py not usable there: ' + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(npz)
    got = {k.replace('::', '/'): z[k] for k in z.files if k != 'n_links'}
    assert set(got) == set(p)
    for k in p:
        assert got[k].dtype == np.float32 and got[k].shape == p[k].shape and np.array_equal(got[k], p[k]), k
    assert int(z['n_links']) == 2 * len(set(k.rsplit('/', 1)[0] for k in p)) + len(p)     # layer group + inner group + datasets
