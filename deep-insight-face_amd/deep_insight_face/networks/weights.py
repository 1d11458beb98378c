"""Weight containers for the embedding networks.

* ``synth_params`` -- seeded synthetic weights: He-normal
  kernels, BN gamma~U(0.5,1.5), beta~N(0,0.1), moving_mean~N(0,0.1),
  moving_variance~U(0.5,1.5), PReLU alpha=0.25, biases~N(0,0.1).  No pretrained face
  weights exist offline (the reference fetches ImageNet weights from the network,
  networks/triplet.py:89-93), so benchmarks and parity tests use these.
* ``save_npz`` / ``load_npz`` -- the on-disk format behind ``save_weights`` /
  ``load_weights``: one ``.npz`` entry per parameter, keyed by the Keras-style name.
"""
import zlib

import numpy as np


def synth_params(spec, seed=2024):
    """spec: iterable of (name, shape).  Each parameter gets its own stream seeded from
    (seed, crc32(name)), so the values do not depend on the order of the table."""
    out = {}
    for name, shape in spec:
        shape = tuple(int(s) for s in shape)
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        leaf = name.rsplit('/', 1)[-1]
        if leaf in ('kernel', 'depthwise_kernel'):
            if len(shape) == 4 and leaf == 'depthwise_kernel':
                fan_in = shape[0] * shape[1]
            elif len(shape) == 4:
                fan_in = shape[0] * shape[1] * shape[2]
            else:
                fan_in = shape[0]
            v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif leaf == 'gamma':
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf in ('beta', 'moving_mean', 'bias'):
            v = rng.standard_normal(shape) * 0.1
        elif leaf == 'moving_variance':
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf == 'alpha':
            v = np.full(shape, 0.25)
        else:
            raise ValueError('unknown parameter kind: %s' % name)
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def save_npz(path, params):
    np.savez(path, **{k.replace('/', '::'): v for k, v in params.items()})


def load_npz(path):
    with np.load(path) as z:
        return {k.replace('::', '/'): np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}


def save_keras_h5(path, params):
    """Keras ``save_weights`` layout: one group per layer, ``weight_names`` / ``layer_names`` attributes.  Through h5py
    where it is installed, otherwise through the pure-Python writer of networks/h5lite.py (same layout, float32)."""
    try:
        import h5py
    except ImportError:
        from . import h5lite
        h5lite.write_keras_weights(path, params)
        return
    by_layer = {}
    for k, v in params.items():
        layer, w = k.rsplit('/', 1)
        by_layer.setdefault(layer, []).append((w, v))
    with h5py.File(path, 'w') as f:
        f.attrs['layer_names'] = [n.encode('utf8') for n in by_layer]
        f.attrs['backend'] = b'tensorflow'
        for layer, ws in by_layer.items():
            g = f.create_group(layer)
            g.attrs['weight_names'] = [('%s/%s:0' % (layer, w)).encode('utf8') for w, _ in ws]
            for w, v in ws:
                g.create_dataset('%s/%s:0' % (layer, w), data=v)
