"""OpenFace NN4.small2 inception network (96x96x3 -> 128-d, L2-normalised) on the MI355X.

Mirrors the reference class deep_insight_face/networks/inceptionv3.py:63-91
(``InceptionNetwork(input_shape, emd_size, weights)`` with ``__call__``,
``predict_on_batch``, ``save_weights``); the layer graph of :93-309 is built inside
libdif.so (csrc/net.hip: build_nn4): LRN, L2-pooling and the inception concats included.
"""
import os
import typing

import numpy as np

from .triplet import DifEmbedder


def load_weights(model_dir_path="./facenet-weights"):
    """The OpenFace CSV weight directory -> {layer name: [arrays]} exactly as the reference builds it
    (inceptionv3.py:28-60): per convolution ``<name>_w.csv`` (flat, in [cout, cin, kh, kw] order ->
    reshaped by the ``conv_shape`` table and transposed to Keras HWIO) and ``<name>_b.csv``; per batch norm
    ``_w`` (gamma), ``_b`` (beta), ``_m`` (moving mean), ``_v`` (moving variance); ``dense_w.csv`` as
    [128, 736] transposed, ``dense_b.csv``.  The shape table is not typed in here: it is read back from the
    library's own parameter table, which tests/test_structure.py pins to the reference's ``conv_shape``."""
    assert model_dir_path is not None, "Invalid model directory path"
    spec = dict(DifEmbedder('nn4', 'v2', 128, (96, 96, 3), max_batch=1).param_spec())
    paths = {n.replace('.csv', ''): os.path.join(model_dir_path, n)
             for n in os.listdir(model_dir_path) if not n.startswith('.')}

    def read(key):
        return np.atleast_1d(np.genfromtxt(paths[key], delimiter=',', dtype=np.float32))

    weights_dict = {}
    for pname, shape in spec.items():
        name, leaf = pname.rsplit('/', 1)
        if name in weights_dict:
            continue
        if 'conv' in name:
            kh, kw, cin, cout = spec[name + '/kernel']
            conv_w = np.transpose(np.reshape(read(name + '_w'), (cout, cin, kh, kw)), (2, 3, 1, 0))
            weights_dict[name] = [conv_w, read(name + '_b')]
        elif 'bn' in name:
            weights_dict[name] = [read(name + '_w'), read(name + '_b'), read(name + '_m'), read(name + '_v')]
        elif 'dense' in name:
            fin, fout = spec[name + '/kernel']
            weights_dict[name] = [np.transpose(np.reshape(read('dense_w'), (fout, fin)), (1, 0)), read('dense_b')]
    return weights_dict


def load_weights_from_FaceNet(FRmodel, model_dir_path):
    """inceptionv3.py:14-25: every layer's arrays from the CSV directory into the model."""
    leaves = {2: ('kernel', 'bias'), 4: ('gamma', 'beta', 'moving_mean', 'moving_variance')}
    params = {}
    for name, arrays in load_weights(model_dir_path).items():
        for leaf, a in zip(leaves[len(arrays)], arrays):
            params['%s/%s' % (name, leaf)] = np.ascontiguousarray(a, dtype=np.float32)
    FRmodel.set_weights(params)


class InceptionNetwork:
    def __init__(self, input_shape: typing.Tuple = (96, 96, 3), emd_size: int = 128, weights=None,
                 max_batch: int = 256) -> None:
        self.input_shape = tuple(input_shape)
        assert self.input_shape == (96, 96, 3), "Invalid Input shape, Shape should be of dimension (96, 96, 3)"
        self.emd_size = emd_size
        self.model = DifEmbedder('nn4', 'v2', emd_size, self.input_shape, max_batch=max_batch, name='nn4.small2')
        if weights is not None:
            # the reference wants Keras ".h5" (inceptionv3.py:70); the native container is ".npz"
            assert weights and weights.endswith((".h5", ".npz")), "Invalid pretrained weights"
            self._load_weights(weights)

    def __call__(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    def __getattr__(self, item: str):
        model = self.__dict__.get('model')
        if model is not None and hasattr(model, item):
            return getattr(model, item)
        raise AttributeError(item)

    def _load_weights(self, model_dir_path: str):
        """Load Model weight from csv (a directory, inceptionv3.py:76-82) or from a weight file."""
        if os.path.basename(model_dir_path).endswith((".h5", ".npz")):
            return self.model.load_weights(model_dir_path)
        return load_weights_from_FaceNet(self.model, model_dir_path)

    def save_weights(self, model_dir_path: str):
        assert model_dir_path and model_dir_path.endswith((".h5", ".npz")), "Invalid weights format"
        self.model.save_weights(model_dir_path)

    def predict_on_batch(self, img):
        return self.model.predict_on_batch(img)
