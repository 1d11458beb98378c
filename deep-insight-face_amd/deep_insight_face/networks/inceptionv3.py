"""OpenFace NN4.small2 inception network (96x96x3 -> 128-d, L2-normalised) on the MI355X.

Mirrors the reference class deep_insight_face/networks/inceptionv3.py:63-91
(``InceptionNetwork(input_shape, emd_size, weights)`` with ``__call__``,
``predict_on_batch``, ``save_weights``); the layer graph of :93-309 is built inside
libdif.so (csrc/net.hip: build_nn4): LRN, L2-pooling and the inception concats included.
"""
import typing

from .triplet import DifEmbedder


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
        return self.model.load_weights(model_dir_path)

    def save_weights(self, model_dir_path: str):
        assert model_dir_path and model_dir_path.endswith((".h5", ".npz")), "Invalid weights format"
        if model_dir_path.endswith(".h5"):
            raise ValueError("Keras HDF5 needs h5py, which is not available here; save as .npz")
        self.model.save_weights(model_dir_path)

    def predict_on_batch(self, img):
        return self.model.predict_on_batch(img)
