"""Image -> embedding wrappers with the reference's signatures
(deep_insight_face/predictions.py:14-156), forward pass on the MI355X.

The per-image path of the reference (``np.expand_dims(img, 0)`` -> ``predict_on_batch``,
predictions.py:91-96,152-156) is kept as ``_embedding``; ``_embedding_batch`` is the
batched form the hot path is built for.  The ``* rescale`` multiply (and, for the siamese
wrapper, keras' vgg16 ``preprocess_input``: RGB->BGR + mean subtraction, predictions.py:95)
is fused into the first kernel instead of making a host pass over the pixels.
"""
from abc import ABCMeta, abstractmethod
from typing import Union

import numpy as np
from six import add_metaclass

from .evaluation import utility

_VGG_MEAN_BGR = (103.939, 116.779, 123.68)   # keras.applications.imagenet_utils, caffe mode


def _resize(image, size):
    """cv2.resize(image, size, interpolation=Image.BICUBIC) (predictions.py:93,154): PIL's BICUBIC
    constant (3) is cv2.INTER_AREA, i.e. area-coverage resampling.  Identity for
    crops already at the target size -- the case the hot path is specified for; other sizes are resampled
    on the device by the library's area-coverage kernel (dif_area_resize, the one behind dif_crop_resize).
    -> uint8 CUDA tensor [h, w, 3]."""
    import torch
    from . import _native as N
    w, h = int(size[0]), int(size[1])
    dev = N.require_device()
    img = np.ascontiguousarray(image)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError('expected an [H, W, 3] image, got shape %s' % (img.shape,))
    if img.shape[0] == h and img.shape[1] == w:
        return torch.from_numpy(img).to(dev)
    if img.dtype != np.uint8:
        raise ValueError('an image of another size than %dx%d must be uint8 to be resized (got %s)' % (w, h, img.dtype))
    src = torch.from_numpy(img).to(dev)
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    N.check(N.lib.dif_area_resize(N.ptr(src), 1, img.shape[0], img.shape[1], N.ptr(out), h, w, N.stream_ptr()))
    return out


@add_metaclass(ABCMeta)
class encoding_base:
    """Singleton-per-class base, as predictions.py:14-35."""

    _instances = {}

    def __init__(self, emd_model, img_size=(96, 96)):
        self.emd_model = emd_model
        self.img_size = img_size

    def read_image(self, img_path) -> np.ndarray:
        from PIL import Image
        return np.array(Image.open(img_path), dtype=np.uint8)

    def __new__(cls, *args, **kwargs):
        if cls not in cls._instances:
            cls._instances[cls] = super(encoding_base, cls).__new__(cls)
        return cls._instances[cls]

    @abstractmethod
    def _embedding(self, image: Union[str, np.ndarray]):
        pass

    def _prep(self, images):
        import torch
        return torch.stack([_resize(im, tuple(self.img_size)) for im in images])


def _to_numpy(out):
    """The wrappers return NumPy like the reference's predict_on_batch, whatever the batch was staged as."""
    return out.cpu().numpy() if hasattr(out, 'cpu') else out


def _saved_transform(model):
    """(scale, bias, bgr, hflip) currently set on a shared model, so a wrapper can put it back."""
    return getattr(model, '_transform', (1.0, (0.0, 0.0, 0.0), False, False))


class TripletPrediction(encoding_base):
    def __init__(self, emd_model, img_size=(96, 96)):
        assert len(img_size) == 2, "Invalid Image size format"
        super(TripletPrediction, self).__init__(emd_model, img_size)

    def _embedding_batch(self, images, rescale: float = 1 / 255.) -> np.ndarray:
        batch = self._prep(images)
        saved = _saved_transform(self.emd_model)
        self.emd_model.set_input_transform(scale=rescale)
        try:
            return _to_numpy(self.emd_model.predict_on_batch(batch))
        finally:
            self.emd_model.set_input_transform(*saved)        # the caller's transform, not the identity

    def _embedding(self, image: np.ndarray, rescale: float = 1 / 255.) -> np.ndarray:
        assert isinstance(image, np.ndarray), "Invalid image format, should be of type numpy array"
        return self._embedding_batch([image], rescale)

    def verify(self, image_path, identity, database, threshold=0.7):
        """(dist, is_valid): L2 distance between the image's embedding and
        database[identity], accepted below `threshold` (predictions.py:104-150)."""
        encoding = self._embedding(image_path)
        stored = np.asarray(database[identity], dtype=np.float32).reshape(1, -1)
        dist = float(np.sqrt(utility.distance(encoding.reshape(1, -1), stored, 0)[0]))
        if dist < threshold:
            print("It's " + str(identity))
            is_valid = True
        else:
            print("It's not " + str(identity))
            is_valid = False
        return dist, is_valid


class SiamesePrediction(encoding_base):
    def __init__(self, emd_model, img_size=(112, 112)):
        assert len(img_size) == 2, "Invalid Image size format"
        super(SiamesePrediction, self).__init__(emd_model, img_size)

    def _embedding_batch(self, images, rescale: float = 1 / 255.) -> np.ndarray:
        batch = self._prep(images)
        saved = _saved_transform(self.emd_model)
        self.emd_model.set_input_transform(scale=rescale, bias=tuple(-m for m in _VGG_MEAN_BGR), bgr=True)
        try:
            return _to_numpy(self.emd_model.predict_on_batch(batch))
        finally:
            self.emd_model.set_input_transform(*saved)

    def _embedding(self, image: np.ndarray, rescale: float = 1 / 255.) -> np.ndarray:
        assert isinstance(image, np.ndarray), "Invalid image format, should be of type numpy array"
        return self._embedding_batch([image], rescale)

    def verify(self, image_path, identity, database, threshold=0.3):
        """The reference pairs the encoding with every stored encoding of `identity`, runs the
        two-tower distance model (networks/siamese.py:22-24: sqrt(max(sum((x-y)^2), eps))) and
        keeps the first pair's value (predictions.py:71-79)."""
        encoding = self._embedding(image_path).reshape(1, -1)
        stored = np.asarray(database[identity], dtype=np.float32)
        stored = stored.reshape(-1, encoding.shape[1])
        d2 = utility.distance(np.repeat(encoding, len(stored), 0), stored, 0)
        dist = float(np.sqrt(np.maximum(d2, 1e-7))[0])
        if dist < threshold:
            print("It's " + str(identity))
            is_valid = True
        else:
            print("It's not " + str(identity))
            is_valid = False
        return dist, is_valid


def get_embedding(name: str, model, img_path, image_size=(112, 112, 3)) -> np.ndarray:
    """predictions.py:38-44 (broken in the reference: `_cls` undefined).  `name` selects the
    wrapper: 'triplet' or 'siamese'."""
    cls = {'triplet': TripletPrediction, 'siamese': SiamesePrediction}.get(str(name).lower())
    if cls is None:
        raise ValueError("name must be 'triplet' or 'siamese', got %r" % (name,))
    return cls(model, img_size=image_size[:-1])._embedding(img_path)
