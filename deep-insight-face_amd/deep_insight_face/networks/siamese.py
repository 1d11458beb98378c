"""Siamese flavour of the embedding networks on the MI355X.

Mirrors the inference side of deep_insight_face/networks/siamese.py: ``euclidean_distance`` (:22-24),
the shared-tower ``bottleneck_network`` with its two heads (:65-132; v1 is the same stack as the triplet
builder's v1, v2 is the siamese-only Conv1x1 / MaxPool('same') / BN / Dense(relu) stack) and
``buildin_models`` (:135-160), whose two-input Keras model -- the same tower on both inputs, then the
distance -- becomes ``SiameseModel``.  Training pieces (contrastive loss, optimiser, initialisers) are
out of scope.
"""
import typing

import numpy as np
import torch

from .. import _native as N
from .triplet import DifEmbedder

EPSILON = 1e-7     # keras.backend.epsilon()


def euclidean_distance(vects):
    """sqrt(max(sum((x - y)^2, axis=1, keepdims=True), epsilon)) -- siamese.py:22-24.  NumPy in ->
    NumPy out, tensors in -> CUDA tensor out; the squared distance is dif_pairwise (metric 0)."""
    x, y = vects
    dev = N.require_device()
    a, was_np = N.to_device_f32(x, dev)
    b, _ = N.to_device_f32(y, dev)
    if a.dim() != 2 or a.shape != b.shape:
        raise ValueError('expected two [N, d] batches of the same shape')
    n = a.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=dev)
    if n:
        N.check(N.lib.dif_pairwise(N.ptr(a), n, N.ptr(b), n, a.shape[1], 0, N.ptr(out), N.stream_ptr()))
    d = torch.sqrt(torch.clamp(out, min=EPSILON))[:, None]
    return d.cpu().numpy() if was_np else d


def eucl_dist_output_shape(shapes):
    shape1, _ = shapes
    return (shape1[0], 1)


class bottleneck_network:
    """Base network shared by the two towers (siamese.py:65-132); same constructor and call."""

    def __init__(self, net="resnet", emd_size=128, input_shape=(112, 112, 3), **kwargs):
        assert net in ('mobilenet', 'resnet', 'vgg16'), "Invalid bottleneck network"
        self.net = net
        self.emd_size = emd_size
        self.input_shape = input_shape
        self.kwargs = kwargs

    def __call__(self, default_model='v1', dropout=0.2):
        attr_name = 'build_models_' + default_model
        assert hasattr(self, attr_name), "Invalid default model version, must be from options (v1, v2)"
        return getattr(self, attr_name)(dropout=dropout)

    def _build(self, head):
        return DifEmbedder(self.net, head, self.emd_size, self.input_shape,
                           max_batch=self.kwargs.get('max_batch', 256))

    def build_models_v1(self, dropout: float = 0.3):
        return self._build('v1')

    def build_models_v2(self, dropout: float = 1.0):
        return self._build('sv2')


class SiameseModel:
    """The two-input distance model of buildin_models: ``predict([a, b])`` -> [N, 1] distances between
    the shared tower's embeddings of a and b."""

    def __init__(self, base_model: DifEmbedder):
        self.base_model = base_model

    def predict(self, pair, batch_size=None, **_):
        a, b = pair
        ea, eb = self.base_model.embed(a), self.base_model.embed(b)
        d = euclidean_distance((ea, eb))
        return d if torch.is_tensor(a) else d.cpu().numpy()

    predict_on_batch = predict
    __call__ = predict


def buildin_models(emd_size: int = 128, input_shape: typing.Tuple[int] = (112, 112, 3), summary: bool = False,
                   **kwargs) -> typing.Tuple[SiameseModel, DifEmbedder]:
    """-> (siamese distance model, shared base model) -- siamese.py:135-160.  (The reference passes
    ``emd_size`` into ``bottleneck_network``'s ``net`` slot and trips its own assert; the intent, a v1
    tower of the configured backbone, is what is built here: ``net=`` in kwargs, default 'resnet'.)"""
    assert len(input_shape) == 3, "Invalid input shape"
    net = kwargs.pop('net', 'resnet')
    base_model = bottleneck_network(net, emd_size, input_shape, **kwargs)(default_model='v1')
    print(">>>>>>> Siamese MODEL Loaded >>>>>>>")
    return SiameseModel(base_model), base_model
