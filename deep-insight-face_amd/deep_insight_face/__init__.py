"""MI355X-native drop-in for the embedding + match hot path of deep_insight_face.

Module names mirror the reference package so that call sites keep working:
``deep_insight_face.evaluation.utility.distance``, ``deep_insight_face.api.face_distance``,
``deep_insight_face.predictions.TripletPrediction``, ``deep_insight_face.networks.triplet.
bottleneck_network``, ``deep_insight_face.oneshot`` (the 1:N gallery match).  Everything
computes through libdif.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
from . import _native  # noqa: F401  (fails loudly when libdif.so is missing)

__all__ = ['_native']
