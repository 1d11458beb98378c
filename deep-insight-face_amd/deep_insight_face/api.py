"""face_recognition-style facade with the reference's def lines
(deep_insight_face/api.py:94,198,242); distances on the MI355X.

The reference module cannot be imported (FACEM_MODEL undefined at
def time, missing external landmark detector); only the signatures are the contract.
Detection / landmark alignment (api.py:107-195) is outside the hot path.
"""
import typing

import numpy as np

from .evaluation import utility
from .exceptions import FaceRecognitionException
from .networks.utils import distance_to_proba, gaussian_kernel_dist_to_prob

_MODEL = {'kind': 'triplet', 'model': None}


def set_face_recognition_model(model, kind='triplet'):
    """Registers the embedding model `face_encodings` uses (the reference loads it at
    import time from config, api.py:71-91)."""
    _MODEL['model'], _MODEL['kind'] = model, kind


def face_distance(face_encodings, face_to_compare):
    """np.linalg.norm(face_encodings - face_to_compare, axis=0) (api.py:94-104): for two
    vectors this is their euclidean distance; an empty list gives np.empty((0))."""
    if len(face_encodings) == 0:
        return np.empty((0))
    a = np.asarray(face_encodings, dtype=np.float32)
    b = np.asarray(face_to_compare, dtype=np.float32)
    a, b = np.broadcast_arrays(a, b)
    if a.ndim == 1:
        return np.sqrt(utility.distance(a[None, :], b[None, :], 0)[0])
    # axis-0 norm of a [n, d] difference: one value per column
    flat_a = np.ascontiguousarray(a.reshape(a.shape[0], -1).T)
    flat_b = np.ascontiguousarray(b.reshape(b.shape[0], -1).T)
    return np.sqrt(utility.distance(flat_a, flat_b, 0)).reshape(a.shape[1:])


def face_encodings(face_image: np.ndarray,
                   image_size: typing.Tuple,
                   do_show_plot: bool = False,
                   detect_and_crop: bool = True):
    """(thumb, encoding) for one face image (api.py:198-221)."""
    if detect_and_crop:
        raise FaceRecognitionException(
            "detect_and_alignment needs the external face_landmark_detector package (api.py:16-25), "
            "which is outside the MI355X hot path: pass an aligned crop with detect_and_crop=False")
    if _MODEL['model'] is None:
        raise ValueError("no embedding model registered: call set_face_recognition_model(model)")
    from .predictions import get_embedding
    thumb = [face_image]
    encoding = get_embedding(_MODEL['kind'], _MODEL['model'], thumb[0], image_size=image_size)
    return thumb, encoding


def compare_faces(known_face_encodings, face_encoding_to_check, tolerance=0.6):
    """(distance, probability) (api.py:242-256): Gaussian-kernel probability inside the
    tolerance, 1/(1+d) outside."""
    distance = face_distance(known_face_encodings[0], face_encoding_to_check[0])
    if distance <= tolerance:
        probability = gaussian_kernel_dist_to_prob(distance)
    else:
        probability = distance_to_proba(distance)
    return distance, probability
