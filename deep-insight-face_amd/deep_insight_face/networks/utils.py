"""Distance helpers of deep_insight_face/networks/utils.py, same names and meaning."""
import numpy as np
import torch

from .. import _native as N


def distance(emb1, emb2):
    """Squared L2 distance over every element (networks/utils.py:4-9), on the GPU."""
    from ..evaluation import utility
    a = emb1 if torch.is_tensor(emb1) else np.asarray(emb1, dtype=np.float32)
    b = emb2 if torch.is_tensor(emb2) else np.asarray(emb2, dtype=np.float32)
    d = utility.distance(a.reshape(1, -1), b.reshape(1, -1), 0)
    return d[0]


def distance_to_proba(distance):
    """[0, inf) -> (0, 1]: 1 / (1 + d)  (networks/utils.py:12-17)."""
    return 1 / (1 + distance)


def gaussian_kernel_dist_to_prob(distance, tuning_factor=1.0):
    """exp(-d / (2 sigma^2))  (networks/utils.py:20-29)."""
    if torch.is_tensor(distance):
        return torch.exp(-distance / (2 * tuning_factor ** 2))
    return np.exp(-distance / (2 * tuning_factor ** 2))


def set_gpu_limit(limit=2):
    """The reference caps TensorFlow's GPU memory (networks/utils.py:42-52); the HIP path
    sizes its workspaces explicitly, so there is nothing to cap."""
    N.require_device()
    return "GPU memory is managed by libdif (no limit applied)"
