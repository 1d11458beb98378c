"""Batched embedding loop of the evaluation path (deep_insight_face/evaluation/evals.py:53-59):
``predict_on_batch`` per generator batch, scattered into one [nrof_images, emd] array."""
import numpy as np


def embed_batches(emd_model, batches, nrof_images, embedding_size):
    """`batches` yields (X_batch[N,H,W,3], y[N]) with y the running image index, as the
    reference's generator does.  Returns (emb_arr float64 [n, emd], lab_arr)."""
    emb_arr = np.zeros((nrof_images, embedding_size))
    lab_arr = np.zeros((nrof_images,))
    idx = 0
    for X_batch, y in batches:
        lab = np.arange(idx, idx + len(y))
        idx += len(y)
        emb = emd_model.predict_on_batch(X_batch)
        lab_arr[lab] = y
        emb_arr[lab, :] = emb
    assert np.array_equal(lab_arr, np.arange(nrof_images)) is True, \
        'Wrong labels used for evaluation, possibly caused by training examples left in the input pipeline'
    return emb_arr, lab_arr
