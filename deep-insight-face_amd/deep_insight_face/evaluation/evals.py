"""Evaluation entry object and its batched embedding loop (deep_insight_face/evaluation/evals.py:19-78).
``embed_batches`` is the reference's loop body (evals.py:49-62) transcribed nearly line for line --
host glue around ``predict_on_batch``, including its assert message -- not a redesign."""
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


def _image_batches(image_paths, batch_size, target_size, rescale=1 / 255.):
    """(X_batch, y) over files, in order, y = running image index: the contract the reference's loop
    checks (evals.py:60-62).  Decoding/resizing is PIL on the host (the reference: img_read_n_resize)."""
    from PIL import Image
    for s in range(0, len(image_paths), batch_size):
        chunk = image_paths[s:s + batch_size]
        X = np.stack([np.asarray(Image.open(p).convert('RGB').resize(target_size[::-1], Image.BILINEAR), dtype=np.float32)
                      for p in chunk]) * np.float32(rescale)
        yield X, np.arange(s, s + len(chunk))


class TripletEvaluate:
    """LFW-protocol evaluation of an embedding model: the caller-facing object of
    deep_insight_face/evaluation/evals.py:19-78, same constructor and ``__call__`` signature.

    ``image_paths`` / ``pairs`` are what ``evaluation.utility.get_paths(lfw_dir, read_pairs(pairs.txt))``
    returns: the 2*P image files (pair members adjacent) and the P same/different flags.  The reference
    feeds them to its training data generator (evals.py:39-46, out of scope here: SURVEY section 2 "datagen");
    this class reads the files in order, or takes an injected ``batches(batch_size)`` iterable of
    (X_batch[N,H,W,3] float, y[N] running index).  The embedding loop is ``embed_batches`` (evals.py:49-62
    transcribed), the statistics are ``utility.evaluate`` (threshold sweep on the MI355X), the printed
    summary lines are the reference's (evals.py:66-75)."""

    def __init__(self, emd_model, image_paths, pairs, batches=None) -> None:
        self.emd_model = emd_model
        self.image_paths = image_paths
        self.pairs = pairs
        self.batches = batches

    def __call__(self, batch_size, nrof_folds, distance_metric, subtract_mean=False,
                 use_fixed_image_standardization=False, use_image_aug_random=False, save_output_detail=False):
        from scipy import interpolate
        from scipy.optimize import brentq
        from sklearn import metrics
        from . import utility
        shape = getattr(self.emd_model, 'output_shape', (128,))
        embedding_size = int(shape[-1])                      # (the reference hard-codes 128: evals.py:38)
        nrof_images = len(self.image_paths)
        if self.batches is not None:
            batches = self.batches(batch_size)
        else:
            h, w = self.emd_model.input_shape[:2]
            batches = _image_batches(list(self.image_paths), batch_size, (h, w))
        emb_arr, lab_arr = embed_batches(self.emd_model, batches, nrof_images, embedding_size)
        issame = np.asarray(self.pairs, dtype=bool)
        assert 2 * len(issame) == nrof_images, 'pairs must hold one flag per two images'
        tpr, fpr, accuracy, f1scores, val, val_std, far = utility.evaluate(
            emb_arr, issame, nrof_folds=nrof_folds, distance_metric=distance_metric, subtract_mean=subtract_mean)
        print('Accuracy: %2.5f+-%2.5f' % (np.mean(accuracy), np.std(accuracy)))
        print('Validation rate: %2.5f+-%2.5f @ FAR=%2.5f' % (val, val_std, far))
        print("F1 Score: %2.5f+-%2.5f" % (np.mean(f1scores), np.std(f1scores)))
        auc = metrics.auc(fpr, tpr)
        print('Area Under Curve (AUC): %1.3f' % auc)
        try:
            eer = brentq(lambda x: 1. - x - interpolate.interp1d(fpr, tpr, fill_value="extrapolate")(x), 0., 1.)
        except ValueError:
            eer = float('nan')
        print('Equal Error Rate (EER): %1.3f' % eer)
        print('>>>> ============== >>>>')
        self.results = dict(tpr=tpr, fpr=fpr, accuracy=accuracy, f1scores=f1scores, val=val, val_std=val_std, far=far,
                            auc=auc, eer=eer, embeddings=emb_arr)
        return self.results
