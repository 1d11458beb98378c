"""LFW ``pairs.txt`` data format (the input side of the evaluation path).

Same semantics as the reference helpers deep_insight_face/evaluation/utility.py:222-262
(``get_paths``, ``add_extension``, ``read_pairs``): a header line, then tab-separated rows of
either ``name  i  j`` (same person, images i and j) or ``name_a  i  name_b  j`` (different
people); image files are ``<dir>/<name>/<name>_%04d.{jpg,png}``.
"""
import os

import numpy as np


def read_pairs(pairs_filename):
    """Rows of the file after its header line, split on tabs -> np.ndarray of lists."""
    with open(pairs_filename, 'r') as fh:
        rows = [ln.strip().split('\t') for ln in fh.readlines()[1:]]
    return np.array(rows, dtype=object) if len({len(r) for r in rows}) > 1 else np.array(rows)


def add_extension(path):
    for ext in ('.jpg', '.png'):
        if os.path.exists(path + ext):
            return path + ext
    raise RuntimeError('No file "%s" with extension png or jpg.' % path)


def _image(lfw_dir, name, index):
    return add_extension(os.path.join(lfw_dir, name, '%s_%04d' % (name, int(index))))


def get_paths(lfw_dir, pairs):
    """-> (flat list of image paths, two per kept pair; list of is-same flags).  Pairs whose
    files are missing are skipped and counted, as in the reference."""
    paths, issame, skipped = [], [], 0
    for pair in pairs:
        try:
            if len(pair) == 3:
                p0, p1, same = _image(lfw_dir, pair[0], pair[1]), _image(lfw_dir, pair[0], pair[2]), True
            elif len(pair) == 4:
                p0, p1, same = _image(lfw_dir, pair[0], pair[1]), _image(lfw_dir, pair[2], pair[3]), False
            else:
                raise RuntimeError('malformed pair %r' % (list(pair),))
        except RuntimeError:
            if len(pair) not in (3, 4):
                raise
            skipped += 1
            continue
        paths += (p0, p1)
        issame.append(same)
    if skipped > 0:
        print('Skipped %d image pairs' % skipped)
    return paths, issame
