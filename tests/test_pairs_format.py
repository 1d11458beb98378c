"""LFW pairs.txt helpers (deep_insight_face/evaluation/utility.py:222-262) -- host-side data
format on the input side of the evaluation path.  No GPU."""
import os

import pytest


def test_pairs_roundtrip(tmp_path):
    from deep_insight_face.evaluation import utility
    root = tmp_path / 'lfw'
    for name, n in (('Ann_Lee', 3), ('Bob_Ray', 2)):
        (root / name).mkdir(parents=True)
        for i in range(1, n + 1):
            ext = '.png' if (name == 'Bob_Ray' and i == 2) else '.jpg'
            (root / name / ('%s_%04d%s' % (name, i, ext))).write_bytes(b'x')
    pairs_file = tmp_path / 'pairs.txt'
    pairs_file.write_text('10\t300\nAnn_Lee\t1\t3\nAnn_Lee\t2\tBob_Ray\t2\nAnn_Lee\t1\t9\n')
    pairs = utility.read_pairs(str(pairs_file))
    assert len(pairs) == 3 and list(pairs[0]) == ['Ann_Lee', '1', '3']
    paths, issame = utility.get_paths(str(root), pairs)
    assert issame == [True, False]                       # the third pair's image 9 does not exist: skipped
    assert len(paths) == 4 and paths[0].endswith(os.path.join('Ann_Lee', 'Ann_Lee_0001.jpg'))
    assert paths[3].endswith('Bob_Ray_0002.png')
    with pytest.raises(RuntimeError, match='No file'):
        utility.add_extension(str(root / 'Ann_Lee' / 'Ann_Lee_0042'))
