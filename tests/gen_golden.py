"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN importable modules
(deep_insight_face/evaluation/utility.py, deep_insight_face/networks/utils.py) on
seeded inputs.  Run once in the build container, where /root/reference exists:

    python tests/gen_golden.py

The fixtures hold inputs' seeds + SHA-1 (the inputs are regenerated from the seed by
tests/golden_inputs.py; the digest guards against RNG drift) and the reference's
outputs.  No reference source text is stored.  Only data travels to the GPU box.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, '/root/reference')

from deep_insight_face.evaluation import utility as ref_utility   # noqa: E402  (the reference)
from deep_insight_face.networks import utils as ref_nutils        # noqa: E402  (the reference)

import golden_inputs as gi   # noqa: E402

OUT = os.path.join(HERE, 'golden')
os.makedirs(OUT, exist_ok=True)


def main():
    assert 'reference' in ref_utility.__file__, ref_utility.__file__

    # 1. row-paired distance, both metrics -------------------------------------------
    e1, e2 = gi.pair_inputs()
    np.savez(os.path.join(OUT, 'distance_pairs.npz'),
             sha=gi.digest(e1, e2),
             d0=ref_utility.distance(e1, e2, 0), d1=ref_utility.distance(e1, e2, 1),
             emd0=ref_utility.get_emd_distance(e1, e2, 0), emd1=ref_utility.get_emd_distance(e1, e2, 1))

    # 2. 1:N match = distance broadcast over the gallery + argmin (config 1: B=8, G=1000)
    for tag, (probes, gallery) in (('match_b8_g1000', gi.match_inputs()),
                                   ('match_ties', gi.match_tie_inputs()),
                                   ('match_unnormalised', gi.match_unnormalised_inputs())):
        rec = {'sha': gi.digest(probes, gallery)}
        for metric in (0, 1):
            full = np.stack([ref_utility.distance(q[None, :], gallery, metric) for q in probes])
            rec['full%d' % metric] = full.astype(np.float32)
            rec['idx%d' % metric] = np.array([int(np.argmin(row)) for row in full], dtype=np.int64)
        np.savez(os.path.join(OUT, tag + '.npz'), **rec)

    # 3. scalar maps of networks/utils.py and the api.py formulas built on them -------
    a, b = gi.vector_inputs()
    ds = np.array([0.0, 0.1, 0.5, 0.6, 0.61, 1.0, 2.5], dtype=np.float64)
    np.savez(os.path.join(OUT, 'scalars.npz'),
             sha=gi.digest(a, b),
             sq_l2=np.float64(ref_nutils.distance(a, b)),
             d=ds,
             proba=np.array([ref_nutils.distance_to_proba(x) for x in ds]),
             gauss=np.array([ref_nutils.gaussian_kernel_dist_to_prob(x) for x in ds]),
             gauss_t2=np.array([ref_nutils.gaussian_kernel_dist_to_prob(x, 2.0) for x in ds]))

    # 4. LFW-protocol pieces ----------------------------------------------------------
    emb1, emb2, issame = gi.roc_inputs()
    thresholds = np.arange(0, 4, 0.01)
    rec = {'sha': gi.digest(emb1, emb2, issame.astype(np.float32))}
    for metric in (0, 1):
        dist = ref_utility.distance(emb1, emb2, metric)
        rec['acc_m%d' % metric] = np.array([ref_utility.calculate_accuracy(t, dist, issame)
                                            for t in (0.2, 0.5, 1.0, 1.5)], dtype=np.float64)
        rec['valfar_m%d' % metric] = np.array([ref_utility.calculate_val_far(t, dist, issame)
                                               for t in (0.2, 0.5, 1.0, 1.5)], dtype=np.float64)
        for sub in (False, True):
            with contextlib.redirect_stdout(io.StringIO()):
                tpr, fpr, acc, f1 = ref_utility.calculate_roc(thresholds, emb1, emb2, issame, nrof_folds=10,
                                                              distance_metric=metric, subtract_mean=sub)
            k = 'm%d_s%d' % (metric, int(sub))
            rec['tpr_' + k], rec['fpr_' + k], rec['acc_' + k], rec['f1_' + k] = tpr, fpr, acc, f1
    np.savez(os.path.join(OUT, 'roc.npz'), **rec)

    # 5. near-ties: the reference's arg-min where float32 distances collide -----------
    probes, gallery = gi.match_near_tie_inputs()
    rec = {'sha': gi.digest(probes, gallery)}
    with np.errstate(invalid='ignore'):
        for metric in (0, 1):
            idx, dmin, nties, nnan = [], [], [], []
            for q in probes:
                d = ref_utility.distance(q[None, :], gallery, metric)
                i = int(np.argmin(d))
                idx.append(i)
                dmin.append(d[i])
                nties.append(int(np.count_nonzero(d == d[i])))
                nnan.append(int(np.count_nonzero(np.isnan(d))))
            rec['idx%d' % metric] = np.array(idx, dtype=np.int64)
            rec['dmin%d' % metric] = np.array(dmin, dtype=np.float32)
            rec['nties%d' % metric] = np.array(nties, dtype=np.int32)     # rows sharing the minimal float32 distance
            rec['nnan%d' % metric] = np.array(nnan, dtype=np.int32)       # rows whose reference distance is NaN (s > 1)
    np.savez(os.path.join(OUT, 'match_near_ties.npz'), **rec)

    # 6. degenerate rows and probes: zero norms, non-finite elements, anti-parallel rows ---------------
    rec = {}
    with np.errstate(all='ignore'):
        for name, probes, gallery in gi.match_degenerate_cases():
            rec[name + '_sha'] = gi.digest(probes, gallery)
            for metric in (0, 1):
                idx, dmin = [], []
                for q in probes:
                    d = ref_utility.distance(q[None, :], gallery, metric)
                    i = int(np.argmin(d))
                    idx.append(i)
                    dmin.append(d[i])
                rec['%s_idx%d' % (name, metric)] = np.array(idx, dtype=np.int64)
                rec['%s_dmin%d' % (name, metric)] = np.array(dmin, dtype=np.float32)
    np.savez(os.path.join(OUT, 'match_degenerate.npz'), **rec)

    # 7. structure the reference holds as DATA (no TensorFlow needed to read it) ------
    write_structure(os.path.join(OUT, 'structure.json'))
    print('wrote', sorted(os.listdir(OUT)))


def write_structure(path):
    """NN4.small2's layer-name list and weight-shape table (networks/inceptionv3.py: WEIGHTS, conv_shape),
    read with `ast` from the module's source without importing it (it needs Keras), and the YOLOv3-face
    Darknet cfg (detector/yolo_cfg/yolov3-face.cfg) parsed into its section list.  Values only."""
    import ast
    import json
    src = open('/root/reference/deep_insight_face/networks/inceptionv3.py').read()
    table = {}
    for node in ast.parse(src).body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in ('WEIGHTS', 'conv_shape'):
            table[node.targets[0].id] = ast.literal_eval(node.value)
    assert len(table['conv_shape']) == 37 and 'dense_layer' in table['WEIGHTS']
    sections = []
    for raw in open('/root/reference/deep_insight_face/detector/yolo_cfg/yolov3-face.cfg'):
        line = raw.split('#', 1)[0].strip()
        if not line:
            continue
        if line.startswith('['):
            sections.append({'type': line.strip('[]')})
        else:
            k, v = [t.strip() for t in line.split('=', 1)]
            sections[-1][k] = v
    keep = {'type', 'batch_normalize', 'filters', 'size', 'stride', 'pad', 'activation', 'from', 'layers', 'mask',
            'anchors', 'classes', 'num', 'width', 'height', 'channels'}
    sections = [{k: v for k, v in s.items() if k in keep} for s in sections]
    def _read(name):
        return open('/root/reference/deep_insight_face/detector/yolo_cfg/' + name).read().split()
    with open(path, 'w') as fh:
        json.dump({'nn4_weights': table['WEIGHTS'], 'nn4_conv_shape': table['conv_shape'],
                   'yolov3_face_cfg': sections, 'yolo_anchors_txt': ' '.join(_read('yolo_anchors.txt')),
                   'face_classes_txt': _read('face_classes.txt')}, fh, indent=0, sort_keys=True)


if __name__ == '__main__':
    main()
