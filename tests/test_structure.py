"""Structure the reference holds as DATA, pinned on both boxes.

tests/golden/structure.json is written by tests/gen_golden.py from the reference tree: the
``WEIGHTS`` name list and ``conv_shape`` table of networks/inceptionv3.py (read with ``ast``; the
module itself needs Keras) and detector/yolo_cfg/yolov3-face.cfg parsed section by section.  The
library's parameter tables (dif_net_param_info) and launch tables (dif_net_op_info), and the oracle's,
are checked against it -- no hand-typed shapes, nothing read from /root/reference at test time."""
import json
import os

import numpy as np
import pytest

from oracle import detector as odet
from oracle import nets


@pytest.fixture(scope='module')
def structure(golden_dir):
    with open(os.path.join(golden_dir, 'structure.json')) as fh:
        return json.load(fh)


def test_nn4_weight_table_is_the_references(structure):
    """Every entry of the reference's conv_shape ([cout, cin, kh, kw], inceptionv3.py:365-403) and every
    name of its WEIGHTS list (:338-363) against the oracle's and the library's parameter tables."""
    from deep_insight_face.networks.inceptionv3 import InceptionNetwork
    lib = dict(InceptionNetwork((96, 96, 3), 128).param_spec())
    ora = dict(nets.nn4_spec(128))
    assert lib == ora
    shapes = structure['nn4_conv_shape']
    assert len(shapes) == 37
    for name, (cout, cin, kh, kw) in shapes.items():
        assert lib[name + '/kernel'] == (kh, kw, cin, cout), name
        assert lib[name + '/bias'] == (cout,), name                     # Conv2D default use_bias (inceptionv3.py:324)
    names = structure['nn4_weights']
    convs = [n for n in names if 'conv' in n]
    bns = [n for n in names if 'bn' in n]
    assert sorted(convs) == sorted(shapes) and len(bns) == 37 and names[-1] == 'dense_layer'
    for n in bns:
        cout = shapes[n.replace('_bn', '_conv') if n.replace('_bn', '_conv') in shapes else 'conv' + n[2:]][0]
        for leaf in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            assert lib['%s/%s' % (n, leaf)] == (cout,), n
    assert lib['dense_layer/kernel'] == (736, 128) and lib['dense_layer/bias'] == (128,)      # inceptionv3.py:54-55
    # nothing else: the library holds exactly the parameters the reference's loader fills
    assert len(lib) == 2 * 37 + 4 * 37 + 2


def _walk_cfg(sections, size):
    """Shape inference over the Darknet section list: per layer (channels, spatial size)."""
    net = sections[0]
    assert net['type'] == 'net'
    c, hw = int(net['channels']), size
    outs, convs, yolos = [], [], []
    for s in sections[1:]:
        t = s['type']
        if t == 'convolutional':
            k, stride, f = int(s['size']), int(s['stride']), int(s['filters'])
            hw_out = hw // stride
            convs.append({'cin': c, 'cout': f, 'k': k, 'stride': stride, 'hw_out': hw_out,
                          'bn': int(s.get('batch_normalize', 0)), 'act': s['activation']})
            c, hw = f, hw_out
        elif t == 'shortcut':
            src = outs[len(outs) + int(s['from'])]
            assert src == (c, hw)
        elif t == 'route':
            ids = [int(v) for v in s['layers'].split(',')]
            srcs = [outs[i if i >= 0 else len(outs) + i] for i in ids]
            assert len({h for _, h in srcs}) == 1
            c, hw = sum(ch for ch, _ in srcs), srcs[0][1]
        elif t == 'upsample':
            hw *= int(s['stride'])
        elif t == 'yolo':
            yolos.append({'hw': hw, 'c': c, 'mask': [int(v) for v in s['mask'].split(',')],
                          'anchors': [int(v) for v in s['anchors'].replace(' ', '').split(',')],
                          'classes': int(s['classes'])})
        else:
            raise AssertionError('unknown cfg section ' + t)
        outs.append((c, hw))
    return convs, yolos


def test_yolov3_face_cfg_layer_by_layer(structure):
    """The reference's cfg (through scripts/yolo_convert_tf.py:60-215 it IS the network) against the library:
    kernel shape, BN-or-bias, and the output map size of all 75 convolutions (from the MACs of the launch
    table), the three detection maps, the anchors and masks the decode uses."""
    from deep_insight_face.detector.run import yolo_v3_face
    sections = structure['yolov3_face_cfg']
    kinds = [s['type'] for s in sections]
    assert kinds.count('convolutional') == 75 and kinds.count('shortcut') == 23
    assert kinds.count('route') == 4 and kinds.count('upsample') == 2 and kinds.count('yolo') == 3
    convs, yolos = _walk_cfg(sections, 416)
    net = yolo_v3_face(1, 416)
    lib = dict(net.param_spec())
    assert lib == dict(odet.yolov3_spec(1))
    macs = {name: m for name, kern, m in net.op_table() if kern.startswith(('conv_igemm', 'stem'))}
    assert len(macs) == 75
    for i, c in enumerate(convs):
        assert lib['conv_%d/kernel' % i] == (c['k'], c['k'], c['cin'], c['cout']), i
        if c['bn']:
            assert c['act'] == 'leaky' and ('bn_%d/gamma' % i) in lib and ('conv_%d/bias' % i) not in lib, i
        else:
            assert c['act'] == 'linear' and lib['conv_%d/bias' % i] == (c['cout'],) and ('bn_%d/gamma' % i) not in lib, i
        assert macs['conv_%d' % i] == float(c['hw_out'] ** 2 * c['k'] ** 2 * c['cin'] * c['cout']), i
    assert [(y['hw'], y['hw'], y['c']) for y in yolos] == net.output_shapes == [(13, 13, 18), (26, 26, 18), (52, 52, 18)]
    # anchors / masks: the cfg's [yolo] sections, the shipped yolo_anchors.txt and what the decode is given
    from deep_insight_face.detector import run as drun
    from deep_insight_face.detector import yolov3 as dy
    anchors_txt = [int(v) for v in structure['yolo_anchors_txt'].replace(',', ' ').split()]
    assert anchors_txt == yolos[0]['anchors'] and all(y['anchors'] == anchors_txt for y in yolos)
    assert [y['mask'] for y in yolos] == [[6, 7, 8], [3, 4, 5], [0, 1, 2]]
    assert np.array_equal(np.asarray(drun.ANCHORS).reshape(-1), np.asarray(anchors_txt, dtype=np.float32))
    assert [list(m) for m in dy.ANCHOR_MASK_3] == [y['mask'] for y in yolos]
    assert structure['face_classes_txt'] == ['face'] and all(y['classes'] == 1 for y in yolos)
