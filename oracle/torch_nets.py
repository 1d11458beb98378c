"""Independent torch-CPU (NCHW, torch.nn.functional) implementation of the embedding
networks.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); **PARITY UNPINNED** like
oracle/nets.py.  Two uses:
  * cross-check of the NumPy oracle (no reference arithmetic exists for the backbones, so two
    independently written implementations must agree): tests/test_oracle_nets.py;
  * bench.py's ``cpu_baseline``: SURVEY.md section 8(d) / BASELINE.md section 3 name torch-CPU ops
    with all host threads as the CPU stand-in for the reference's TF2/Keras CPU path
    (predictions.py:152-156 -> networks/triplet.py:119-141), which cannot run here.
Not the reference, not the product."""
import torch
import torch.nn.functional as F


def _t(p, name):
    return torch.from_numpy(p[name])


def _conv(x, p, name, stride=1, pad=0, bias=True):
    w = _t(p, name + '/kernel').permute(3, 2, 0, 1).contiguous()      # HWIO -> OIHW
    b = _t(p, name + '/bias') if bias and (name + '/bias') in p else None
    if isinstance(pad, tuple):                                         # (left, right, top, bottom)
        x = F.pad(x, pad)
        pad = 0
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def _bn(x, p, name, eps):
    return F.batch_norm(x, _t(p, name + '/moving_mean'), _t(p, name + '/moving_variance'),
                        _t(p, name + '/gamma'), _t(p, name + '/beta'), False, 0.0, eps)


def resnet50v2(x, p):
    eps = 1.001e-5
    y = _conv(x, p, 'conv1_conv', 2, 3)
    y = F.max_pool2d(F.pad(y, (1, 1, 1, 1)), 3, 2)
    for si, (f, blocks, s1) in enumerate(((64, 3, 2), (128, 4, 2), (256, 6, 2), (512, 3, 1))):
        for b in range(1, blocks + 1):
            n = 'conv%d_block%d' % (si + 2, b)
            stride = s1 if b == blocks else 1
            pre = F.relu(_bn(y, p, n + '_preact_bn', eps))
            if b == 1:
                sc = _conv(pre, p, n + '_0_conv', stride)
            else:
                sc = F.max_pool2d(y, 1, stride) if stride > 1 else y
            z = F.relu(_bn(_conv(pre, p, n + '_1_conv', bias=False), p, n + '_1_bn', eps))
            z = F.relu(_bn(_conv(z, p, n + '_2_conv', stride, 1, bias=False), p, n + '_2_bn', eps))
            y = sc + _conv(z, p, n + '_3_conv')
    return F.relu(_bn(y, p, 'post_bn', eps))


def head_gdc(f, p):
    y = _bn(_conv(f, p, 'head_conv', bias=False), p, 'head_bn1', 1e-3)
    y = F.prelu(y, _t(p, 'head_prelu/alpha'))
    dw = _t(p, 'head_dw/depthwise_kernel').permute(2, 3, 0, 1).contiguous()     # [C,1,H,W]
    y = F.conv2d(y, dw, groups=y.shape[1])
    y = _bn(y, p, 'head_bn2', 1e-3)
    y = _conv(y, p, 'head_pw', bias=False).flatten(1)
    y = y @ _t(p, 'head_dense/kernel')
    return y / y.pow(2).sum(1, keepdim=True).clamp_min(1e-12).sqrt()


def head_v1(f, p):
    y = F.relu(_conv(f, p, 'v1_conv1', pad=(0, 1, 0, 1)))
    y = F.max_pool2d(y, 2)
    y = F.relu(_conv(y, p, 'v1_conv2', pad=(0, 1, 0, 1)))
    y = F.max_pool2d(y, 2)
    y = y.permute(0, 2, 3, 1).flatten(1)                                # Keras flattens NHWC
    return y @ _t(p, 'embeddings/kernel') + _t(p, 'embeddings/bias')


def head_sv2(f, p):
    y = F.max_pool2d(F.relu(_conv(f, p, 'sv2_conv1')), 2, 2, ceil_mode=True)
    y = F.max_pool2d(F.relu(_conv(y, p, 'sv2_conv2')), 2, 2, ceil_mode=True)
    y = _bn(y, p, 'bn', 1e-3)
    y = y.permute(0, 2, 3, 1).reshape(y.shape[0], -1)                  # Keras flattens NHWC
    return F.relu(y @ _t(p, 'norm_embedding/kernel') + _t(p, 'norm_embedding/bias'))


def iresnet(x, p, layers):
    eps = 1e-5
    y = F.prelu(_bn(_conv(x, p, 'conv1', 1, 1, bias=False), p, 'bn1', eps), _t(p, 'prelu/alpha'))
    for li, nblk in enumerate(layers):
        for b in range(nblk):
            n = 'layer%d_%d' % (li + 1, b)
            stride = 2 if b == 0 else 1
            z = _conv(_bn(y, p, n + '_bn1', eps), p, n + '_conv1', 1, 1, bias=False)
            z = F.prelu(_bn(z, p, n + '_bn2', eps), _t(p, n + '_prelu/alpha'))
            z = _bn(_conv(z, p, n + '_conv2', stride, 1, bias=False), p, n + '_bn3', eps)
            sc = _bn(_conv(y, p, n + '_down_conv', stride, bias=False), p, n + '_down_bn', eps) if b == 0 else y
            y = z + sc
    y = _bn(y, p, 'bn2', eps).flatten(1)                                 # NCHW flatten: c*HW + h*W + w
    y = y @ _t(p, 'fc/kernel') + _t(p, 'fc/bias')
    y = F.batch_norm(y, _t(p, 'features/moving_mean'), _t(p, 'features/moving_variance'),
                     _t(p, 'features/gamma'), _t(p, 'features/beta'), False, 0.0, eps)
    return y / y.pow(2).sum(1, keepdim=True).clamp_min(1e-12).sqrt()


def vgg16(x, p):
    y = x
    for b, n in enumerate((2, 2, 3, 3, 3), 1):
        for i in range(1, n + 1):
            y = F.relu(_conv(y, p, 'block%d_conv%d' % (b, i), 1, 1))
        y = F.max_pool2d(y, 2, 2)
    return y


def mobilenetv2(x, p):
    eps = 1e-3

    def cpad(t):                       # keras correct_pad, as F.pad's (left, right, top, bottom)
        h, w = t.shape[2], t.shape[3]
        return ((0, 1) if w % 2 == 0 else (1, 1)) + ((0, 1) if h % 2 == 0 else (1, 1))

    def dw(t, name, stride):
        k = _t(p, name + '/depthwise_kernel').permute(2, 3, 0, 1).contiguous()     # [3,3,C,1] -> [C,1,3,3]
        if stride == 2:
            return F.conv2d(F.pad(t, cpad(t)), k, None, stride=2, groups=t.shape[1])
        return F.conv2d(t, k, None, stride=1, padding=1, groups=t.shape[1])

    y = F.relu6(_bn(_conv(x, p, 'Conv1', 2, cpad(x), bias=False), p, 'bn_Conv1', eps))
    block, cin = 0, 32
    for t, c, n, s in ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2),
                       (6, 320, 1, 1)):
        for i in range(n):
            stride = s if i == 0 else 1
            pre = 'expanded_conv' if block == 0 else 'block_%d' % block
            h = y
            if block:
                h = F.relu6(_bn(_conv(h, p, pre + '_expand', bias=False), p, pre + '_expand_BN', eps))
            h = F.relu6(_bn(dw(h, pre + '_depthwise', stride), p, pre + '_depthwise_BN', eps))
            h = _bn(_conv(h, p, pre + '_project', bias=False), p, pre + '_project_BN', eps)
            y = y + h if (cin == c and stride == 1) else h
            cin = c
            block += 1
    return F.relu6(_bn(_conv(y, p, 'Conv_1', bias=False), p, 'Conv_1_bn', eps))


def embed(x_nhwc, p, arch, head='v2'):
    x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        if arch in ('resnet', 'vgg16', 'mobilenet'):
            f = {'resnet': resnet50v2, 'vgg16': vgg16, 'mobilenet': mobilenetv2}[arch](x, p)
            if head == 'v2':
                return head_gdc(f, p).numpy()
            if head == 'v1':
                return head_v1(f, p).numpy()
            if head == 'sv2':
                return head_sv2(f, p).numpy()
            return f.permute(0, 2, 3, 1).contiguous().numpy()
        layers = {'iresnet50': (3, 4, 14, 3), 'iresnet100': (3, 13, 30, 3)}[arch]
        return iresnet(x, p, layers).numpy()


# ---------------------------------------------------------------------------- NN4.small2
def _cbr(x, p, conv, bn, stride=1, pad=0):
    y = _conv(x, p, conv, stride, pad)
    return F.relu(_bn(y, p, bn, 1e-5))


def _branch(x, p, layer, stride, pad):
    return _cbr(_cbr(x, p, layer + '_conv1', layer + '_bn1'), p, layer + '_conv2', layer + '_bn2', stride, pad)


def _lrn(x):
    # torch divides alpha by the window size: alpha' = 1e-4 * 11 reproduces tf.nn.lrn(depth_radius=5)
    return F.local_response_norm(x, size=11, alpha=11e-4, beta=0.75, k=1.0)


def _l2pool(x):
    return torch.sqrt(F.avg_pool2d(x * x, 3, 3) * 9)


def nn4(x, p):
    y = _cbr(x, p, 'conv1', 'bn1', 2, 3)
    y = F.max_pool2d(F.pad(y, (1, 1, 1, 1)), 3, 2)
    y = _lrn(y)
    y = _cbr(y, p, 'conv2', 'bn2')
    y = _cbr(y, p, 'conv3', 'bn3', 1, 1)
    y = _lrn(y)
    y = F.max_pool2d(F.pad(y, (1, 1, 1, 1)), 3, 2)
    cat = lambda ts: torch.cat(ts, dim=1)   # noqa: E731
    y = cat([_branch(y, p, 'inception_3a_3x3', 1, 1), _branch(y, p, 'inception_3a_5x5', 1, 2),
             F.pad(_cbr(F.max_pool2d(y, 3, 2), p, 'inception_3a_pool_conv', 'inception_3a_pool_bn'), (3, 4, 3, 4)),
             _cbr(y, p, 'inception_3a_1x1_conv', 'inception_3a_1x1_bn')])
    y = cat([_branch(y, p, 'inception_3b_3x3', 1, 1), _branch(y, p, 'inception_3b_5x5', 1, 2),
             F.pad(_cbr(_l2pool(y), p, 'inception_3b_pool_conv', 'inception_3b_pool_bn'), (4, 4, 4, 4)),
             _cbr(y, p, 'inception_3b_1x1_conv', 'inception_3b_1x1_bn')])
    y = cat([_branch(y, p, 'inception_3c_3x3', 2, 1), _branch(y, p, 'inception_3c_5x5', 2, 2),
             F.pad(F.max_pool2d(y, 3, 2), (0, 1, 0, 1))])
    y = cat([_branch(y, p, 'inception_4a_3x3', 1, 1), _branch(y, p, 'inception_4a_5x5', 1, 2),
             F.pad(_cbr(_l2pool(y), p, 'inception_4a_pool_conv', 'inception_4a_pool_bn'), (2, 2, 2, 2)),
             _cbr(y, p, 'inception_4a_1x1_conv', 'inception_4a_1x1_bn')])
    y = cat([_branch(y, p, 'inception_4e_3x3', 2, 1), _branch(y, p, 'inception_4e_5x5', 2, 2),
             F.pad(F.max_pool2d(y, 3, 2), (0, 1, 0, 1))])
    y = cat([_branch(y, p, 'inception_5a_3x3', 1, 1),
             F.pad(_cbr(_l2pool(y), p, 'inception_5a_pool_conv', 'inception_5a_pool_bn'), (1, 1, 1, 1)),
             _cbr(y, p, 'inception_5a_1x1_conv', 'inception_5a_1x1_bn')])
    y = cat([_branch(y, p, 'inception_5b_3x3', 1, 1),
             F.pad(_cbr(F.max_pool2d(y, 3, 2), p, 'inception_5b_pool_conv', 'inception_5b_pool_bn'), (1, 1, 1, 1)),
             _cbr(y, p, 'inception_5b_1x1_conv', 'inception_5b_1x1_bn')])
    y = F.avg_pool2d(y, 3, 1).flatten(1)
    y = y @ _t(p, 'dense_layer/kernel') + _t(p, 'dense_layer/bias')
    return y / y.pow(2).sum(1, keepdim=True).clamp_min(1e-12).sqrt()


def embed_nn4(x_nhwc, p):
    x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        return nn4(x, p).numpy()


# ---------------------------------------------------------------------------- YOLOv3-face
def yolov3(x_nhwc, p):
    import numpy as np
    x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2).contiguous()
    ci = [0]

    def cbl(t, k, stride):
        i = ci[0]
        ci[0] += 1
        w = _t(p, 'conv_%d/kernel' % i).permute(3, 2, 0, 1).contiguous()
        if stride == 2:
            t = F.conv2d(F.pad(t, (1, 0, 1, 0)), w, None, stride=2)
        else:
            t = F.conv2d(t, w, None, padding=1 if k == 3 else 0)
        return F.leaky_relu(_bn(t, p, 'bn_%d' % i, 1e-3), 0.1)

    def linear(t):
        i = ci[0]
        ci[0] += 1
        return F.conv2d(t, _t(p, 'conv_%d/kernel' % i).permute(3, 2, 0, 1).contiguous(), _t(p, 'conv_%d/bias' % i))

    with torch.no_grad():
        y = cbl(x, 3, 1)
        routes = {}
        for cout, blocks in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
            y = cbl(y, 3, 2)
            for _ in range(blocks):
                y = y + cbl(cbl(y, 1, 1), 3, 1)
            routes[cout] = y

        def head(t):
            for k in (1, 3, 1, 3, 1):
                t = cbl(t, k, 1)
            return t, linear(cbl(t, 3, 1))

        b, y13 = head(y)
        b, y26 = head(torch.cat([F.interpolate(cbl(b, 1, 1), scale_factor=2, mode='nearest'), routes[512]], 1))
        b, y52 = head(torch.cat([F.interpolate(cbl(b, 1, 1), scale_factor=2, mode='nearest'), routes[256]], 1))
    return [np.ascontiguousarray(t.permute(0, 2, 3, 1).numpy()) for t in (y13, y26, y52)]


# ---------------------------------------------------------------------------- MTCNN (oracle/mtcnn.py; not in the reference)
def mtcnn(x_nhwc, p, stage):
    """stage 'pnet' / 'rnet' / 'onet'; x already normalised; -> head output, NHWC map for P-Net, [N, C] otherwise."""
    x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2).contiguous()

    def cp(t, name, prelu=True):
        t = _conv(t, p, name)
        return F.prelu(t, _t(p, name + '_prelu/alpha')) if prelu else t

    with torch.no_grad():
        if stage == 'pnet':
            y = F.max_pool2d(cp(x, 'conv1'), 2, 2, ceil_mode=True)
            y = cp(cp(cp(y, 'conv2'), 'conv3'), 'head', False)
            return y.permute(0, 2, 3, 1).contiguous().numpy()
        if stage == 'rnet':
            y = F.max_pool2d(cp(x, 'conv1'), 3, 2, ceil_mode=True)
            y = F.max_pool2d(cp(y, 'conv2'), 3, 2, ceil_mode=True)
            y = cp(cp(cp(y, 'conv3'), 'fc1'), 'head', False)
            return y.flatten(1).numpy()
        y = F.max_pool2d(cp(x, 'conv1'), 3, 2, ceil_mode=True)
        y = F.max_pool2d(cp(y, 'conv2'), 3, 2, ceil_mode=True)
        y = F.max_pool2d(cp(y, 'conv3'), 2, 2, ceil_mode=True)
        y = cp(cp(cp(y, 'conv4'), 'fc1'), 'head', False)
        return y.flatten(1).numpy()
