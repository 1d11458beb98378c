"""NumPy restatement of the embedding forward pass.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  **PARITY UNPINNED** for everything in this file:

  * ResNet50V2 is ``tensorflow.keras.applications.ResNet50V2(include_top=False)``
    as selected by ``deep_insight_face/networks/triplet.py:87-91`` /
    ``networks/siamese.py:81-85`` -- third-party, un-vendored, no pinned version
    (requirements.txt is empty); TensorFlow is not installed here.  The layer
    table below restates the public Keras definition.
  * The heads follow the in-repo builders line by line:
    GDC head  = ``networks/triplet.py:119-141`` (build_models_v2),
    small head = ``networks/triplet.py:102-117`` / ``networks/siamese.py:91-105``
    (build_models_v1).
  * IResNet-50/100 and ArcMargin are absent from the reference; they follow the
    public ArcFace definitions.

All tensors are NHWC ("channels_last", as the reference: ``networks/inceptionv3.py:98``)
and are computed in ``dtype`` (float32 like the reference; float64 is offered
as an arbiter when two float32 implementations disagree in the last bits).
Convolution kernels are Keras HWIO ``[kh, kw, cin, cout]``; dense kernels are
``[in, out]``.
"""
import math

import numpy as np

BN_EPS_RESNET = 1.001e-5   # keras.applications resnet BN epsilon
BN_EPS_KERAS = 1e-3        # keras.layers.BatchNormalization default (head: triplet.py:127,130)
BN_EPS_IRESNET = 1e-5      # ArcFace IResNet (torch BatchNorm default)


# --------------------------------------------------------------------------- layers
def conv2d(x, w, bias=None, stride=1, pad=(0, 0, 0, 0)):
    """x [N,H,W,C], w [kh,kw,C,O]; pad = (top, bottom, left, right) explicit zeros.
    im2col + one matmul per call."""
    kh, kw, cin, cout = w.shape
    n, h, wd, c = x.shape
    assert c == cin, (x.shape, w.shape)
    if any(pad):
        x = np.pad(x, ((0, 0), (pad[0], pad[1]), (pad[2], pad[3]), (0, 0)))
    hp, wp = x.shape[1], x.shape[2]
    ho = (hp - kh) // stride + 1
    wo = (wp - kw) // stride + 1
    if kh == 1 and kw == 1:
        cols = x[:, ::stride, ::stride, :][:, :ho, :wo, :].reshape(n * ho * wo, cin)
    else:
        s = x.strides
        win = np.lib.stride_tricks.as_strided(
            x, shape=(n, ho, wo, kh, kw, cin),
            strides=(s[0], s[1] * stride, s[2] * stride, s[1], s[2], s[3]), writeable=False)
        cols = win.reshape(n * ho * wo, kh * kw * cin)
    y = cols @ w.reshape(kh * kw * cin, cout)
    if bias is not None:
        y = y + bias
    return y.reshape(n, ho, wo, cout)


def same_pad(size, k, stride):
    """TensorFlow 'SAME' padding for one axis -> (before, after)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


def batchnorm(x, p, prefix, eps):
    """Inference BN the way Keras evaluates it: x * (gamma * rsqrt(var+eps)) +
    (beta - mean * gamma * rsqrt(var+eps))."""
    g = p[prefix + '/gamma']
    b = p[prefix + '/beta']
    m = p[prefix + '/moving_mean']
    v = p[prefix + '/moving_variance']
    scale = (g / np.sqrt(v + np.asarray(eps, dtype=x.dtype))).astype(x.dtype)
    shift = (b - m * scale).astype(x.dtype)
    return x * scale + shift


def relu(x):
    return np.maximum(x, 0)


def prelu(x, alpha):
    return np.where(x >= 0, x, x * alpha)


def maxpool(x, k, stride, pad=(0, 0, 0, 0), pad_value=0.0):
    """Max pooling.  Keras ResNet50V2 pads with an explicit ZeroPadding2D before a
    VALID pool (pad_value 0); 'same' pools pad with -inf."""
    if any(pad):
        x = np.pad(x, ((0, 0), (pad[0], pad[1]), (pad[2], pad[3]), (0, 0)), constant_values=pad_value)
    n, h, w, c = x.shape
    ho = (h - k) // stride + 1
    wo = (w - k) // stride + 1
    s = x.strides
    win = np.lib.stride_tricks.as_strided(
        x, shape=(n, ho, wo, k, k, c),
        strides=(s[0], s[1] * stride, s[2] * stride, s[1], s[2], s[3]), writeable=False)
    return win.max(axis=(3, 4))


def l2_normalize(x, eps=1e-12):
    """tf.nn.l2_normalize(axis=1): x * rsqrt(max(sum(x^2), eps)).  triplet.py:138."""
    ss = np.sum(x * x, axis=1, keepdims=True)
    return x / np.sqrt(np.maximum(ss, np.asarray(eps, dtype=x.dtype)))


# --------------------------------------------------------------------------- ResNet50V2
RESNET50V2_STACKS = ((64, 3, 2), (128, 4, 2), (256, 6, 2), (512, 3, 1))  # (filters, blocks, stride on last)


def _block_v2(x, p, name, filters, stride, conv_shortcut):
    pre = relu(batchnorm(x, p, name + '_preact_bn', BN_EPS_RESNET))
    if conv_shortcut:
        sc = conv2d(pre, p[name + '_0_conv/kernel'], p[name + '_0_conv/bias'], stride=stride)
    elif stride > 1:
        sc = x[:, ::stride, ::stride, :]            # MaxPooling2D(1, strides=stride)
    else:
        sc = x
    y = conv2d(pre, p[name + '_1_conv/kernel'])
    y = relu(batchnorm(y, p, name + '_1_bn', BN_EPS_RESNET))
    y = conv2d(y, p[name + '_2_conv/kernel'], stride=stride, pad=(1, 1, 1, 1))
    y = relu(batchnorm(y, p, name + '_2_bn', BN_EPS_RESNET))
    y = conv2d(y, p[name + '_3_conv/kernel'], p[name + '_3_conv/bias'])
    return sc + y


def resnet50v2(x, p):
    """[N,H,W,3] -> [N,H/32 (ceil),W/32,2048] (4x4x2048 at 112 px)."""
    y = conv2d(x, p['conv1_conv/kernel'], p['conv1_conv/bias'], stride=2, pad=(3, 3, 3, 3))
    y = maxpool(y, 3, 2, pad=(1, 1, 1, 1))
    for si, (filters, blocks, stride1) in enumerate(RESNET50V2_STACKS):
        s = 'conv%d' % (si + 2)
        y = _block_v2(y, p, s + '_block1', filters, 1, True)
        for b in range(2, blocks):
            y = _block_v2(y, p, s + '_block%d' % b, filters, 1, False)
        y = _block_v2(y, p, s + '_block%d' % blocks, filters, stride1, False)
    return relu(batchnorm(y, p, 'post_bn', BN_EPS_RESNET))


def resnet50v2_spec(in_ch=3):
    spec = [('conv1_conv/kernel', (7, 7, in_ch, 64)), ('conv1_conv/bias', (64,))]
    cin = 64

    def bn(name, c):
        return [(name + '/' + k, (c,)) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')]

    for si, (f, blocks, _) in enumerate(RESNET50V2_STACKS):
        s = 'conv%d' % (si + 2)
        for b in range(1, blocks + 1):
            n = '%s_block%d' % (s, b)
            spec += bn(n + '_preact_bn', cin)
            if b == 1:
                spec += [(n + '_0_conv/kernel', (1, 1, cin, 4 * f)), (n + '_0_conv/bias', (4 * f,))]
            spec += [(n + '_1_conv/kernel', (1, 1, cin, f))] + bn(n + '_1_bn', f)
            spec += [(n + '_2_conv/kernel', (3, 3, f, f))] + bn(n + '_2_bn', f)
            spec += [(n + '_3_conv/kernel', (1, 1, f, 4 * f)), (n + '_3_conv/bias', (4 * f,))]
            cin = 4 * f
    spec += bn('post_bn', cin)
    return spec


# --------------------------------------------------------------------------- VGG16 / MobileNetV2
# keras.applications.VGG16 / MobileNetV2(alpha=1.0), include_top=False: the other two backbones
# bottleneck_network accepts (triplet.py:77,87-93).  Third-party, unpinned, absent from /root/reference
# and not installed: restated from the public Keras layer tables (names and shapes as Keras has them).
VGG16_CFG = ((64, 2), (128, 2), (256, 3), (512, 3), (512, 3))
MOBILENETV2_CFG = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
MOBILENETV2_EPS = 1e-3


def vgg16(x, p):
    """[N,H,W,3] -> [N,H/32,W/32,512]: 13 x (Conv3x3 'same' + bias + ReLU), MaxPool2x2/2 after each block."""
    y = x
    for b, (c, n) in enumerate(VGG16_CFG, 1):
        for i in range(1, n + 1):
            name = 'block%d_conv%d' % (b, i)
            y = relu(conv2d(y, p[name + '/kernel'], p[name + '/bias'], pad=(1, 1, 1, 1)))
        y = maxpool(y, 2, 2)
    return y


def vgg16_spec(in_ch=3):
    spec, cin = [], in_ch
    for b, (c, n) in enumerate(VGG16_CFG, 1):
        for i in range(1, n + 1):
            spec += [('block%d_conv%d/kernel' % (b, i), (3, 3, cin, c)), ('block%d_conv%d/bias' % (b, i), (c,))]
            cin = c
    return spec


def relu6(x):
    return np.minimum(np.maximum(x, np.asarray(0, x.dtype)), np.asarray(6, x.dtype))


def depthwise_conv2d(x, w, stride=1, pad=(0, 0, 0, 0)):
    """x [N,H,W,C], w [kh,kw,C,1] (Keras depthwise_kernel), explicit zero padding."""
    kh, kw = w.shape[:2]
    xp = np.pad(x, ((0, 0), (pad[0], pad[1]), (pad[2], pad[3]), (0, 0)))
    ho = (xp.shape[1] - kh) // stride + 1
    wo = (xp.shape[2] - kw) // stride + 1
    y = np.zeros((x.shape[0], ho, wo, x.shape[3]), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            y += xp[:, i:i + (ho - 1) * stride + 1:stride, j:j + (wo - 1) * stride + 1:stride, :] * w[i, j, :, 0]
    return y


def _correct_pad(size):
    """keras imagenet_utils.correct_pad for a 3x3 kernel: (before, after)."""
    return (0, 1) if size % 2 == 0 else (1, 1)


def mobilenetv2(x, p):
    """[N,H,W,3] -> [N,H/32 (ceil),W/32,1280] (4x4x1280 at 112 px)."""
    e = MOBILENETV2_EPS
    ph, pw = _correct_pad(x.shape[1]), _correct_pad(x.shape[2])
    y = relu6(batchnorm(conv2d(x, p['Conv1/kernel'], stride=2, pad=ph + pw), p, 'bn_Conv1', e))
    block, cin = 0, 32
    for t, c, n, s in MOBILENETV2_CFG:
        for i in range(n):
            stride = s if i == 0 else 1
            pre = 'expanded_conv' if block == 0 else 'block_%d' % block
            h = y
            if block:
                h = relu6(batchnorm(conv2d(h, p[pre + '_expand/kernel']), p, pre + '_expand_BN', e))
            if stride == 1:
                h = depthwise_conv2d(h, p[pre + '_depthwise/depthwise_kernel'], 1, (1, 1, 1, 1))
            else:
                h = depthwise_conv2d(h, p[pre + '_depthwise/depthwise_kernel'], 2,
                                     _correct_pad(h.shape[1]) + _correct_pad(h.shape[2]))
            h = relu6(batchnorm(h, p, pre + '_depthwise_BN', e))
            h = batchnorm(conv2d(h, p[pre + '_project/kernel']), p, pre + '_project_BN', e)
            y = y + h if (cin == c and stride == 1) else h
            cin = c
            block += 1
    return relu6(batchnorm(conv2d(y, p['Conv_1/kernel']), p, 'Conv_1_bn', e))


def mobilenetv2_spec(in_ch=3):
    def bn(name, c):
        return [(name + '/' + k, (c,)) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
    spec = [('Conv1/kernel', (3, 3, in_ch, 32))] + bn('bn_Conv1', 32)
    block, cin = 0, 32
    for t, c, n, s in MOBILENETV2_CFG:
        for i in range(n):
            pre = 'expanded_conv' if block == 0 else 'block_%d' % block
            mid = cin * t
            if block:
                spec += [(pre + '_expand/kernel', (1, 1, cin, mid))] + bn(pre + '_expand_BN', mid)
            spec += [(pre + '_depthwise/depthwise_kernel', (3, 3, mid, 1))] + bn(pre + '_depthwise_BN', mid)
            spec += [(pre + '_project/kernel', (1, 1, mid, c))] + bn(pre + '_project_BN', c)
            cin = c
            block += 1
    return spec + [('Conv_1/kernel', (1, 1, cin, 1280))] + bn('Conv_1_bn', 1280)


# --------------------------------------------------------------------------- heads
def head_gdc(feat, p, emd):
    """build_models_v2, deep_insight_face/networks/triplet.py:119-141:
    Conv1x1(512,no bias) -> BN -> PReLU(shared_axes=[1,2]) -> DepthwiseConv2D(kernel=H)
    -> BN -> Conv1x1(emd,no bias) -> [Dropout = identity] -> Flatten ->
    Dense(emd,no bias) -> l2_normalize(axis=1)."""
    y = conv2d(feat, p['head_conv/kernel'])
    y = batchnorm(y, p, 'head_bn1', BN_EPS_KERAS)
    y = prelu(y, p['head_prelu/alpha'])
    dw = p['head_dw/depthwise_kernel']                 # [H, W, 512, 1], kernel = full extent
    assert dw.shape[0] == y.shape[1] and dw.shape[1] == y.shape[2], (dw.shape, y.shape)
    y = np.einsum('nhwc,hwc->nc', y, dw[..., 0])[:, None, None, :]
    y = batchnorm(y, p, 'head_bn2', BN_EPS_KERAS)
    y = conv2d(y, p['head_pw/kernel'])
    y = y.reshape(y.shape[0], -1)
    y = y @ p['head_dense/kernel']
    return l2_normalize(y)


def head_sv2(feat, p, emd):
    """The siamese builder's build_models_v2, deep_insight_face/networks/siamese.py:107-128:
    Conv1x1(128, bias, relu) -> MaxPooling2D(padding='same') -> Conv1x1(128, bias, relu) ->
    MaxPooling2D(padding='same') -> BatchNormalization -> Flatten -> [Dropout] -> Dense(emd, relu)."""
    def pool_same(y):
        ph, pw = y.shape[1] % 2, y.shape[2] % 2           # 'same' for k = 2, s = 2: pad after on odd sizes
        return maxpool(y, 2, 2, pad=(0, ph, 0, pw), pad_value=-np.inf)
    y = pool_same(relu(conv2d(feat, p['sv2_conv1/kernel'], p['sv2_conv1/bias'])))
    y = pool_same(relu(conv2d(y, p['sv2_conv2/kernel'], p['sv2_conv2/bias'])))
    y = batchnorm(y, p, 'bn', BN_EPS_KERAS)
    y = y.reshape(y.shape[0], -1)
    return relu(y @ p['norm_embedding/kernel'] + p['norm_embedding/bias'])


def head_sv2_spec(cin, hw, emd):
    h2 = -(-(-(-hw // 2)) // 2)
    return ([('sv2_conv1/kernel', (1, 1, cin, 128)), ('sv2_conv1/bias', (128,)),
             ('sv2_conv2/kernel', (1, 1, 128, 128)), ('sv2_conv2/bias', (128,))] +
            [('bn/' + k, (128,)) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')] +
            [('norm_embedding/kernel', (h2 * h2 * 128, emd)), ('norm_embedding/bias', (emd,))])


def head_gdc_spec(cin, hw, emd):
    def bn(name, c):
        return [(name + '/' + k, (c,)) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
    return ([('head_conv/kernel', (1, 1, cin, 512))] + bn('head_bn1', 512) +
            [('head_prelu/alpha', (512,)), ('head_dw/depthwise_kernel', (hw, hw, 512, 1))] +
            bn('head_bn2', 512) +
            [('head_pw/kernel', (1, 1, 512, emd)), ('head_dense/kernel', (emd, emd))])


def head_v1(feat, p, emd):
    """build_models_v1, deep_insight_face/networks/triplet.py:102-117:
    Conv2x2(64,same,relu) -> MaxPool2 -> Conv2x2(32,same,relu) -> MaxPool2 ->
    Flatten -> Dense(emd, bias).  No L2-normalise (commented out at :113)."""
    def same(x, k):
        t, b = same_pad(x.shape[1], k, 1)
        l, r = same_pad(x.shape[2], k, 1)
        return (t, b, l, r)
    y = relu(conv2d(feat, p['v1_conv1/kernel'], p['v1_conv1/bias'], pad=same(feat, 2)))
    y = maxpool(y, 2, 2)
    y = relu(conv2d(y, p['v1_conv2/kernel'], p['v1_conv2/bias'], pad=same(y, 2)))
    y = maxpool(y, 2, 2)
    y = y.reshape(y.shape[0], -1)
    return y @ p['embeddings/kernel'] + p['embeddings/bias']


def head_v1_spec(cin, hw, emd):
    h2 = (hw // 2) // 2
    return [('v1_conv1/kernel', (2, 2, cin, 64)), ('v1_conv1/bias', (64,)),
            ('v1_conv2/kernel', (2, 2, 64, 32)), ('v1_conv2/bias', (32,)),
            ('embeddings/kernel', (h2 * h2 * 32, emd)), ('embeddings/bias', (emd,))]


# --------------------------------------------------------------------------- IResNet
IRESNET_LAYERS = {'iresnet50': (3, 4, 14, 3), 'iresnet100': (3, 13, 30, 3)}
IRESNET_WIDTHS = (64, 128, 256, 512)


def _iblock(x, p, name, stride, downsample):
    y = batchnorm(x, p, name + '_bn1', BN_EPS_IRESNET)
    y = conv2d(y, p[name + '_conv1/kernel'], pad=(1, 1, 1, 1))
    y = batchnorm(y, p, name + '_bn2', BN_EPS_IRESNET)
    y = prelu(y, p[name + '_prelu/alpha'])
    y = conv2d(y, p[name + '_conv2/kernel'], stride=stride, pad=(1, 1, 1, 1))
    y = batchnorm(y, p, name + '_bn3', BN_EPS_IRESNET)
    if downsample:
        sc = conv2d(x, p[name + '_down_conv/kernel'], stride=stride)
        sc = batchnorm(sc, p, name + '_down_bn', BN_EPS_IRESNET)
    else:
        sc = x
    return y + sc


def iresnet(x, p, arch='iresnet100'):
    """[N,112,112,3] -> unit-norm [N,512].  Public ArcFace IResNet: stem
    Conv3x3(64)-BN-PReLU, 4 stages of IBasicBlock (each stage stride 2, 1x1
    downsample on its first block), BN -> flatten (channel-major, as the NCHW
    original) -> FC 512 -> BN1d, then L2-normalise."""
    y = conv2d(x, p['conv1/kernel'], pad=(1, 1, 1, 1))
    y = batchnorm(y, p, 'bn1', BN_EPS_IRESNET)
    y = prelu(y, p['prelu/alpha'])
    for li, nblk in enumerate(IRESNET_LAYERS[arch]):
        for b in range(nblk):
            y = _iblock(y, p, 'layer%d_%d' % (li + 1, b), 2 if b == 0 else 1, b == 0)
    y = batchnorm(y, p, 'bn2', BN_EPS_IRESNET)
    n = y.shape[0]
    flat = np.transpose(y, (0, 3, 1, 2)).reshape(n, -1)          # c*HW + h*W + w
    y = flat @ p['fc/kernel'] + p['fc/bias']
    y = batchnorm(y, p, 'features', BN_EPS_IRESNET)
    return l2_normalize(y)


def iresnet_spec(arch='iresnet100', in_ch=3, emd=512, final_hw=7):
    def bn(name, c):
        return [(name + '/' + k, (c,)) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
    spec = [('conv1/kernel', (3, 3, in_ch, 64))] + bn('bn1', 64) + [('prelu/alpha', (64,))]
    cin = 64
    for li, nblk in enumerate(IRESNET_LAYERS[arch]):
        cout = IRESNET_WIDTHS[li]
        for b in range(nblk):
            n = 'layer%d_%d' % (li + 1, b)
            spec += bn(n + '_bn1', cin) + [(n + '_conv1/kernel', (3, 3, cin, cout))]
            spec += bn(n + '_bn2', cout) + [(n + '_prelu/alpha', (cout,))]
            spec += [(n + '_conv2/kernel', (3, 3, cout, cout))] + bn(n + '_bn3', cout)
            if b == 0:
                spec += [(n + '_down_conv/kernel', (1, 1, cin, cout))] + bn(n + '_down_bn', cout)
            cin = cout
    spec += bn('bn2', 512) + [('fc/kernel', (512 * final_hw * final_hw, emd)), ('fc/bias', (emd,))]
    spec += bn('features', emd)
    return spec


# --------------------------------------------------------------------------- ArcMargin
def arcmargin_logits(emb, weight, labels=None, s=64.0, m=0.5):
    """ArcFace (Deng et al. 2019) additive angular margin head; absent from the
    reference.  cos = e_hat . w_hat; for the label class
    phi = cos(theta + m) guarded by the usual threshold (cos > cos(pi-m) else
    cos - sin(pi-m)*m); logits = s * (onehot*phi + (1-onehot)*cos).
    ``labels`` None (inference) -> s * cos."""
    e = emb / np.sqrt(np.maximum(np.sum(emb * emb, axis=1, keepdims=True), 1e-12)).astype(emb.dtype)
    w = weight / np.sqrt(np.maximum(np.sum(weight * weight, axis=1, keepdims=True), 1e-12)).astype(weight.dtype)
    cos = e @ w.T
    if labels is None:
        return (cos * np.asarray(s, dtype=cos.dtype))
    cm, sm = math.cos(m), math.sin(m)
    th, mm = math.cos(math.pi - m), math.sin(math.pi - m) * m
    rows = np.arange(emb.shape[0])
    c = cos[rows, labels]
    sine = np.sqrt(np.clip(1.0 - c * c, 0.0, 1.0))
    phi = c * cm - sine * sm
    phi = np.where(c > th, phi, c - mm)
    out = cos.copy()
    out[rows, labels] = phi
    return (out * np.asarray(s, dtype=cos.dtype)).astype(cos.dtype)


# --------------------------------------------------------------------------- whole models
def model_spec(arch, emd=512, input_hw=112, head='v2'):
    """Ordered (name, shape) list for a full embedding model."""
    if arch == 'resnet':
        hw = input_hw
        for _ in range(2):
            hw = -(-hw // 2)
        for (_, _, s) in RESNET50V2_STACKS:
            if s == 2:
                hw = -(-hw // 2)
        spec = resnet50v2_spec()
        if head == 'v2':
            spec += head_gdc_spec(2048, hw, emd)
        elif head == 'v1':
            spec += head_v1_spec(2048, hw, emd)
        elif head == 'sv2':
            spec += head_sv2_spec(2048, hw, emd)
        return spec
    if arch in ('vgg16', 'mobilenet'):
        spec = vgg16_spec() if arch == 'vgg16' else mobilenetv2_spec()
        cfeat = 512 if arch == 'vgg16' else 1280
        hw = input_hw // 32 if arch == 'vgg16' else -(-input_hw // 32)
        if head == 'v2':
            spec += head_gdc_spec(cfeat, hw, emd)
        elif head == 'v1':
            spec += head_v1_spec(cfeat, hw, emd)
        elif head == 'sv2':
            spec += head_sv2_spec(cfeat, hw, emd)
        return spec
    if arch in IRESNET_LAYERS:
        return iresnet_spec(arch, emd=emd, final_hw=input_hw // 16)
    if arch == 'nn4':
        return nn4_spec(emd)
    raise ValueError(arch)


def embed(x, p, arch, emd=512, head='v2'):
    """x: [N,H,W,3] already scaled (predictions.py:154 multiplies by 1/255)."""
    if arch in ('resnet', 'vgg16', 'mobilenet'):
        f = {'resnet': resnet50v2, 'vgg16': vgg16, 'mobilenet': mobilenetv2}[arch](x, p)
        if head == 'v2':
            return head_gdc(f, p, emd)
        if head == 'v1':
            return head_v1(f, p, emd)
        if head == 'sv2':
            return head_sv2(f, p, emd)
        if head == 'v3':
            return f
        raise ValueError(head)
    if arch == 'nn4':
        return nn4_small2(x, p)
    return iresnet(x, p, arch)


def cast_params(p, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in p.items()}


# --------------------------------------------------------------------------- NN4.small2 (OpenFace)
# deep_insight_face/networks/inceptionv3.py:93-309 -- the one CNN fully defined inside the
# reference repository.  TensorFlow is not installed, so the Keras / tf.nn primitives it calls
# are restated from their public definitions: Conv2D (bias, VALID on an explicitly padded map),
# BatchNormalization(epsilon=1e-5) at inference, tf.nn.lrn(depth_radius=5, bias=1, alpha, beta),
# MaxPooling2D / AveragePooling2D (VALID).  Still "parity unpinned": no TF run pins these.
NN4_EPS = 1e-5


def lrn(x, depth_radius=5, bias=1.0, alpha=1e-4, beta=0.75):
    """tf.nn.lrn: x / (bias + alpha * sum_{|j-c|<=r} x_j^2) ** beta over the channel axis
    (inceptionv3.py:95: LRN2D = tf.nn.lrn(x, alpha=1e-4, beta=0.75))."""
    sq = x * x
    c = x.shape[-1]
    pad = np.pad(sq, [(0, 0)] * (x.ndim - 1) + [(depth_radius, depth_radius)])
    s = np.zeros_like(x)
    for j in range(2 * depth_radius + 1):
        s = s + pad[..., j:j + c]
    return x / np.power(np.asarray(bias, x.dtype) + np.asarray(alpha, x.dtype) * s, np.asarray(beta, x.dtype))


def avgpool(x, k, stride):
    n, h, w, c = x.shape
    ho, wo = (h - k) // stride + 1, (w - k) // stride + 1
    s = x.strides
    win = np.lib.stride_tricks.as_strided(x, shape=(n, ho, wo, k, k, c),
                                          strides=(s[0], s[1] * stride, s[2] * stride, s[1], s[2], s[3]),
                                          writeable=False)
    return win.mean(axis=(3, 4), dtype=x.dtype)


def _nn4_cbr(x, p, conv, bn, stride=1, pad=0):
    y = conv2d(x, p[conv + '/kernel'], p[conv + '/bias'], stride=stride, pad=(pad, pad, pad, pad))
    return relu(batchnorm(y, p, bn, NN4_EPS))


def _nn4_branch(x, p, layer, stride, pad):
    """conv2d_bn with two convolutions (inceptionv3.py:312-335)."""
    y = _nn4_cbr(x, p, layer + '_conv1', layer + '_bn1')
    return _nn4_cbr(y, p, layer + '_conv2', layer + '_bn2', stride=stride, pad=pad)


def _zpad(x, t, b, l, r):
    return np.pad(x, ((0, 0), (t, b), (l, r), (0, 0)))


def _l2pool(x):
    """x**2 -> AveragePooling2D(3, strides 3) -> * 9 -> sqrt (inceptionv3.py:160-163)."""
    return np.sqrt(avgpool(x * x, 3, 3) * np.asarray(9, x.dtype))


def nn4_small2(x, p):
    """[N,96,96,3] -> unit-norm [N,emd].  inceptionv3.py:93-309."""
    y = _nn4_cbr(x, p, 'conv1', 'bn1', stride=2, pad=3)
    y = maxpool(y, 3, 2, pad=(1, 1, 1, 1))
    y = lrn(y)
    y = _nn4_cbr(y, p, 'conv2', 'bn2')
    y = _nn4_cbr(y, p, 'conv3', 'bn3', pad=1)
    y = lrn(y)
    y = maxpool(y, 3, 2, pad=(1, 1, 1, 1))
    # 3a
    b3 = _nn4_branch(y, p, 'inception_3a_3x3', 1, 1)
    b5 = _nn4_branch(y, p, 'inception_3a_5x5', 1, 2)
    bp = _zpad(_nn4_cbr(maxpool(y, 3, 2), p, 'inception_3a_pool_conv', 'inception_3a_pool_bn'), 3, 4, 3, 4)
    b1 = _nn4_cbr(y, p, 'inception_3a_1x1_conv', 'inception_3a_1x1_bn')
    y = np.concatenate([b3, b5, bp, b1], axis=3)
    # 3b
    b3 = _nn4_branch(y, p, 'inception_3b_3x3', 1, 1)
    b5 = _nn4_branch(y, p, 'inception_3b_5x5', 1, 2)
    bp = _zpad(_nn4_cbr(_l2pool(y), p, 'inception_3b_pool_conv', 'inception_3b_pool_bn'), 4, 4, 4, 4)
    b1 = _nn4_cbr(y, p, 'inception_3b_1x1_conv', 'inception_3b_1x1_bn')
    y = np.concatenate([b3, b5, bp, b1], axis=3)
    # 3c
    b3 = _nn4_branch(y, p, 'inception_3c_3x3', 2, 1)
    b5 = _nn4_branch(y, p, 'inception_3c_5x5', 2, 2)
    bp = _zpad(maxpool(y, 3, 2), 0, 1, 0, 1)
    y = np.concatenate([b3, b5, bp], axis=3)
    # 4a
    b3 = _nn4_branch(y, p, 'inception_4a_3x3', 1, 1)
    b5 = _nn4_branch(y, p, 'inception_4a_5x5', 1, 2)
    bp = _zpad(_nn4_cbr(_l2pool(y), p, 'inception_4a_pool_conv', 'inception_4a_pool_bn'), 2, 2, 2, 2)
    b1 = _nn4_cbr(y, p, 'inception_4a_1x1_conv', 'inception_4a_1x1_bn')
    y = np.concatenate([b3, b5, bp, b1], axis=3)
    # 4e
    b3 = _nn4_branch(y, p, 'inception_4e_3x3', 2, 1)
    b5 = _nn4_branch(y, p, 'inception_4e_5x5', 2, 2)
    bp = _zpad(maxpool(y, 3, 2), 0, 1, 0, 1)
    y = np.concatenate([b3, b5, bp], axis=3)
    # 5a
    b3 = _nn4_branch(y, p, 'inception_5a_3x3', 1, 1)
    bp = _zpad(_nn4_cbr(_l2pool(y), p, 'inception_5a_pool_conv', 'inception_5a_pool_bn'), 1, 1, 1, 1)
    b1 = _nn4_cbr(y, p, 'inception_5a_1x1_conv', 'inception_5a_1x1_bn')
    y = np.concatenate([b3, bp, b1], axis=3)
    # 5b
    b3 = _nn4_branch(y, p, 'inception_5b_3x3', 1, 1)
    bp = _zpad(_nn4_cbr(maxpool(y, 3, 2), p, 'inception_5b_pool_conv', 'inception_5b_pool_bn'), 1, 1, 1, 1)
    b1 = _nn4_cbr(y, p, 'inception_5b_1x1_conv', 'inception_5b_1x1_bn')
    y = np.concatenate([b3, bp, b1], axis=3)
    y = avgpool(y, 3, 1).reshape(y.shape[0], -1)
    y = y @ p['dense_layer/kernel'] + p['dense_layer/bias']
    return l2_normalize(y)


# conv name -> (cout, cin, kh, kw), the reference's own table (inceptionv3.py:365-403)
NN4_CONV_SHAPE = {
    'conv1': (64, 3, 7, 7), 'conv2': (64, 64, 1, 1), 'conv3': (192, 64, 3, 3),
    'inception_3a_1x1_conv': (64, 192, 1, 1), 'inception_3a_pool_conv': (32, 192, 1, 1),
    'inception_3a_5x5_conv1': (16, 192, 1, 1), 'inception_3a_5x5_conv2': (32, 16, 5, 5),
    'inception_3a_3x3_conv1': (96, 192, 1, 1), 'inception_3a_3x3_conv2': (128, 96, 3, 3),
    'inception_3b_3x3_conv1': (96, 256, 1, 1), 'inception_3b_3x3_conv2': (128, 96, 3, 3),
    'inception_3b_5x5_conv1': (32, 256, 1, 1), 'inception_3b_5x5_conv2': (64, 32, 5, 5),
    'inception_3b_pool_conv': (64, 256, 1, 1), 'inception_3b_1x1_conv': (64, 256, 1, 1),
    'inception_3c_3x3_conv1': (128, 320, 1, 1), 'inception_3c_3x3_conv2': (256, 128, 3, 3),
    'inception_3c_5x5_conv1': (32, 320, 1, 1), 'inception_3c_5x5_conv2': (64, 32, 5, 5),
    'inception_4a_3x3_conv1': (96, 640, 1, 1), 'inception_4a_3x3_conv2': (192, 96, 3, 3),
    'inception_4a_5x5_conv1': (32, 640, 1, 1), 'inception_4a_5x5_conv2': (64, 32, 5, 5),
    'inception_4a_pool_conv': (128, 640, 1, 1), 'inception_4a_1x1_conv': (256, 640, 1, 1),
    'inception_4e_3x3_conv1': (160, 640, 1, 1), 'inception_4e_3x3_conv2': (256, 160, 3, 3),
    'inception_4e_5x5_conv1': (64, 640, 1, 1), 'inception_4e_5x5_conv2': (128, 64, 5, 5),
    'inception_5a_3x3_conv1': (96, 1024, 1, 1), 'inception_5a_3x3_conv2': (384, 96, 3, 3),
    'inception_5a_pool_conv': (96, 1024, 1, 1), 'inception_5a_1x1_conv': (256, 1024, 1, 1),
    'inception_5b_3x3_conv1': (96, 736, 1, 1), 'inception_5b_3x3_conv2': (384, 96, 3, 3),
    'inception_5b_pool_conv': (96, 736, 1, 1), 'inception_5b_1x1_conv': (256, 736, 1, 1),
}


def nn4_spec(emd=128):
    spec = []
    for conv, (co, ci, kh, kw) in NN4_CONV_SHAPE.items():
        spec += [(conv + '/kernel', (kh, kw, ci, co)), (conv + '/bias', (co,))]
        bn = conv.replace('_conv', '_bn') if conv.startswith('inception') else conv.replace('conv', 'bn')
        spec += [(bn + '/' + k, (co,)) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
    spec += [('dense_layer/kernel', (736, emd)), ('dense_layer/bias', (emd,))]
    return spec
