"""The NumPy oracle of the embedding networks against an independently written
torch-CPU implementation (oracle/torch_nets.py), plus shape / parameter-count facts from
the public model definitions, plus the library's parameter table.  No GPU."""
import numpy as np
import pytest

from oracle import torch_nets as torch_ref
from oracle import nets


def crops(n, hw=112, seed=1234):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (n, hw, hw, 3), dtype=np.uint8).astype(np.float32) / np.float32(255.0))


def synth(arch, emd, head='v2', hw=112):
    from deep_insight_face.networks.weights import synth_params
    return synth_params(nets.model_spec(arch, emd, hw, head))


def cosine_gap(a, b):
    a = a.reshape(a.shape[0], -1).astype(np.float64)
    b = b.reshape(b.shape[0], -1).astype(np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


def test_param_counts_and_shapes():
    n = lambda spec: sum(int(np.prod(s)) for _, s in spec)   # noqa: E731
    assert n(nets.resnet50v2_spec()) == 23_564_800            # keras ResNet50V2(include_top=False)
    assert n(nets.model_spec('iresnet100', 512)) == 65_225_792
    assert n(nets.model_spec('iresnet50', 512)) == 43_628_992
    assert n(nets.vgg16_spec()) == 14_714_688                 # keras VGG16(include_top=False)
    assert n(nets.mobilenetv2_spec()) == 2_257_984            # keras MobileNetV2(alpha=1.0, include_top=False)
    assert nets.mobilenetv2(crops(1), synth('mobilenet', 512, 'v3')).shape == (1, 4, 4, 1280)
    assert nets.vgg16(crops(1), synth('vgg16', 512, 'v3')).shape == (1, 3, 3, 512)
    p = synth('resnet', 512, 'v3')
    f = nets.embed(crops(1), p, 'resnet', head='v3')
    assert f.shape == (1, 4, 4, 2048)
    f96 = nets.resnet50v2(crops(1, 96), p)
    assert f96.shape == (1, 3, 3, 2048)


def test_library_param_table_matches_oracle():
    from deep_insight_face.networks.triplet import DifEmbedder
    for arch, head, emd in (('resnet', 'v2', 512), ('resnet', 'v1', 128), ('resnet', 'v3', 512),
                            ('iresnet50', 'v2', 512), ('iresnet100', 'v2', 512), ('vgg16', 'v2', 512),
                            ('mobilenet', 'v1', 128), ('mobilenet', 'v2', 512), ('mobilenet', 'v3', 512), ('resnet', 'sv2', 128),
                            ('vgg16', 'sv2', 64)):
        m = DifEmbedder(arch, head, emd, (112, 112, 3))
        assert dict(m.param_spec()) == dict(nets.model_spec(arch, emd, 112, head)), (arch, head)
        m.close()
    m = DifEmbedder('resnet', 'v2', 512, (112, 112, 3))
    assert abs(m.flops_per_image / 2 - 0.9466e9) < 1e6         # 0.947 GMAC
    m = DifEmbedder('iresnet100', 'v2', 512, (112, 112, 3))
    assert abs(m.flops_per_image / 2 - 12.09e9) < 1e7          # 12.09 GMAC
    with pytest.raises(ValueError):
        DifEmbedder('resnet18', 'v2', 512, (112, 112, 3))


@pytest.mark.parametrize('arch,head,emd,n', [('resnet', 'v2', 512, 2), ('resnet', 'v1', 128, 2),
                                             ('resnet', 'v3', 512, 1), ('iresnet50', 'v2', 512, 1),
                                             ('vgg16', 'v2', 512, 1), ('mobilenet', 'v2', 512, 2),
                                             ('mobilenet', 'v3', 512, 1), ('resnet', 'sv2', 128, 2),
                                             ('vgg16', 'sv2', 128, 1)])
def test_oracle_vs_torch(arch, head, emd, n):
    p = synth(arch, emd, head)
    x = crops(n)
    a = nets.embed(x, p, arch, emd, head)
    b = torch_ref.embed(x, p, arch, head)
    assert a.shape == b.shape and a.dtype == np.float32
    assert np.all(np.isfinite(a))
    assert cosine_gap(a, b).max() < 1e-5
    np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * np.abs(b).max())


def test_oracle_float64_arbiter():
    """float32 oracle vs the same restatement in float64: rounding noise only."""
    p = synth('resnet', 512)
    x = crops(2)
    a = nets.embed(x, p, 'resnet', 512, 'v2')
    b = nets.embed(x.astype(np.float64), nets.cast_params(p, np.float64), 'resnet', 512, 'v2')
    assert cosine_gap(a, b).max() < 1e-6


def test_arcmargin_oracle():
    rng = np.random.default_rng(0)
    e = rng.standard_normal((6, 64)).astype(np.float32)
    w = rng.standard_normal((10, 64)).astype(np.float32)
    lab = np.array([0, 3, 9, 1, 1, 5])
    plain = nets.arcmargin_logits(e, w)
    marg = nets.arcmargin_logits(e, w, lab)
    cos = plain / 64.0
    assert np.all(np.abs(cos) <= 1 + 1e-6)
    off = np.ones_like(plain, dtype=bool)
    off[np.arange(6), lab] = False
    assert np.array_equal(plain[off], marg[off])
    th = np.arccos(np.clip(cos[np.arange(6), lab], -1, 1))
    want = np.where(cos[np.arange(6), lab] > np.cos(np.pi - 0.5), np.cos(th + 0.5),
                    cos[np.arange(6), lab] - np.sin(np.pi - 0.5) * 0.5) * 64
    np.testing.assert_allclose(marg[np.arange(6), lab], want, atol=1e-4)
