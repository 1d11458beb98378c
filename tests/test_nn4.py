"""OpenFace NN4.small2 (deep_insight_face/networks/inceptionv3.py:93-309): the NumPy oracle
against an independently written torch-CPU implementation (CPU), and the HIP forward against
the oracle (GPU)."""
import numpy as np
import pytest

from oracle import torch_nets as torch_ref
from oracle import nets


def crops96(n, seed=1234):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, 96, 96, 3), dtype=np.uint8).astype(np.float32) / np.float32(255.0)


def synth(emd=128):
    from deep_insight_face.networks.weights import synth_params
    return synth_params(nets.model_spec('nn4', emd, 96))


def cosine_gap(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


def test_nn4_shapes_follow_the_reference_table():
    # the reference's own conv_shape table (inceptionv3.py:365-403) as [cout, cin, kh, kw]
    spec = dict(nets.nn4_spec(128))
    assert spec['conv1/kernel'] == (7, 7, 3, 64)
    assert spec['inception_3a_5x5_conv2/kernel'] == (5, 5, 16, 32)
    assert spec['inception_4e_3x3_conv2/kernel'] == (3, 3, 160, 256)
    assert spec['inception_5b_1x1_conv/kernel'] == (1, 1, 736, 256)
    assert spec['dense_layer/kernel'] == (736, 128)
    assert len([k for k in spec if k.endswith('/kernel')]) == 38     # 37 convolutions + dense


def test_nn4_library_table_matches_oracle():
    from deep_insight_face.networks.inceptionv3 import InceptionNetwork
    net = InceptionNetwork((96, 96, 3), 128)
    assert dict(net.param_spec()) == dict(nets.model_spec('nn4', 128, 96))
    with pytest.raises(AssertionError, match='Invalid Input shape'):
        InceptionNetwork((112, 112, 3), 128)          # inceptionv3.py:66


def test_nn4_oracle_vs_torch():
    p = synth()
    x = crops96(3)
    a = nets.embed(x, p, 'nn4', 128)
    b = torch_ref.embed_nn4(x, p)
    assert a.shape == (3, 128) and np.all(np.isfinite(a))
    np.testing.assert_allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-5)
    assert cosine_gap(a, b).max() < 1e-5
    np.testing.assert_allclose(a, b, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 7])
def test_nn4_gpu_vs_oracle(cuda, n):
    from deep_insight_face.networks.inceptionv3 import InceptionNetwork
    net = InceptionNetwork((96, 96, 3), 128, max_batch=4)
    net.init_synthetic(2024)
    p = net.get_weights()
    x = crops96(n, seed=5)
    got = net.predict_on_batch(x)
    want = nets.embed(x, p, 'nn4', 128)
    assert got.shape == (n, 128) and got.dtype == np.float32
    assert cosine_gap(got, want).max() < 1e-5
    np.testing.assert_allclose(got, want, atol=2e-4)
    assert np.array_equal(net(x), got)          # __call__ passthrough, deterministic
