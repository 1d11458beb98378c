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


def test_openface_csv_weights_round_trip(tmp_path):
    """The reference's CSV weight directory (inceptionv3.py:28-60): written here the way OpenFace ships it
    ([cout, cin, kh, kw]-ordered flat conv kernels, [128, 736] dense), read by the mirror of the reference's
    loader, compared with the source parameters; then through InceptionNetwork._load_weights."""
    import os
    from deep_insight_face.networks import inceptionv3 as inc
    p = synth()
    d = str(tmp_path)
    for name, a in p.items():
        layer, leaf = name.rsplit('/', 1)
        if leaf == 'kernel' and a.ndim == 4:
            np.savetxt(os.path.join(d, layer + '_w.csv'), np.transpose(a, (3, 2, 0, 1)).reshape(1, -1), delimiter=',', fmt='%.9g')
        elif leaf == 'kernel':
            np.savetxt(os.path.join(d, 'dense_w.csv'), a.T.reshape(1, -1), delimiter=',', fmt='%.9g')
        else:
            suffix = {'bias': '_b', 'gamma': '_w', 'beta': '_b', 'moving_mean': '_m', 'moving_variance': '_v'}[leaf]
            stem = 'dense' if layer == 'dense_layer' else layer
            np.savetxt(os.path.join(d, stem + suffix + '.csv'), a.reshape(1, -1), delimiter=',', fmt='%.9g')
    w = inc.load_weights(d)
    assert len(w) == 37 + 37 + 1
    assert w['conv1'][0].shape == (7, 7, 3, 64) and np.array_equal(w['conv1'][0], p['conv1/kernel'])
    assert np.array_equal(w['dense_layer'][0], p['dense_layer/kernel'])
    assert np.array_equal(w['inception_4e_5x5_bn2'][3], p['inception_4e_5x5_bn2/moving_variance'])
    net = inc.InceptionNetwork((96, 96, 3), 128)
    net._load_weights(d)
    got = net.get_weights()
    assert set(got) == set(p) and all(np.array_equal(got[k], p[k]) for k in p)
