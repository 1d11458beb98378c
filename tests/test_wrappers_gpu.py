"""The reference-signature wrappers (predictions.py, api.py, evaluation/evals.py,
networks/utils.py) on top of the HIP path, against the oracle's restatement of the same
reference lines."""
import numpy as np
import pytest

from oracle import distance as od
from oracle import nets

pytestmark = pytest.mark.gpu


def crops_u8(n, hw=112, seed=1234):
    return np.random.default_rng(seed).integers(0, 256, (n, hw, hw, 3), dtype=np.uint8)


def cosine_gap(a, b):
    a = a.reshape(a.shape[0], -1).astype(np.float64)
    b = b.reshape(b.shape[0], -1).astype(np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


@pytest.fixture(scope='module')
def model():
    from deep_insight_face.networks.triplet import bottleneck_network
    m = bottleneck_network('resnet', 128, (112, 112, 3), max_batch=8)('v2').init_synthetic(7)
    yield m
    m.close()


def test_triplet_prediction(cuda, model, capsys):
    from deep_insight_face.predictions import TripletPrediction
    p = model.get_weights()
    tp = TripletPrediction(model, img_size=(112, 112))
    assert TripletPrediction(model, img_size=(112, 112)) is tp          # singleton per class (predictions.py:27-31)
    img = crops_u8(1, seed=3)[0]
    emb = tp._embedding(img)                                             # image * 1/255 -> predict_on_batch
    want = nets.embed(img[None].astype(np.float32) / np.float32(255), p, 'resnet', 128, 'v2')
    assert emb.shape == (1, 128) and cosine_gap(emb, want).max() < 1e-5
    with pytest.raises(AssertionError, match='Invalid image format'):
        tp._embedding('not-an-array')
    db = {'alice': want[0], 'bob': -want[0]}
    d, ok = tp.verify(img, 'alice', db, threshold=0.7)
    assert ok and d < 1e-2
    d, ok = tp.verify(img, 'bob', db, threshold=0.7)
    assert (not ok) and abs(d - 2.0) < 1e-3
    out = capsys.readouterr().out
    assert "It's alice" in out and "It's not bob" in out
    batch = tp._embedding_batch(list(crops_u8(3, seed=4)))
    assert batch.shape == (3, 128)


def test_siamese_prediction_applies_vgg_preprocess(cuda, model):
    from deep_insight_face.predictions import SiamesePrediction
    p = model.get_weights()
    sp = SiamesePrediction(model, img_size=(112, 112))
    img = crops_u8(1, seed=8)[0]
    emb = sp._embedding(img)
    x = img[None].astype(np.float32) / np.float32(255)
    mean = np.array([103.939, 116.779, 123.68], dtype=np.float32)
    want = nets.embed(x[..., ::-1] - mean, p, 'resnet', 128, 'v2')      # keras vgg16.preprocess_input, caffe mode
    assert cosine_gap(emb, want).max() < 1e-5
    db = {'carol': [want[0], want[0]]}
    d, ok = sp.verify(img, 'carol', db, threshold=0.3)
    assert ok and d < 1e-2


def test_api_distances(cuda):
    from deep_insight_face import api
    from deep_insight_face.networks import utils as nu
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(128).astype(np.float32), rng.standard_normal(128).astype(np.float32)
    assert api.face_distance([], b).shape == (0,)
    np.testing.assert_allclose(api.face_distance(a, b), od.face_distance(a, b), rtol=1e-6)
    A = rng.standard_normal((5, 128)).astype(np.float32)
    np.testing.assert_allclose(api.face_distance(A, b), od.face_distance(A, b), rtol=1e-5)   # axis-0 quirk kept
    for scale in (0.01, 1.0):
        d, pr = api.compare_faces([a * scale], [b * scale])
        od_, opr = od.compare_faces([a * scale], [b * scale])
        np.testing.assert_allclose([d, pr], [od_, opr], rtol=1e-5)
    np.testing.assert_allclose(nu.distance(a, b), od.sq_l2(a, b), rtol=1e-6)
    assert nu.distance_to_proba(0.5) == od.distance_to_proba(0.5)
    assert nu.gaussian_kernel_dist_to_prob(0.5, 2.0) == od.gaussian_kernel_dist_to_prob(0.5, 2.0)
    from deep_insight_face.exceptions import FaceRecognitionException
    with pytest.raises(FaceRecognitionException):
        api.face_encodings(np.zeros((112, 112, 3), np.uint8), (112, 112, 3))    # detector is out of scope


def test_api_face_encodings_with_registered_model(cuda, model):
    from deep_insight_face import api
    api.set_face_recognition_model(model, 'triplet')
    img = crops_u8(1, seed=5)[0]
    thumb, enc = api.face_encodings(img, (112, 112, 3), detect_and_crop=False)
    assert thumb[0] is img and enc.shape == (1, 128)
    np.testing.assert_allclose(np.linalg.norm(enc), 1.0, atol=1e-5)


def test_batched_eval_loop(cuda, model):
    """evaluation/evals.py:53-59: predict_on_batch per batch scattered into one array."""
    from deep_insight_face.evaluation.evals import embed_batches
    x = crops_u8(10, seed=6).astype(np.float32) / np.float32(255)
    batches = [(x[0:4], np.arange(0, 4)), (x[4:8], np.arange(4, 8)), (x[8:10], np.arange(8, 10))]
    emb, lab = embed_batches(model, batches, 10, 128)
    assert emb.shape == (10, 128) and emb.dtype == np.float64
    whole = model.predict_on_batch(x[:8])
    assert cosine_gap(emb[:8], whole).max() < 1e-6
    with pytest.raises(AssertionError, match='Wrong labels'):
        embed_batches(model, [(x[0:4], np.array([0, 1, 2, 5]))], 4, 128)


def test_two_lane_forward_matches_single_lane(cuda, monkeypatch):
    """With two lanes batches >= 64 are split over two lanes (caller's stream + an internal HIP stream, separate
    activation buffers): results must equal the single-lane forward up to float32 rounding, for
    ragged splits too, and a few rows are checked against the oracle."""
    from deep_insight_face.networks.triplet import DifEmbedder
    n = 131
    x = crops_u8(n, seed=11).astype(np.float32) / np.float32(255)
    monkeypatch.setenv('DIF_STREAMS', '2')                         # (the default picks by work per launch)
    two = DifEmbedder('resnet', 'v2', 128, (112, 112, 3), max_batch=160).init_synthetic(7)
    two._finalize()                                                # the lane count is fixed at finalize
    monkeypatch.setenv('DIF_STREAMS', '1')
    one = DifEmbedder('resnet', 'v2', 128, (112, 112, 3), max_batch=160).init_synthetic(7)
    one._finalize()
    monkeypatch.delenv('DIF_STREAMS')
    a = two.predict_on_batch(x)
    b = one.predict_on_batch(x)
    assert a.shape == (n, 128) and np.all(np.isfinite(a))
    assert cosine_gap(a, b).max() < 1e-6
    assert np.array_equal(two.predict_on_batch(x), a)             # deterministic
    rows = [0, 65, 66, 130]                                       # both lanes, both ends
    want = nets.embed(x[rows], two.get_weights(), 'resnet', 128, 'v2')
    assert cosine_gap(a[rows], want).max() < 1e-5
    two.close()
    one.close()


def test_siamese_module(cuda):
    """networks/siamese.py mirror: euclidean_distance = sqrt(max(sum sq diff, eps)) keepdims, the
    two-tower model = distance between the shared tower's embeddings, the siamese v2 head."""
    from deep_insight_face.networks import siamese
    rng = np.random.default_rng(3)
    a = rng.standard_normal((7, 128)).astype(np.float32)
    b = rng.standard_normal((7, 128)).astype(np.float32)
    b[2] = a[2]
    want = np.sqrt(np.maximum(((a - b) ** 2).sum(1, keepdims=True), 1e-7))
    got = siamese.euclidean_distance((a, b))
    assert got.shape == (7, 1)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    assert siamese.eucl_dist_output_shape(((7, 128), (7, 128))) == (7, 1)
    model, base = siamese.buildin_models(128, (112, 112, 3), max_batch=4)
    base.init_synthetic(9)
    x1, x2 = crops_u8(3, seed=1).astype(np.float32) / 255, crops_u8(3, seed=2).astype(np.float32) / 255
    d = model.predict([x1, x2])
    e1, e2 = base.predict_on_batch(x1), base.predict_on_batch(x2)
    np.testing.assert_allclose(d, np.sqrt(np.maximum(((e1 - e2) ** 2).sum(1, keepdims=True), 1e-7)), rtol=1e-4, atol=1e-5)
    assert float(model.predict([x1, x1]).max()) < 1e-3
    v2 = siamese.bottleneck_network('resnet', 64, (112, 112, 3), max_batch=2)('v2')
    assert v2.output_shape == (64,) and dict(v2.param_spec())['norm_embedding/kernel'] == (128, 64)
    with pytest.raises(AssertionError, match='Invalid bottleneck network'):
        siamese.bottleneck_network('inception')
    base.close()
    v2.close()


def test_prediction_wrappers_keep_the_callers_transform(cuda):
    """_embedding must put back the transform the caller had set on the shared model (bench.py and
    FramePipeline run uint8 crops with scale 1/255), not reset it to the identity."""
    from deep_insight_face.networks.triplet import DifEmbedder
    from deep_insight_face.predictions import SiamesePrediction, TripletPrediction
    model = DifEmbedder('resnet', 'v2', 128, (112, 112, 3), max_batch=4).init_synthetic(3)
    model.set_input_transform(scale=1 / 255.)
    u8 = crops_u8(2, seed=9)
    before = model.predict_on_batch(u8)
    TripletPrediction(model, img_size=(112, 112))._embedding(u8[0])
    assert model._transform[0] == pytest.approx(1 / 255.)
    assert np.array_equal(model.predict_on_batch(u8), before)
    SiamesePrediction(model, img_size=(112, 112))._embedding(u8[1])
    assert model._transform == (pytest.approx(1 / 255.), (0.0, 0.0, 0.0), False, False)
    assert np.array_equal(model.predict_on_batch(u8), before)
    model.close()


def test_triplet_evaluate_object(cuda, tmp_path, capsys):
    """evaluation/evals.py:19-78 mirror: TripletEvaluate(emd_model, image_paths, pairs)(batch_size, nrof_folds,
    distance_metric, subtract_mean) runs the batched embedding loop and the LFW-protocol statistics; checked
    against the oracle's calculate_roc on the same embeddings, with files on disk and with injected batches."""
    from PIL import Image
    from deep_insight_face.evaluation.evals import TripletEvaluate
    from deep_insight_face.networks.triplet import DifEmbedder
    from oracle import evalproto as oe
    model = DifEmbedder('resnet', 'v1', 128, (112, 112, 3), max_batch=16).init_synthetic(5)
    rng = np.random.default_rng(12)
    npairs = 30
    base = rng.integers(0, 256, (npairs, 112, 112, 3), dtype=np.uint8)
    issame = rng.random(npairs) < 0.5
    imgs = []
    for i in range(npairs):
        other = np.clip(base[i].astype(np.int64) + rng.integers(-20, 21, base[i].shape), 0, 255).astype(np.uint8) \
            if issame[i] else rng.integers(0, 256, base[i].shape, dtype=np.uint8)
        imgs += [base[i], other]
    paths = []
    for k, im in enumerate(imgs):
        p = str(tmp_path / ('img%03d.png' % k))
        Image.fromarray(im).save(p)
        paths.append(p)
    ev = TripletEvaluate(model, paths, issame)
    res = ev(batch_size=16, nrof_folds=5, distance_metric=1, subtract_mean=False)
    out = capsys.readouterr().out
    assert 'Accuracy:' in out and 'Validation rate:' in out and 'Area Under Curve (AUC):' in out
    x = np.stack(imgs).astype(np.float32) / np.float32(255)
    emb = np.concatenate([model.predict_on_batch(x[s:s + 16]) for s in range(0, 2 * npairs, 16)])
    assert cosine_gap(res['embeddings'], emb).max() < 1e-6
    tpr, fpr, acc, f1 = oe.calculate_roc(np.arange(0, 4, 0.01), res['embeddings'][0::2], res['embeddings'][1::2], issame, 5, 1, False)
    np.testing.assert_allclose(res['tpr'], tpr, atol=1e-12)
    np.testing.assert_allclose(res['fpr'], fpr, atol=1e-12)
    np.testing.assert_allclose(res['accuracy'], acc, atol=1e-12)
    # injected batches (the reference's generator contract: y = running image index)
    def batches(bs):
        for s in range(0, 2 * npairs, bs):
            yield x[s:s + bs], np.arange(s, min(s + bs, 2 * npairs))
    res2 = TripletEvaluate(model, paths, issame, batches=batches)(16, 5, 1)
    np.testing.assert_allclose(res2['accuracy'], res['accuracy'])
    with pytest.raises(AssertionError, match='Wrong labels'):
        TripletEvaluate(model, paths, issame, batches=lambda bs: [(x[:16], np.arange(1, 17))])(16, 5, 1)
    model.close()


def test_wrapper_resizes_off_size_crops_on_the_device(cuda, model):
    """a crop that is not at the embedder's input size is resampled by the library's
    area-coverage kernel (dif_area_resize), which is what the reference's
    ``cv2.resize(image, size, interpolation=Image.BICUBIC)`` selects (predictions.py:93,154: PIL's
    BICUBIC constant 3 == cv2.INTER_AREA).  Checked against oracle/imageops.area_resize -- PARITY UNPINNED:
    cv2 is absent from the image; the oracle restates cv2's uint8 paths operation by operation from OpenCV's
    published source (2 x 2 and other integer ratios, the float32 DecimateAlpha tables for fractional ratios with
    round-half-to-even, the 11-bit fixed-point linear path with area-mode coefficients when an axis enlarges) and
    the device kernel performs the same operations in the same order: the two must be EQUAL, every pixel
    (round 3 compared against a float64 coverage mean and tolerated up to 10 % of pixels one level off)."""
    import torch
    from deep_insight_face import predictions
    from deep_insight_face.predictions import TripletPrediction
    from oracle import imageops as oi
    rng = np.random.default_rng(21)
    p = model.get_weights()
    tp = TripletPrediction(model, img_size=(112, 112))
    for shape in ((224, 224), (336, 336), (160, 160), (250, 250), (300, 180), (117, 131), (112, 131), (96, 96), (64, 80),
                  (200, 100), (113, 448)):
        img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
        got = predictions._resize(img, (112, 112))
        assert torch.is_tensor(got) and got.is_cuda and got.dtype == torch.uint8 and tuple(got.shape) == (112, 112, 3)
        want = oi.area_resize(img, 112)
        assert np.array_equal(got.cpu().numpy(), want), (shape, int(np.abs(got.cpu().numpy().astype(np.int32) - want).max()),
                                                         float((got.cpu().numpy() != want).mean()))
        if shape in ((336, 336), (112, 131), (200, 100), (113, 448)):
            continue
        emb = tp._embedding(img)
        assert isinstance(emb, np.ndarray) and emb.shape == (1, 128)
        ref = nets.embed(want[None].astype(np.float32) / np.float32(255), p, 'resnet', 128, 'v2')
        assert cosine_gap(emb, ref).max() < 1e-5
    # non-square target (cv2 takes (width, height))
    img = rng.integers(0, 256, (200, 100, 3), dtype=np.uint8)
    got = predictions._resize(img, (50, 100))
    assert tuple(got.shape) == (100, 50, 3)
    assert np.array_equal(got.cpu().numpy(), oi.area_resize(img, (50, 100)))
    # mixed batch: one crop at size, one not
    out = tp._embedding_batch([rng.integers(0, 256, (112, 112, 3), dtype=np.uint8),
                               rng.integers(0, 256, (150, 150, 3), dtype=np.uint8)])
    assert out.shape == (2, 128)
    with pytest.raises(ValueError):
        predictions._resize(rng.random((150, 150, 3)), (112, 112))       # float images are not resized
