"""Seeded inputs shared by tests/gen_golden.py (which feeds them to the reference) and
the tests (which feed them to the oracle and to the HIP path).  SURVEY.md section 8(d)
defines the synthetic data: gallery = L2-normalised standard-normal rows (seed 7),
probes = gallery rows + 0.05*noise, re-normalised."""
import hashlib

import numpy as np

D = 512


def digest(*arrays):
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


def _unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def pair_inputs(n=64, seed=1234):
    rng = np.random.default_rng(seed)
    e1 = rng.standard_normal((n, D)).astype(np.float32)
    e2 = (e1 + 0.7 * rng.standard_normal((n, D))).astype(np.float32)
    return e1, e2


def gallery(g, seed=7, d=D):
    rng = np.random.default_rng(seed)
    return _unit(rng.standard_normal((g, d)))


def probes_from(gal, b, seed=11, noise=0.05):
    rng = np.random.default_rng(seed)
    pick = rng.permutation(gal.shape[0])[:b]
    p = gal[pick] + noise * rng.standard_normal((b, gal.shape[1])).astype(np.float32)
    return _unit(p), pick.astype(np.int64)


def match_inputs(b=8, g=1000):
    gal = gallery(g)
    p, _ = probes_from(gal, b)
    return p, gal


def match_tie_inputs(b=8, g=600):
    """Duplicated gallery rows: np.argmin must return the FIRST of the equal minima."""
    gal = gallery(g, seed=21)
    gal[300:340] = gal[100:140]          # exact copies at higher indices
    gal[500:520] = gal[100:120]          # and a third copy
    rng = np.random.default_rng(22)
    pick = np.array([100, 105, 119, 120, 139, 301, 510, 7])
    p = _unit(gal[pick] + 0.05 * rng.standard_normal((b, D)).astype(np.float32))
    return p, gal


def match_unnormalised_inputs(b=8, g=500):
    """Rows of very different length: cosine must normalise, squared L2 must not."""
    rng = np.random.default_rng(31)
    gal = rng.standard_normal((g, D)).astype(np.float32) * rng.uniform(0.1, 10.0, (g, 1)).astype(np.float32)
    pick = rng.permutation(g)[:b]
    p = (gal[pick] * rng.uniform(0.5, 2.0, (b, 1)) + 0.3 * rng.standard_normal((b, D))).astype(np.float32)
    return p, gal


def vector_inputs(seed=5):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(128).astype(np.float32), rng.standard_normal(128).astype(np.float32)


def roc_inputs(npairs=300, d=128, seed=99):
    rng = np.random.default_rng(seed)
    issame = rng.random(npairs) < 0.5
    e1 = _unit(rng.standard_normal((npairs, d)))
    far = _unit(rng.standard_normal((npairs, d)))
    near = _unit(e1 + 0.6 * _unit(rng.standard_normal((npairs, d))))
    e2 = np.where(issame[:, None], near, far).astype(np.float32)
    return e1, e2, issame


def match_near_tie_inputs(b=48, g=4096, seed=41):
    """Near-ties (VERDICT r01 weak #3 / ADVICE): gallery rows that are 1-ulp, 1e-7-scale and 1e-4-scale
    perturbations of each other and of the probes, exact duplicates among them, at shuffled positions, in a
    gallery of otherwise unrelated rows; a second half of the gallery is UNNORMALISED with large |g|^2 (the
    squared-L2 search key |g|^2 - 2 q.g cancels badly there).  Returns (probes [b, D], gallery [g, D])."""
    rng = np.random.default_rng(seed)
    gal = _unit(rng.standard_normal((g, D)))
    scale = np.ones((g, 1), dtype=np.float32)
    scale[g // 2:] = rng.uniform(4.0, 40.0, (g - g // 2, 1)).astype(np.float32)
    slots = rng.permutation(g)
    used = 0
    probes = np.zeros((b, D), dtype=np.float32)
    for p in range(b):
        base = _unit(rng.standard_normal((1, D)))[0]
        kind = p % 4
        n_near = int(rng.integers(3, 24))
        rows = slots[used:used + n_near]
        used += n_near
        for j, r in enumerate(rows):
            v = base.copy()
            if kind == 0:                                   # a handful of components moved by one ulp
                k = rng.integers(0, D, 8)
                v[k] = np.nextafter(v[k], np.float32(np.inf) * np.sign(rng.standard_normal(8)).astype(np.float32))
            elif kind == 1:                                 # 1e-7-scale relative noise on every component
                v = (v * (1.0 + 1e-7 * rng.standard_normal(D))).astype(np.float32)
            elif kind == 2:                                 # 1e-4 perturbation (near-duplicate enrolments)
                v = (v + 1e-4 * rng.standard_normal(D) / np.sqrt(D)).astype(np.float32)
            else:                                           # exact duplicates of two slightly different rows
                if j % 2:
                    v[:4] = np.nextafter(v[:4], np.float32(2.0))
            gal[r] = v * (scale[r] if kind != 3 else np.float32(1.0))
            if kind == 3:
                scale[r] = 1.0
        noise = (0.0, 1e-7, 1e-3, 0.05)[int(rng.integers(0, 4))]
        q = base + noise * rng.standard_normal(D).astype(np.float32) / np.float32(np.sqrt(D))
        probes[p] = (q * np.float32(rng.uniform(0.5, 3.0))).astype(np.float32)
    for r in slots[used:]:
        gal[r] = gal[r] * scale[r]
    return probes.astype(np.float32), gal.astype(np.float32)
