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


def match_degenerate_cases(g=4608, seed=77):
    """Inputs the reference's distance has no guard for (evaluation/utility.py:58-62 divides by the norm
    product and feeds arccos unclamped), so its answer is whatever IEEE arithmetic and np.argmin's
    first-NaN rule give: zero-norm / tiny / huge / non-finite gallery rows at several positions, rows exactly
    anti-parallel to a probe (similarity may round below -1), zero / tiny / huge / non-finite probes, an
    all-zero gallery.  Returns [(name, probes [b, D], gallery [g, D])]; every gallery has >= 4096 rows."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    cases = []

    def base():
        gal = _unit(rng.standard_normal((g, D)))
        p, pick = probes_from(gal, 24, seed=int(rng.integers(1 << 30)))
        p = (p * rng.uniform(0.5, 3.0, (24, 1))).astype(f32)
        return gal, p

    # 1. zero-norm rows at several positions (the first one after a few hundred ordinary rows)
    gal, p = base()
    gal[[417, 1500, 1501, 4000, g - 1]] = 0
    cases.append(('zero_rows', p, gal))

    # 2. zero-norm row at index 0 and in the last tile
    gal, p = base()
    gal[[0, g - 3]] = 0
    cases.append(('zero_row_first', p, gal))

    # 3. non-finite elements: +inf, -inf, NaN, a row with both; no zero rows
    gal, p = base()
    gal[900, 17] = np.inf
    gal[3100, 400] = -np.inf
    gal[2200, 5] = np.nan
    gal[2201, [1, 2]] = [np.nan, np.inf]
    gal[4500, 0] = np.inf
    gal[4500, 1] = -np.inf
    cases.append(('nonfinite_rows', p, gal))

    # 4. NaN row first, inf rows later (metric 0: the NaN row wins; metric 1: the lowest of all of them)
    gal, p = base()
    gal[3000, 100] = np.nan
    gal[1200, 7] = np.inf
    cases.append(('nan_after_inf', p, gal))

    # 5. tiny and huge rows that are ordinary for the reference: scaled copies of the probes' true neighbours
    gal, p = base()
    for k, (row, sc) in enumerate(((50, 1e-17), (700, 1e-18), (1300, 3e-16), (2600, 1e16), (3900, 4e17), (4400, 1e-16))):
        gal[row] = (_unit(p[k][None])[0].astype(np.float64) * sc).astype(f32)
    gal[2000] = (gal[2000].astype(np.float64) * 1e-25).astype(f32)      # squares underflow to 0: norm 0 -> NaN
    gal[2500] = (gal[2500].astype(np.float64) * 1e19).astype(f32)       # squares sum to +inf: similarity 0
    cases.append(('tiny_huge_rows', p, gal))
    gal2 = gal.copy()
    gal2[2000] = gal2[2001]
    cases.append(('tiny_huge_rows_no_nan', p, gal2))

    # 6. anti-parallel rows: g = -c q, so the similarity is -1 up to rounding and lands below -1 for some
    gal, p = base()
    pa = np.concatenate([p, _unit(rng.standard_normal((40, D))) * rng.uniform(0.3, 5.0, (40, 1)).astype(f32)]).astype(f32)
    rows = rng.permutation(g)[:pa.shape[0]]
    for k, r in enumerate(rows):
        gal[r] = (-pa[k] * f32(rng.uniform(0.25, 4.0))).astype(f32)
    cases.append(('antiparallel', pa, gal))

    # 7. degenerate probes against an ordinary gallery (+ one zero row late, one inf row)
    gal, p = base()
    bad = p[:12].copy()
    bad[0] = 0
    bad[1, 3] = np.nan
    bad[2, 9] = np.inf
    bad[3, 9] = -np.inf
    bad[4] = (bad[4].astype(np.float64) * 1e-25).astype(f32)            # |q|^2 underflows to 0
    bad[5] = (bad[5].astype(np.float64) * 1e-17).astype(f32)            # tiny but valid
    bad[6] = (bad[6].astype(np.float64) * 1e17).astype(f32)             # huge but valid
    bad[7] = (bad[7].astype(np.float64) * 1e20).astype(f32)             # |q|^2 overflows to +inf
    bad[8, :] = np.nan
    bad[9] = 0
    bad[9, 0] = 1e-30
    bad[10, 100] = np.inf
    bad[10, 101] = np.nan
    probes = np.concatenate([bad, p[12:]]).astype(f32)
    cases.append(('odd_probes', probes, gal))
    gal2 = gal.copy()
    gal2[3333] = 0
    gal2[150, 9] = np.inf                                               # same position / sign as probe 2's inf
    cases.append(('odd_probes_odd_rows', probes, gal2))

    # 8. a gallery of nothing but zero rows (an unfilled shard)
    cases.append(('all_zero_gallery', probes, np.zeros((4096, D), dtype=f32)))
    return cases
