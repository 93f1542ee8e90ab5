"""not gpu: the scalar-codebook solver that replaces the reference's scikit-learn calls (formats/sog.py:561,
processing/gpu_ops.py:48-52 for D = 1) -- its numpy restatement (oracle/kmeans.py:kmeans1d_sorted, the checker of
csrc/kmeans1d.hip) against scikit-learn itself, the reference's actual arithmetic for these calls.  The reference is
unseeded, so the bar is quality: inertia no worse than MiniBatchKMeans' on the same values."""
import numpy as np
import pytest

from oracle import kmeans as okm

CASES = {
    "scales_50k": lambda r: (r.standard_normal(50000) - 4.0).astype(np.float32),                 # sog.py:396-402 sample
    "dc_50k": lambda r: r.standard_normal(50000).astype(np.float32),                             # sog.py:438-443 sample
    "palette_flat": lambda r: (r.standard_normal(400_000) * 0.1).astype(np.float32),             # sog.py:557-561 (scaled down)
    "bimodal": lambda r: np.concatenate([r.standard_normal(25000) * 0.2 - 3, r.standard_normal(25000) * 2 + 1]).astype(np.float32),
    "outliers": lambda r: np.concatenate([r.standard_normal(49990) * 0.1, r.standard_normal(10) * 50]).astype(np.float32),
    "uniform": lambda r: r.random(50000).astype(np.float32),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_sorted_run_solver_is_at_least_sklearn_quality(name):
    x = CASES[name](np.random.default_rng(7))
    cent, inertia = okm.kmeans1d_sorted(x, 256, 50)
    assert np.all(np.diff(cent) >= 0) and len(cent) == 256
    assert abs(okm.inertia_1d(x, cent) - inertia[0]) <= 1e-6 * inertia[0] + 1e-9   # the prefix-sum inertia is the real one
    assert inertia[0] == min(inertia[1], inertia[2])
    np.random.seed(3)
    sk = min(okm.inertia_1d(x, okm.sklearn_codebook_561(x)) for _ in range(2))
    assert inertia[0] <= 1.0 * sk, (inertia, sk)


def test_fewer_distinct_values_than_centroids():
    x = np.random.default_rng(1).integers(0, 40, 50000).astype(np.float32)
    cent, inertia = okm.kmeans1d_sorted(x, 256, 50)
    assert inertia[0] <= 1e-6 and set(np.unique(x)) <= set(cent.tolist())
    cent, inertia = okm.kmeans1d_sorted(np.full(1000, 2.5, np.float32), 16, 10)
    assert inertia[0] == 0.0 and np.all(cent == np.float32(2.5))


def test_kmeans_pp_restatement_separated_clusters():
    """D^2 sampling must take one row from every cluster when the clusters are tight and far apart"""
    rng = np.random.default_rng(0)
    centers = rng.random((12, 5)) * 1000.0
    which = rng.integers(0, 12, 6000)
    x = (centers[which] + rng.standard_normal((6000, 5)) * 1e-3).astype(np.float32)
    idx = okm.kmeans_pp_restated(x, 12, rng.random(1 + 11 * 4), 4)
    assert sorted(which[idx].tolist()) == list(range(12))
    idx = okm.kmeans_pp_restated(x, 12, rng.random(12), 1)
    assert sorted(which[idx].tolist()) == list(range(12))
