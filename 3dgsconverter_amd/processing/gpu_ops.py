"""Drop-in for gsconverter/processing/gpu_ops.py: same public names
(``HAS_TAICHI``, ``kmeans``, ``filter_sor_gpu``), HIP/gfx950 underneath.

  * ``HAS_TAICHI`` keeps its name because callers read it as "is the accelerated backend
    there" (data_processor.py:143, sog.py:524); it is True iff a gfx950 device is visible.
    ``HAS_HIP`` is the honest alias.
  * ``kmeans``  (reference :27-52, Lloyd kernels :57-96, driver :178-191): exactly
    ``max_iter`` x (assign, update) on the GPU from a random-sample init drawn like the
    reference does (np.random.choice, unseeded unless the caller seeds numpy); ``use_gpu=False``
    selects the quality of the reference's scikit-learn path (k-means++ seeding) -- still on the GPU.
    No sklearn fallback.
  * ``filter_sor_gpu`` (reference :193-263): returns the boolean survivor mask; exact KNN
    (not the 27-cell / K<=50 approximation), statistics with numpy's exact f32 arithmetic.
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from ..utils import status_print

HAS_HIP = _lib.has_hip()
HAS_TAICHI = HAS_HIP  # the capability flag callers test (see module docstring)


def kmeans(data: np.ndarray, k: int, max_iter=10, tolerance=1e-4, use_gpu=True, verbose=False,
           init_centroids=None, init=None):
    """reference :27-52.  Two paths, like the reference, both on the GPU here:

    * ``use_gpu=True`` (default; the reference's Taichi path :178-191): uniformly random rows as initial centroids
      (np.random.choice, :182), exactly ``max_iter`` x (assign, update), labels one step older than the centroids;
    * ``use_gpu=False`` or ``init="k-means++"`` (the reference's scikit-learn path :48-52, MiniBatchKMeans with k-means++
      seeding -- what it runs on a machine without Taichi): k-means++ seeding + ``max_iter`` full-batch Lloyd steps +
      labels of the returned centroids; for D == 1 (the scalar codebooks of formats/sog.py:402,443,561) the sorted-array
      solver of csrc/kmeans1d.hip (deterministic, centroids ascending).  Unseeded in the reference: the contract is
      quality (inertia <= sklearn's), see tests/test_kmeans_gpu.py.
    ``tolerance`` is accepted and unused, as in the reference."""
    N, D = data.shape
    if k >= N:  # reference :30-31 (returns the input dtype, not forced to f32)
        return data.copy(), np.arange(N, dtype=np.int32)
    data32 = np.ascontiguousarray(data, dtype=np.float32)
    quality = (not use_gpu) or (init is not None and str(init).lower() in ("k-means++", "kmeans++", "quality"))
    if init_centroids is None and quality:
        if D == 1 and k <= 1024:
            if verbose:
                status_print(f"[GPU_OPS] scalar K-Means on HIP gfx950 (sorted runs): N={N}, K={k}")
            cent, labels = _lib.kmeans1d(data32.reshape(-1), int(k), iters=max(int(max_iter), 50), want_labels=True)
            return cent.reshape(-1, 1), labels
        if verbose:
            status_print(f"[GPU_OPS] k-means++ seeding + Lloyd on HIP gfx950: N={N}, D={D}, K={k}, iters={max_iter}")
        trials = _lib.kmeans_pp_trials(int(k))
        seeds = _lib.kmeans_pp(data32, int(k), np.random.random_sample(1 + (int(k) - 1) * trials), trials)
        cent, _ = _lib.kmeans_lloyd(data32, seeds, int(max_iter))
        _, labels = _lib.kmeans_lloyd(data32, cent, 1)     # labels OF the returned centroids (compute_labels=True, :50)
        return cent, labels
    if init_centroids is None:
        init_centroids = data32[np.random.choice(N, k, replace=False)]  # reference :182
    if verbose:
        status_print(f"[GPU_OPS] Lloyd K-Means on HIP gfx950: N={N}, D={D}, K={k}, iters={max_iter}")
    return _lib.kmeans_lloyd(data32, np.ascontiguousarray(init_centroids, dtype=np.float32), int(max_iter))


def filter_sor_gpu(data_np: np.ndarray, k: int = 25, threshold_factor: float = 1.0, verbose=False):
    N, D = data_np.shape
    if D != 3:
        raise ValueError("Requires 3D data")
    if not 1 <= int(k) <= 2047:
        raise ValueError(f"SOR: k={k} is outside the supported range 1..2047 of the MI355X path")
    res = _lib.sor_filter(np.ascontiguousarray(data_np, dtype=np.float32), int(k), float(threshold_factor),
                          want_mean=False)
    if verbose:
        status_print(f"[SOR] HIP exact KNN: mean={res['mean']:.6g} std={res['std']:.6g} thr={res['threshold']:.6g}")
    return res["mask"]


def quantize_to_codebook(vals: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    """GPU version of the closure in formats/sog.py:408-419 (nearest entry of a sorted codebook)."""
    return _lib.quantize_sorted_codebook(np.ascontiguousarray(vals, dtype=np.float32),
                                         np.ascontiguousarray(codebook, dtype=np.float32))
