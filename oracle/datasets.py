"""Deterministic synthetic clouds shared by oracle/make_golden.py, tests/ and bench.py.

TEST INFRASTRUCTURE ONLY (bench.py re-implements ``uniform`` for its timed input so
that the product path does not import oracle/).
"""
from __future__ import annotations

import numpy as np


def uniform(n: int, extent: float = 10.0, seed: int = 0) -> np.ndarray:
    """rng.random((N,3), f32) * f32(L)."""
    rng = np.random.default_rng(seed)
    return rng.random((n, 3), dtype=np.float32) * np.float32(extent)


def clustered(n: int, seed: int = 0) -> np.ndarray:
    """Gaussian blobs of very different density + 2 % far 'flyers' (what SOR is for)."""
    rng = np.random.default_rng(seed)
    n_out = max(1, n // 50)
    n_in = n - n_out
    centers = rng.random((6, 3)) * 20.0 - 10.0
    sigmas = np.array([0.05, 0.2, 0.5, 1.0, 1.5, 0.1])
    which = rng.integers(0, 6, n_in)
    pts = centers[which] + rng.standard_normal((n_in, 3)) * sigmas[which, None]
    out = rng.random((n_out, 3)) * 80.0 - 40.0
    xyz = np.concatenate([pts, out]).astype(np.float32)
    return xyz[rng.permutation(n)]


def duplicates(n: int, seed: int = 0) -> np.ndarray:
    """Coarsely rounded coordinates: many exact duplicates and exact distance ties."""
    rng = np.random.default_rng(seed)
    xyz = np.round(rng.random((n, 3)) * 12.0) * 0.25
    return xyz.astype(np.float32)


def scene_with_floaters(n: int, seed: int = 0, far: float = 500.0) -> np.ndarray:
    """What a captured 3DGS scene looks like to a uniform grid: 99.5 % of the splats in a 10^3 box, 0.5 % 'floaters'
    spread over (2*far)^3 -- the bounding box is 10^6 times the volume that matters."""
    rng = np.random.default_rng(seed)
    n_out = max(1, n // 200)
    pts = rng.random((n - n_out, 3)) * 10.0
    out = rng.random((n_out, 3)) * (2.0 * far) - far
    xyz = np.concatenate([pts, out]).astype(np.float32)
    return xyz[rng.permutation(n)]


def lattice(m: int) -> np.ndarray:
    """m^3 integer lattice: every neighbour shell is an exact tie."""
    g = np.arange(m, dtype=np.float32)
    return np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).copy()


def two_blobs(n: int, seed: int = 0, ratio: float = 0.25, gap: float = 6.0) -> np.ndarray:
    """Two separated uniform boxes (big + small) + sparse background: exercises keep_multicluster."""
    rng = np.random.default_rng(seed)
    n_small = int(n * ratio)
    n_bg = n // 20
    n_big = n - n_small - n_bg
    big = rng.random((n_big, 3)) * np.array([4.0, 4.0, 4.0])
    small = rng.random((n_small, 3)) * np.array([2.0, 2.0, 2.0]) + np.array([4.0 + gap, 0.0, 0.0])
    bg = rng.random((n_bg, 3)) * 40.0 - 15.0
    xyz = np.concatenate([big, small, bg]).astype(np.float32)
    return xyz[rng.permutation(len(xyz))]


def centered(n: int, extent: float, seed: int = 0) -> np.ndarray:
    """Uniform in [-extent/2, extent/2): negative voxel keys."""
    return (uniform(n, extent, seed) - np.float32(extent / 2)).astype(np.float32)


def make(spec: dict) -> np.ndarray:
    kind = spec["kind"]
    if kind == "uniform":
        return uniform(spec["n"], spec.get("extent", 10.0), spec.get("seed", 0))
    if kind == "clustered":
        return clustered(spec["n"], spec.get("seed", 0))
    if kind == "duplicates":
        return duplicates(spec["n"], spec.get("seed", 0))
    if kind == "lattice":
        return lattice(spec["m"])
    if kind == "two_blobs":
        return two_blobs(spec["n"], spec.get("seed", 0), spec.get("ratio", 0.25), spec.get("gap", 6.0))
    if kind == "scene_with_floaters":
        return scene_with_floaters(spec["n"], spec.get("seed", 0), spec.get("far", 500.0))
    if kind == "centered":
        return centered(spec["n"], spec.get("extent", 10.0), spec.get("seed", 0))
    raise ValueError(kind)
