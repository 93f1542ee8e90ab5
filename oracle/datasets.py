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


# ---- K-Means / SOG inputs (oracle/make_golden_kmeans.py, tests/test_kmeans_*.py) ---------------------
def km_data(spec: dict) -> np.ndarray:
    """Synthetic K-Means inputs (SURVEY.md 8(d) config 5 distributions, scaled down)."""
    rng = np.random.default_rng(spec["seed"])
    kind = spec["kind"]
    if kind == "normal":      # f_rest ~ N(0, 0.1^2), scales ~ N(-4, 1), f_dc ~ N(0, 1)
        return (rng.standard_normal((spec["n"], spec["d"])) * spec.get("sigma", 1.0) + spec.get("mu", 0.0)).astype(np.float32)
    if kind == "repeated":    # every distinct row appears `rep` times: coinciding initial centroids -> empty clusters
        base = (rng.standard_normal((spec["n"] // spec["rep"], spec["d"])) * 0.1).astype(np.float32)
        return np.repeat(base, spec["rep"], axis=0)[rng.permutation(spec["n"] // spec["rep"] * spec["rep"])]
    if kind == "lattice2d":   # integer lattice: exact distance ties between centroids (lowest index must win)
        g = np.arange(spec["m"], dtype=np.float32)
        pts = np.stack(np.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
        return np.tile(pts, (spec["rep"], 1))[rng.permutation(len(pts) * spec["rep"])].copy()
    raise ValueError(kind)


def splat_dtype(sh_degree: int = 3):
    """The reference's standard table layout (structures.py:23-59, no prefix, no RGB): 62 x f4 at degree 3."""
    n_rest = 3 * ((sh_degree + 1) ** 2 - 1)
    names = (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % i for i in range(n_rest)] +
             ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    return [(nm, "f4") for nm in names]


def sog_scene(n: int, seed: int, sh_degree: int = 3) -> np.ndarray:
    """A synthetic splat table (structured array) with every column the SOG writer reads."""
    rng = np.random.default_rng(seed)
    a = np.zeros(n, dtype=splat_dtype(sh_degree))
    xyz = (rng.standard_normal((n, 3)) * 3.0).astype(np.float32)
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    for i in range(3):
        a["f_dc_%d" % i] = rng.standard_normal(n).astype(np.float32)
        a["scale_%d" % i] = (rng.standard_normal(n) - 4.0).astype(np.float32)
    for i in range(3 * ((sh_degree + 1) ** 2 - 1)):
        a["f_rest_%d" % i] = (rng.standard_normal(n) * 0.1).astype(np.float32)
    a["opacity"] = (rng.standard_normal(n) * 2.0).astype(np.float32)
    q = rng.standard_normal((n, 4)).astype(np.float32)
    for i in range(4):
        a["rot_%d" % i] = q[:, i]
    return a
