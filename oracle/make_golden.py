"""Generate tests/golden/ by RUNNING THE REFERENCE ITSELF in the build container.

    python -m oracle.make_golden            # needs /root/reference

TEST INFRASTRUCTURE ONLY.  SOR vectors come from the reference's
``DataProcessor.remove_flyers`` CPU branch (locals captured from its frame, see
oracle/refload.py), density vectors from its ``apply_density_filter``.  The K-Means
entries written HERE are regression pins of the oracle's own restatement; the
reference-generated K-Means / SOG fixtures are made by oracle/make_golden_kmeans.py,
the full-size (10M-splat) hashes by oracle/make_golden_large.py.
Scalars are stored as hex of their IEEE bytes so comparisons are exact.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import datasets, refload, kmeans as okm  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()[:16]


def f32hex(x) -> str:
    return np.float32(x).tobytes().hex()


SOR_CASES = [
    # name, dataset spec, k, sigma, intensity, store full mean_dists?
    ("sor_u100k_k8_s1", {"kind": "uniform", "n": 100000, "extent": 10.0, "seed": 0}, 8, 1.0, None, False),
    ("sor_u100k_k8_s10p5", {"kind": "uniform", "n": 100000, "extent": 10.0, "seed": 0}, 8, 10.5, None, False),
    ("sor_u1m_k16_s1", {"kind": "uniform", "n": 1000000, "extent": 10.0, "seed": 0}, 16, 1.0, None, False),
    ("sor_u1m_k16_s2", {"kind": "uniform", "n": 1000000, "extent": 10.0, "seed": 0}, 16, 2.0, None, False),
    ("sor_u20k_k16_s1", {"kind": "uniform", "n": 20000, "extent": 10.0, "seed": 1}, 16, 1.0, None, True),
    ("sor_u50k_k32_s1", {"kind": "uniform", "n": 50000, "extent": 10.0, "seed": 2}, 32, 1.0, None, False),
    ("sor_clustered30k_k16_s2", {"kind": "clustered", "n": 30000, "seed": 3}, 16, 2.0, None, True),
    ("sor_dups8k_k8_s1", {"kind": "duplicates", "n": 8000, "seed": 4}, 8, 1.0, None, True),
    ("sor_lattice17_k6_s1", {"kind": "lattice", "m": 17}, 6, 1.0, None, True),
    ("sor_lattice17_k26_s0p5", {"kind": "lattice", "m": 17}, 26, 0.5, None, True),
    ("sor_tiny10_k16", {"kind": "uniform", "n": 10, "extent": 1.0, "seed": 5}, 16, 1.0, None, True),
    ("sor_tiny17_k16", {"kind": "uniform", "n": 17, "extent": 1.0, "seed": 5}, 16, 1.0, None, True),
    ("sor_u20k_int1", {"kind": "uniform", "n": 20000, "extent": 10.0, "seed": 6}, 25, 10.5, 1, False),
    ("sor_u20k_int5", {"kind": "uniform", "n": 20000, "extent": 10.0, "seed": 6}, 25, 10.5, 5, False),
    ("sor_u20k_int10", {"kind": "uniform", "n": 20000, "extent": 10.0, "seed": 6}, 25, 10.5, 10, False),
    ("sor_centered40k_k16_s1", {"kind": "centered", "n": 40000, "extent": 200.0, "seed": 7}, 16, 1.0, None, False),
    # a scene whose bounding box is inflated 10^6 x by far floaters (what SOR is for; adaptive grid on the GPU)
    ("sor_scene150k_k16_s1", {"kind": "scene_with_floaters", "n": 150000, "seed": 3}, 16, 1.0, None, False),
]

DENSITY_CASES = [
    # name, dataset, kwargs
    ("dens_u1m_L5_s0p5", {"kind": "uniform", "n": 1000000, "extent": 5.0, "seed": 0}, {"sensitivity": 0.5}),
    ("dens_u1m_L5_s0p5_multi", {"kind": "uniform", "n": 1000000, "extent": 5.0, "seed": 0},
     {"sensitivity": 0.5, "keep_multicluster": True}),
    ("dens_u1m_L8_s0p1", {"kind": "uniform", "n": 1000000, "extent": 8.0, "seed": 0}, {"sensitivity": 0.1}),
    ("dens_u100k_L1_s0p5_all", {"kind": "uniform", "n": 100000, "extent": 1.0, "seed": 0}, {"sensitivity": 0.5}),
    ("dens_u200k_L10_s0p5_none", {"kind": "uniform", "n": 200000, "extent": 10.0, "seed": 0}, {"sensitivity": 0.5}),
    ("dens_blobs_single", {"kind": "two_blobs", "n": 120000, "seed": 1},
     {"voxel_size": 0.5, "threshold_percentage": 0.05}),
    ("dens_blobs_multi", {"kind": "two_blobs", "n": 120000, "seed": 1},
     {"voxel_size": 0.5, "threshold_percentage": 0.05, "keep_multicluster": True}),
    ("dens_blobs_tiny_multi", {"kind": "two_blobs", "n": 120000, "seed": 1, "ratio": 0.02},
     {"voxel_size": 0.5, "threshold_percentage": 0.05, "keep_multicluster": True}),
    ("dens_centered_s0p9_none", {"kind": "centered", "n": 300000, "extent": 3.0, "seed": 2}, {"sensitivity": 0.9}),
    ("dens_centered_v0p7", {"kind": "centered", "n": 300000, "extent": 3.0, "seed": 2},
     {"voxel_size": 0.7, "threshold_percentage": 1.0}),
    ("dens_clustered_default", {"kind": "clustered", "n": 200000, "seed": 3}, {}),
    ("dens_clustered_s0", {"kind": "clustered", "n": 200000, "seed": 3}, {"sensitivity": 0.0}),
    ("dens_clustered_s1_multi", {"kind": "clustered", "n": 200000, "seed": 3},
     {"sensitivity": 1.0, "keep_multicluster": True}),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {"sor": {}, "density": {}, "kmeans": {}}
    arrays = {}

    for name, spec, k, sigma, intensity, full in SOR_CASES:
        xyz = datasets.make(spec)
        cap = refload.reference_sor(xyz, k, sigma, intensity=intensity)
        md = cap["mean_dists"]
        entry = {
            "dataset": spec, "k_arg": k, "sigma_arg": sigma, "intensity": intensity,
            "k_used": int(cap["k"]), "sigma_used": float(cap["threshold_factor"]),
            "xyz_sha": sha(xyz.tobytes()), "n": int(len(xyz)),
            "mean_hex": f32hex(cap["mean"]), "std_hex": f32hex(cap["std"]), "threshold_hex": f32hex(cap["threshold"]),
            "threshold_type": type(cap["threshold"]).__name__,
            "survivors": int(cap["mask"].sum()),
            "mean_dists_sha": sha(md.tobytes()),
            "mask_sha": sha(np.packbits(cap["mask"]).tobytes()),
        }
        arrays[name + "__mask"] = np.packbits(cap["mask"])
        if full:
            arrays[name + "__mean_dists"] = md
        else:  # a strided sample so a failure can be localised without the full array
            arrays[name + "__mean_dists_sample"] = md[:: max(1, len(md) // 2048)][:2048]
        cases["sor"][name] = entry
        print(name, entry["survivors"], entry["threshold_hex"], entry["mask_sha"])

    for name, spec, kw in DENSITY_CASES:
        xyz = datasets.make(spec)
        ref = refload.reference_density(xyz, **kw)
        entry = {"dataset": spec, "kwargs": kw, "xyz_sha": sha(xyz.tobytes()), "n": int(len(xyz)),
                 "kept": int(ref["mask"].sum()), "mask_sha": sha(np.packbits(ref["mask"]).tobytes()),
                 "messages": ref["messages"]}
        arrays[name + "__mask"] = np.packbits(ref["mask"])
        cases["density"][name] = entry
        print(name, entry["kept"], entry["mask_sha"], ref["messages"])

    # K-Means: restated call-site contract (sog.py closures) + regression pins of the oracle Lloyd
    rng = np.random.default_rng(11)
    cb = np.sort(rng.standard_normal(256).astype(np.float32))
    vals = np.concatenate([rng.standard_normal(5000).astype(np.float32) * 1.5,
                           cb[::7], (cb[:-1] + cb[1:]) / np.float32(2)]).astype(np.float32)
    arrays["kmeans_quant__cb"] = cb
    arrays["kmeans_quant__vals"] = vals
    arrays["kmeans_quant__idx"] = okm.quantize_to_codebook(vals, cb)
    cases["kmeans"]["quantize"] = {"pinned": "restated from formats/sog.py:408-419 (closure, not importable)"}
    cases["kmeans"]["sog_sh_plan"] = {
        str((n, lvl)): okm.sog_sh_plan(n, lvl)
        for n in (2000, 50000, 1000000, 10000000) for lvl in (0, 2, 5, 9)}
    data = (rng.standard_normal((4000, 9)) * 0.1).astype(np.float32)
    init = data[rng.choice(4000, 64, replace=False)]
    cent, lab, cnt = okm.lloyd(data, init, 10)
    arrays["kmeans_lloyd__data"] = data
    arrays["kmeans_lloyd__init"] = init
    arrays["kmeans_lloyd__cent"] = cent
    arrays["kmeans_lloyd__labels"] = lab
    cases["kmeans"]["lloyd_4000x9_k64_it10"] = {
        "pinned": "oracle restatement (f64 sums) with injected init; reference-generated fixtures: kmeans_ref.json",
        "inertia": okm.inertia(data, cent, lab)}

    import numpy, scipy
    cases["_meta"] = {"numpy": numpy.__version__, "scipy": scipy.__version__,
                      "generator": "oracle/make_golden.py run against /root/reference (v0.8, CPU fallbacks)"}
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "arrays.npz"), **arrays)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
