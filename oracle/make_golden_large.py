"""Hash-only golden vectors at BASELINE.json's full sizes, produced by RUNNING THE REFERENCE ITSELF.

    python -m oracle.make_golden_large [name ...]       # needs /root/reference; minutes of CPU per case

TEST INFRASTRUCTURE ONLY.  The arrays are too large to commit (a 10M-splat mask is 1.25 MB), so only
their SHA-256 prefixes, the statistics (hex of the IEEE bytes) and the survivor count go to
tests/golden/large_cases.json.  Same capture as oracle/make_golden.py: the locals of the reference's
``DataProcessor.remove_flyers`` CPU branch (data_processor.py:156-180) / its ``apply_density_filter``.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import datasets, refload  # noqa: E402
from oracle.make_golden import sha, f32hex  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "large_cases.json")

SOR_LARGE = [
    # the bench workload (BASELINE.json target config): 10M uniform splats, L=5, seed 0, k=16, sigma 1
    ("sor_u10m_L5_k16_s1", {"kind": "uniform", "n": 10_000_000, "extent": 5.0, "seed": 0}, 16, 1.0),
    # BASELINE.json configs[3]: 50M uniform splats, L=10, seed 0, k=32, sigma 1 -- the whole cloud
    # through the reference's own remove_flyers CPU branch (cKDTree over 50M points: minutes, ~15 GB)
    ("sor_u50m_L10_k32_s1", {"kind": "uniform", "n": 50_000_000, "extent": 10.0, "seed": 0}, 32, 1.0),
    # what SOR exists for, at the bench's size (bench.py configs.floaters_10m; the Morton-tree path): a 10^3 scene + 0.5 %
    # floaters in a 1000^3 box, k=16 (round 4: the tree path's full mask against a reference run, not against the other GPU path)
    ("sor_floaters10m_k16_s1", {"kind": "scene_with_floaters", "n": 10_000_000, "seed": 0}, 16, 1.0),
]
# BASELINE.json configs[2]: density sensitivity 0.5 on the same cloud, then SOR k=16 on the survivors
CHAIN_LARGE = [
    ("chain_u10m_L5_dens0p5_sor_k16_s1", {"kind": "uniform", "n": 10_000_000, "extent": 5.0, "seed": 0}, 0.5, 16, 1.0),
]


def main(argv):
    want = set(argv)
    cases = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            cases = json.load(f)
    for name, spec, k, sigma in SOR_LARGE:
        if want and name not in want:
            continue
        xyz = datasets.make(spec)
        t0 = time.time()
        cap = refload.reference_sor(xyz, k, sigma)
        cases[name] = {
            "dataset": spec, "k": k, "sigma": sigma, "n": int(len(xyz)), "xyz_sha": sha(xyz.tobytes()),
            "mean_hex": f32hex(cap["mean"]), "std_hex": f32hex(cap["std"]), "threshold_hex": f32hex(cap["threshold"]),
            "survivors": int(cap["mask"].sum()), "mean_dists_sha": sha(cap["mean_dists"].tobytes()),
            "mask_sha": sha(np.packbits(cap["mask"]).tobytes()),
            "reference_seconds_here": round(time.time() - t0, 1), "reference_cores_here": os.cpu_count()}
        print(name, cases[name])
    for name, spec, sens, k, sigma in CHAIN_LARGE:
        if want and name not in want:
            continue
        xyz = datasets.make(spec)
        t0 = time.time()
        dens = refload.reference_density(xyz, sensitivity=sens)
        t1 = time.time()
        kept = xyz[dens["mask"]]
        cap = refload.reference_sor(kept, k, sigma)
        final = np.zeros(len(xyz), dtype=bool)
        final[np.nonzero(dens["mask"])[0][cap["mask"]]] = True
        cases[name] = {
            "dataset": spec, "sensitivity": sens, "k": k, "sigma": sigma, "n": int(len(xyz)),
            "xyz_sha": sha(xyz.tobytes()), "density_kept": int(dens["mask"].sum()),
            "density_mask_sha": sha(np.packbits(dens["mask"]).tobytes()), "density_messages": dens["messages"],
            "sor_threshold_hex": f32hex(cap["threshold"]), "sor_mean_dists_sha": sha(cap["mean_dists"].tobytes()),
            "final_survivors": int(final.sum()), "final_mask_sha": sha(np.packbits(final).tobytes()),
            "reference_seconds_here": {"density": round(t1 - t0, 1), "sor": round(time.time() - t1, 1)},
            "reference_cores_here": os.cpu_count()}
        print(name, cases[name])
    import numpy, scipy
    cases["_meta"] = {"numpy": numpy.__version__, "scipy": scipy.__version__,
                      "generator": "oracle/make_golden_large.py run against /root/reference (v0.8, CPU fallbacks)"}
    with open(OUT, "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main(sys.argv[1:])
