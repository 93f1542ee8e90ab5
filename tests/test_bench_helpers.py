"""CPU: the pieces of bench.py that decide what a multi-GPU run is checked against -- the look-up of committed reference-run
hashes (tests/golden/*.json) for the global seed-0 clouds, and the generator those hashes were computed on."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

from oracle import datasets  # noqa: E402


def test_golden_lookup_finds_the_baseline_configurations(large_cases, golden_cases):
    g = bench.golden_sor(50_000_000, 10.0, 32, 1.0)          # BASELINE configs[3]
    assert g and g["mask_sha"] == large_cases["sor_u50m_L10_k32_s1"]["mask_sha"] and g["survivors"] == 42361425
    g = bench.golden_sor(10_000_000, 5.0, 16, 1.0)           # the headline's cloud
    assert g and g["survivors"] == large_cases["sor_u10m_L5_k16_s1"]["survivors"]
    g = bench.golden_sor(1_000_000, 10.0, 16, 1.0)           # SURVEY's KAT SOR-1M (BASELINE configs[1])
    assert g and g["mask_sha"] == "bb601219805e74a7" == golden_cases["sor"]["sor_u1m_k16_s1"]["mask_sha"] and g["survivors"] == 848169
    assert bench.golden_sor(1_000_000, 10.0, 16, 2.0)["mask_sha"] == "9e7e05a43d98a513"
    assert bench.golden_sor(1_000_000, 10.0, 17, 1.0) is None and bench.golden_sor(999_999, 10.0, 16, 1.0) is None
    c = bench.golden_chain(10_000_000, 5.0, 0.5, 16, 1.0)    # BASELINE configs[2]
    assert c and c["final_survivors"] == 8110488 and c["density_kept"] == 9602310
    assert bench.golden_chain(10_000_000, 5.0, 0.4, 16, 1.0) is None


def test_bench_generator_is_the_one_the_goldens_were_made_with(golden_cases):
    x = bench.synth_uniform(100_000, 10.0, 0)
    np.testing.assert_array_equal(x, datasets.uniform(100_000, 10.0, 0))
    assert hashlib.sha256(x.tobytes()).hexdigest()[:16] == golden_cases["sor"]["sor_u100k_k8_s1"]["xyz_sha"]
    # an index shard of the global cloud is a slice of it, whatever the number of ranks
    for world in (2, 3, 8):
        parts = [x[r * len(x) // world:(r + 1) * len(x) // world] for r in range(world)]
        np.testing.assert_array_equal(np.concatenate(parts), x)
