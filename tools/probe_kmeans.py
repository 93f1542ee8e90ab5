"""GPU box: BASELINE.json configs[4] (bench.py:run_kmeans) alone, for `rocprofv3 --kernel-trace --stats` (the per-kernel
averages of a Lloyd iteration on ONE lane; concurrent lanes only change how the kernels overlap) and for A/B runs of library
knobs:   python tools/probe_kmeans.py [lanes] [name=value ...]      (PROBE_LEVEL=5: another --compression_level, i.e. K per chunk)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # 0 = the batched call (round 5), N > 0 = N concurrent lanes
params = [(kv.split("=")[0], float(kv.split("=")[1])) for kv in sys.argv[2:]]
r = bench.run_kmeans(L, L.Context(0), gsx, 10_000_000, 2, 1, cpu=False, lanes=lanes, params=params, level=int(os.environ.get("PROBE_LEVEL", "2")))
print(lanes, params, r["ms_per_step"], r["kernel_ms_per_step"], r["workload"][-110:])
