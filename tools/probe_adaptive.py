"""GPU box: one adaptive SOR step on the clouds a uniform grid is bad at, with the library's level trace (GSX_TRACE_LEVELS=1)
and wall times; run under `rocprofv3 --kernel-trace --stats` for the per-kernel view.
    python tools/probe_adaptive.py clustered 1000000 | floaters 10000000"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (its generators)


def main():
    kind, n = sys.argv[1], int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    gsx = importlib.import_module("3dgsconverter_amd")
    L = gsx._lib
    ctx = L.Context(0)
    xyz = bench.synth_clustered(n, 0) if kind == "clustered" else bench.synth_scene_with_floaters(n, 0)
    ctx.set_param("adaptive", 1)
    b = bench.SorBench(L, ctx, xyz, 16, 1.0)
    b.step()
    ctx.synchronize()
    os.environ["GSX_TRACE_LEVELS"] = "1"
    for r in range(reps):
        if r > 0:
            os.environ.pop("GSX_TRACE_LEVELS", None)
        t0 = time.perf_counter()
        b.step()
        ctx.synchronize()
        print("%s %d: step %.3f ms" % (kind, n, (time.perf_counter() - t0) * 1e3), flush=True)
    mask, stats = b.results()
    print("survivors", int(mask.sum()), "threshold", float(stats[2]), b.info())


if __name__ == "__main__":
    main()
