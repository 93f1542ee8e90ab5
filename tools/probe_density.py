"""GPU box: the density stage of BASELINE configs[2] alone (10M uniform splats, L=5, sensitivity 0.5 -> voxel 1.1, 0.55 %), on the
device chain, for `rocprofv3 --kernel-trace --stats` and A/B runs:   python tools/probe_density.py [n] [L] [sensitivity] [cloud]
(cloud: uniform (default) | floaters | blobs -- bench.py's generators; L is ignored for the last two)"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ext = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
sens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
voxel, thr = max(0.1, 2.0 - 1.8 * sens), 0.1 + 0.9 * sens
cloud = sys.argv[4] if len(sys.argv) > 4 else "uniform"
xyz = bench.synth_uniform(n, ext, 0) if cloud == "uniform" else (bench.synth_scene_with_floaters(n, 0) if cloud == "floaters" else bench.synth_clustered(n, 0))
ch = L.DeviceChain(xyz, keep_pristine=True)
minpts = int(n * (thr / 100.0))
res = ch.density_filter(voxel, minpts, False)
print("status", res["status"], "left", res["left"], "unique", res["n_unique"], "clusters", res["kept_clusters"], "largest", res["largest"])
ts = []
for _ in range(10):
    ch.restart()
    ch._box = ch._box if ch._box is not None else None
    ch.ctx.synchronize()
    t0 = time.perf_counter()
    ch.density_filter(voxel, minpts, False)
    ch.ctx.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("density_filter + compaction, host clock ms:", ["%.3f" % t for t in ts])
ch.close()
