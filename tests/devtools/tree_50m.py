"""GPU box: 50M splats (a 10^3 scene + 0.5 % far floaters; six blobs + flyers) through adaptive mode with the tree path and with the
level-by-level grid refinement -- bit-identical mean distances, and the times.  usage: python tests/devtools/tree_50m.py"""
import importlib, os, sys, time, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib
for kind in ("floaters", "clustered"):
    n = 50_000_000
    xyz = bench.synth_scene_with_floaters(n, 1) if kind == "floaters" else bench.synth_clustered(n, 1)
    ctx = L.Context(0)
    ctx.set_param("adaptive", 1)
    rows = ctx.alloc(xyz.nbytes).upload(xyz)
    out = ctx.alloc(4 * n)
    res = {}
    for tree in (1, 0):
        ctx.set_param("tree", tree)
        ctx.sor_knn(rows.ptr, rows.ptr + 4, rows.ptr + 8, 3, n, 0, n, 16, out.ptr)
        ctx.synchronize()
        t0 = time.perf_counter()
        info = ctx.sor_knn(rows.ptr, rows.ptr + 4, rows.ptr + 8, 3, n, 0, n, 16, out.ptr, want_info=True)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        got = out.download(np.float32, n)
        res[tree] = hashlib.sha256(got.tobytes()).hexdigest()[:16]
        print("%s 50M tree=%d: %.1f ms  algo %d  leaves/bricks %d  fallback %d  finite %s  sha %s" % (
            kind, tree, dt, info["algo"], info["n_bricks"], info["n_fallback"], bool(np.isfinite(got).all()), res[tree]), flush=True)
    print(kind, "identical:", res[1] == res[0], flush=True)
    rows.free(); out.free(); ctx.close()
