"""Randomised parity sweep on the GPU box: gsx_sor_filter (host API, adaptive grid, MFMA filter) and the device
API with both filters against the cKDTree restatement, over random sizes / k / cloud shapes.  Not a test
(tests/ holds a fixed subset); prints one line per case and a summary.  usage: python tests/devtools/fuzz_parity.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib
from oracle import datasets, sor as osor


def make(rng):
    kind = rng.choice(["uniform", "uniform", "clustered", "scene", "duplicates", "plane", "line", "scaled", "tinyblob"])
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 5000), rng.integers(5000, 80000), rng.integers(80000, 300000)]))
    seed = int(rng.integers(0, 1 << 30))
    r2 = np.random.default_rng(seed)
    if kind == "uniform":
        xyz = datasets.uniform(n, float(rng.choice([1e-3, 1.0, 10.0, 1e4])), seed)
    elif kind == "clustered":
        xyz = datasets.clustered(max(n, 60), seed)
    elif kind == "scene":
        xyz = datasets.scene_with_floaters(max(n, 400), seed, far=float(rng.choice([50.0, 500.0, 5000.0])))
    elif kind == "duplicates":
        xyz = datasets.duplicates(max(n, 40), seed)
    elif kind == "plane":
        xyz = (r2.random((n, 3)) * np.array([10.0, 10.0, 0.0])).astype(np.float32)
    elif kind == "line":
        xyz = (r2.random((n, 3)) * np.array([10.0, 0.0, 0.0]) + np.array([0.0, 3.0, -2.0])).astype(np.float32)
    elif kind == "scaled":
        xyz = (r2.random((n, 3)) * np.array([1000.0, 1.0, 0.01]) + 12345.0).astype(np.float32)
    else:  # a tight blob inside a sparse cloud
        m = max(n // 3, 1)
        xyz = np.concatenate([r2.random((n - m, 3)) * 10.0, 5.0 + r2.standard_normal((m, 3)) * 1e-3]).astype(np.float32)
    k = int(rng.choice([1, 3, 8, 16, 16, 25, 32, 50, 64]))
    return kind, np.ascontiguousarray(xyz), k


def main(cases=40, seed=0):
    rng = np.random.default_rng(seed)
    ctx = L.Context(0)
    bad = 0
    t_start = time.time()
    for c in range(cases):
        kind, xyz, k = make(rng)
        n = len(xyz)
        ref = osor.sor(xyz, k, 1.0)
        res = L.sor_filter(xyz, k, 1.0, want_info=True)
        ok = np.array_equal(res["mean_dists"].view(np.uint32), ref["mean_dists"].view(np.uint32)) and np.array_equal(res["mask"], ref["mask"])
        dev_ok = True
        if n >= 2048:
            cols = [np.ascontiguousarray(xyz[:, a]) for a in range(3)]
            d = [ctx.alloc(4 * n).upload(col) for col in cols]
            out = ctx.alloc(4 * n)
            for mf in (0, 1):
                ctx.set_param("filter_mfma", mf)
                ctx.sor_knn(d[0].ptr, d[1].ptr, d[2].ptr, 1, n, 0, n, k, out.ptr, algo=2)
                got = out.download(np.float32, n)
                dev_ok &= np.array_equal(got.view(np.uint32), ref["mean_dists"].view(np.uint32))
            for a in d + [out]:
                a.free()
        tree_ok, tinfo = True, None
        if n > k:   # the Morton-tree path asked for explicitly (adaptive mode above only takes it for uneven clouds)
            rt = L.sor_filter(xyz, k, 1.0, algo=3, want_info=True)
            tinfo = rt["info"]
            tree_ok = np.array_equal(rt["mean_dists"].view(np.uint32), ref["mean_dists"].view(np.uint32)) and np.array_equal(rt["mask"], ref["mask"])
        bad += not (ok and dev_ok and tree_ok)
        print("%3d %-10s n=%7d k=%2d host=%s(algo %d) dev(mf0,mf1)=%s tree=%s deferred=%d refined=%d%s" % (
            c, kind, n, k, "ok" if ok else "MISMATCH", res["info"]["algo"], "ok" if dev_ok else "MISMATCH", "ok" if tree_ok else "MISMATCH",
            res["info"]["n_deferred_bricks"], res["info"]["n_refined"],
            "" if tinfo is None else " leaves=%d near=%d descents=%d" % (tinfo["n_bricks"], tinfo["n_fallback"], tinfo["n_exhaustive"])), flush=True)
    ctx.close()
    print("fuzz: %d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t_start))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
