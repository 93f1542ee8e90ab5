"""Randomised sweep #4: fewer, LARGER clouds (0.3M ... 3M points) -- more rim bricks, more multi-batch bricks, more leaves per
launch -- device API (grid, both filters), adaptive host entry, tree path; three repetitions each (the order inside a cell varies
from run to run).  usage: python tests/devtools/fuzz_large.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib
from oracle import datasets, sor as osor


def make(rng):
    kind = rng.choice(["uniform", "plane", "clustered", "scene", "slab", "shell"])
    n = int(rng.integers(300_000, 3_000_000))
    seed = int(rng.integers(0, 1 << 30))
    r2 = np.random.default_rng(seed)
    if kind == "uniform":
        xyz = datasets.uniform(n, float(rng.choice([1.0, 10.0, 1e3])), seed)
    elif kind == "plane":
        xyz = (r2.random((n, 3)) * np.array([10.0, float(rng.choice([10.0, 2.0])), 0.0])).astype(np.float32)
    elif kind == "clustered":
        xyz = datasets.clustered(n, seed)
    elif kind == "scene":
        xyz = datasets.scene_with_floaters(n, seed, far=float(rng.choice([50.0, 500.0])))
    elif kind == "slab":
        xyz = (r2.random((n, 3)) * np.array([20.0, 10.0, 0.05])).astype(np.float32)
    else:
        v = r2.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        xyz = (v * (5.0 + 0.01 * r2.standard_normal((n, 1)))).astype(np.float32)
    return kind, np.ascontiguousarray(xyz), int(rng.choice([8, 16, 25, 32, 50, 64]))


def main(cases=10, seed=0):
    rng = np.random.default_rng(seed)
    bad, t0 = 0, time.time()
    ctx = L.Context(0)
    for c in range(cases):
        kind, xyz, k = make(rng)
        n = len(xyz)
        ref = osor.mean_dists_ckdtree(xyz, k)
        rows = ctx.alloc(xyz.nbytes).upload(xyz)
        out = ctx.alloc(4 * n)
        what = []
        for rep in range(3):
            for name, algo, ad, mf in (("grid mf0", 2, 0, 0), ("grid mf1", 2, 0, 1), ("adaptive", 0, 1, 1), ("tree", 3, 0, 1)):
                if name.startswith("grid") and kind in ("clustered", "scene"):
                    continue   # a single grid on these clouds is quadratic: adaptive mode is the product path
                ctx.set_param("adaptive", ad); ctx.set_param("filter_mfma", mf)
                ctx.sor_knn(rows.ptr, rows.ptr + 4, rows.ptr + 8, 3, n, 0, n, k, out.ptr, algo=algo)
                got = out.download(np.float32, n)
                nb = int(np.count_nonzero(got.view(np.uint32) != ref.view(np.uint32)))
                if nb:
                    what.append("%s rep %d: %d wrong" % (name, rep, nb))
        rows.free(); out.free()
        bad += bool(what)
        print("%3d %-9s n=%8d k=%2d %s" % (c, kind, n, k, "ok" if not what else "MISMATCH " + "; ".join(what)), flush=True)
    ctx.close()
    print("fuzz_large: %d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
