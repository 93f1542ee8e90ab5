"""Randomised sweep #3 on the GPU box: the K-Means entry points over random shapes (N, D, K -- including the shapes the matrix-core
assign is not built for), data kinds (normal, clustered, duplicate rows, constant columns, large offsets) against the oracle:
first assign (labels equal unless the two distances are within the float32 noise), Lloyd trajectory (inertia), the quantiser
(integer-exact), the scalar solver (inertia <= its restatement's + noise).  usage: python tests/devtools/fuzz_kmeans.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib
from oracle import kmeans as okm


def make(rng):
    n = int(rng.choice([rng.integers(2, 300), rng.integers(300, 6000), rng.integers(6000, 80000)]))
    d = int(rng.choice([1, 2, 3, 7, 9, 16, 24, 33, 45, 48, 64, int(rng.integers(1, 65))]))
    k = int(min(n - 1, rng.choice([1, 2, 5, 17, 64, 100, 128, 256, 1000, 1024, 1500, 2048, int(rng.integers(1, 600))])))
    k = max(k, 1)
    kind = rng.choice(["normal", "clustered", "dups", "constcol", "offset", "tiny"])
    r2 = np.random.default_rng(int(rng.integers(0, 1 << 30)))
    x = r2.standard_normal((n, d)).astype(np.float32)
    if kind == "clustered":
        c = r2.standard_normal((max(k // 3, 1), d)) * 5
        x = (c[r2.integers(0, len(c), n)] + 0.05 * r2.standard_normal((n, d))).astype(np.float32)
    elif kind == "dups":
        x = x[r2.integers(0, max(n // 4, 1), n)]
    elif kind == "constcol":
        x[:, r2.integers(0, d)] = np.float32(3.25)
    elif kind == "offset":
        x = (x * 0.01 + 1000.0).astype(np.float32)
    elif kind == "tiny":
        x = (x * 1e-6).astype(np.float32)
    return kind, np.ascontiguousarray(x), k, int(rng.integers(1, 5)), r2


def main(cases=40, seed=0):
    rng = np.random.default_rng(seed)
    bad, t0 = 0, time.time()
    for c in range(cases):
        kind, x, k, iters, r2 = make(rng)
        n, d = x.shape
        what = []
        init = x[r2.choice(n, k, replace=False)].copy()
        # first assign
        _, lab = L.kmeans_lloyd(x, init, 1)
        ref = okm.assign(x, init)
        if not ((lab >= 0).all() and (lab < k).all()):
            what.append("labels out of range")
        else:
            idx, gap = okm.assign_margin(x, init, lab, ref)
            # both distances computed in float32 by the reference: equal up to the noise of ||x||^2-sized terms
            noise = 1e-5 * (1.0 + float(np.abs(x).max()) ** 2 * d / max(float(((x[idx] - init[ref[idx]]) ** 2).sum(1).min()) if len(idx) else 1.0, 1e-30))
            if len(idx) and not (gap <= max(1e-5, min(noise, 1e-2))).all():
                what.append("assign: %d labels differ, worst relative gap %.3g" % (len(idx), gap.max()))
        # trajectory
        cent, labels = L.kmeans_lloyd(x, init, iters)
        rcent, rlabels, _ = okm.lloyd(x, init, iters)
        ig, ir = okm.inertia(x, cent, labels), okm.inertia(x, rcent, rlabels)
        if not np.isfinite(cent).all():
            what.append("non-finite centroids")
        elif abs(ig - ir) > 2e-3 * max(ir, 1e-30) + 1e-12:
            what.append("lloyd x%d: inertia %.6g vs %.6g" % (iters, ig, ir))
        # quantiser (sorted codebook of up to 256 entries, values incl. exact codebook entries and far outliers)
        cb = np.sort(r2.standard_normal(int(r2.integers(1, 257))).astype(np.float32))
        vals = np.concatenate([r2.standard_normal(5000).astype(np.float32), cb, np.float32([-1e9, 1e9, 0.0])])
        got = gsx.gpu_ops.quantize_to_codebook(vals, cb)
        if not np.array_equal(np.asarray(got, np.uint8), okm.quantize_to_codebook(vals, cb)):
            what.append("quantize")
        # scalar solver
        if d == 1 and k <= 1024 and n > k:
            cent1, _ = L.kmeans1d(x.reshape(-1), k, iters=20, want_labels=True)
            rc, _ = okm.kmeans1d_sorted(x.reshape(-1), k, 20)
            i1, i2 = okm.inertia_1d(x.reshape(-1), cent1), okm.inertia_1d(x.reshape(-1), rc)
            if not (i1 <= i2 * (1 + 1e-6) + 1e-12):
                what.append("kmeans1d inertia %.6g vs restatement %.6g" % (i1, i2))
        bad += bool(what)
        print("%3d %-9s n=%6d d=%2d k=%4d it=%d %s" % (c, kind, n, d, k, iters, "ok" if not what else "MISMATCH " + "; ".join(what)), flush=True)
    print("fuzz_kmeans: %d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
