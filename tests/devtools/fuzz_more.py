"""Randomised sweep #2 on the GPU box: what fuzz_parity.py does not reach -- shares of the bricks / leaves (replicated multi-GPU
exchange), sub-range queries, brute force, the density filter through the drop-in class (eager and device chain) and density -> SOR
chained on the device -- against the oracle.  usage: python tests/devtools/fuzz_more.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib
from oracle import density as oden, sor as osor
import fuzz_parity as fz


def table(xyz):
    arr = np.zeros(len(xyz), dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("i", "i8")])
    arr["x"], arr["y"], arr["z"], arr["i"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], np.arange(len(xyz))
    return arr


def main(cases=40, seed=0):
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for c in range(cases):
        kind, xyz, k = fz.make(rng)
        n = len(xyz)
        if n < 64:
            continue
        what = []
        ref = osor.mean_dists_ckdtree(xyz, k)
        ctx = L.Context(0)
        rows = ctx.alloc(xyz.nbytes).upload(xyz)
        # shares (grid bricks with / without adaptive mode, tree leaves)
        nshares = int(rng.integers(2, 7))
        for algo, ad in ((2, 0), (2, 1), (3, 0)):
            if algo == 3 and n <= k:
                continue
            ctx.set_param("adaptive", ad)
            out = ctx.alloc(4 * n)
            total, filled = np.zeros(n, np.float32), 0
            for s in range(nshares):
                ctx.sor_knn_share(rows.ptr, rows.ptr + 4, rows.ptr + 8, 3, n, k, s, nshares, out.ptr, algo=algo)
                part = out.download(np.float32, n)
                total = total + part
            if not np.array_equal(total.view(np.uint32), ref.view(np.uint32)):
                what.append("shares(algo %d adaptive %d x%d)" % (algo, ad, nshares))
            out.free()
        # sub-range queries, every algorithm
        q0 = int(rng.integers(0, n - 1)); qc = int(rng.integers(1, n - q0 + 1))
        for algo in (1, 2, 3):
            if (algo == 1 and n > 20000) or (algo == 3 and n <= k):
                continue
            ctx.set_param("adaptive", int(rng.integers(0, 2)))
            out = ctx.alloc(4 * qc)
            ctx.sor_knn(rows.ptr, rows.ptr + 4, rows.ptr + 8, 3, n, q0, qc, k, out.ptr, algo=algo)
            if not np.array_equal(out.download(np.float32, qc).view(np.uint32), ref[q0:q0 + qc].view(np.uint32)):
                what.append("subrange(algo %d)" % algo)
            out.free()
        rows.free(); ctx.close()
        # density filter (eager class and device chain), then SOR on the survivors
        vs = float(rng.choice([0.05, 0.3, 1.0, 2.5])) * (float(np.ptp(xyz, axis=0).max()) / 10.0 + 1e-6)
        tp = float(rng.choice([0.001, 0.01, 0.05, 0.32]))
        multi = bool(rng.integers(0, 2))
        try:
            dref = oden.density_filter(xyz, vs, tp, keep_multicluster=multi)
        except Exception as e:   # the oracle refuses what the reference refuses (e.g. overflow of the voxel index)
            dref = None
        if dref is not None and dref["unique_voxels"] < 3_000_000:
            for lazy in (False, True):
                dp = gsx.processing.data_processor.ChainedDataProcessor(table(xyz)) if lazy else gsx.DataProcessor(table(xyz))
                dp.apply_density_filter(vs, tp, keep_multicluster=multi)
                want = np.nonzero(dref["mask"])[0]
                if lazy and len(want) > k + 1:
                    dp.remove_flyers(k, 1.0)
                    sref = osor.sor(xyz[want], k, 1.0)
                    want = want[sref["mask"]]
                got = dp.data["i"]
                if not np.array_equal(got, want):
                    what.append("density%s(%g,%g,%s): %d vs %d" % ("+sor chain" if lazy else "", vs, tp, multi, len(got), len(want)))
        bad += bool(what)
        print("%3d %-10s n=%7d k=%2d %s" % (c, kind, n, k, "ok" if not what else "MISMATCH " + "; ".join(what)), flush=True)
    print("fuzz_more: %d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
