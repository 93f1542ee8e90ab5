"""GPU box: the Morton-tree KNN (algo 3, csrc/sor_tree.hip) against cKDTree on small clouds, then timings on the clouds a
uniform grid is bad at.    python tests/devtools/probe_tree.py [check|time] ..."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def clouds():
    from oracle import datasets
    rng = np.random.default_rng(5)
    yield "uniform20k", datasets.uniform(20000, 10.0, 1), 16
    yield "uniform100k_k8", datasets.uniform(100000, 10.0, 2), 8
    yield "uniform50k_k32", datasets.uniform(50000, 10.0, 3), 32
    yield "uniform30k_k64", datasets.uniform(30000, 10.0, 4), 64
    yield "uniform30k_k25", datasets.uniform(30000, 10.0, 4), 25
    yield "clustered200k", bench.synth_clustered(200000, 0), 16
    yield "floaters300k", bench.synth_scene_with_floaters(300000, 0), 16
    d = rng.random((5000, 3)).astype(np.float32)
    yield "dups", np.concatenate([d, d[:2000], np.repeat(d[:3], 300, axis=0)]), 8
    p = rng.random((30000, 3)).astype(np.float32); p[:, 2] = 0.5
    yield "plane", p, 16
    l = np.zeros((3000, 3), np.float32); l[:, 0] = rng.random(3000)
    yield "line", l, 5
    yield "tiny40", rng.random((40, 3)).astype(np.float32), 16
    yield "far_offset", (datasets.uniform(40000, 1.0, 7) + np.float32(5000.0)).astype(np.float32), 16
    g = np.stack(np.meshgrid(*[np.arange(17, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    yield "lattice", g, 6


def check(L):
    from oracle import sor as osor
    bad = 0
    for name, xyz, k in clouds():
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        t0 = time.perf_counter()
        res = L.sor_filter(xyz, k, 1.0, algo=3, want_info=True)
        dt = time.perf_counter() - t0
        ref = osor.mean_dists_ckdtree(xyz, k)
        md = res["mean_dists"]
        diff = np.nonzero(md.view(np.uint32) != ref.view(np.uint32))[0]
        print("%-16s n=%7d k=%2d: %s  leaves=%d fallback=%d  (%.1f ms)" % (
            name, len(xyz), k, "OK" if len(diff) == 0 else "MISMATCH %d first %s gpu %s ref %s" % (len(diff), diff[:5], md[diff[:5]], ref[diff[:5]]),
            res["info"]["n_bricks"], res["info"]["n_fallback"], dt * 1e3), flush=True)
        bad += len(diff) > 0
    print("check:", "all exact" if bad == 0 else "%d clouds differ" % bad)
    return bad


def timeit(L, kind, n, tree, reps=3):
    ctx = L.Context(0)
    xyz = bench.synth_clustered(n, 0) if kind == "clustered" else (bench.synth_scene_with_floaters(n, 0) if kind == "floaters" else bench.synth_uniform(n, 10.0, 0))
    ctx.set_param("adaptive", 1)
    ctx.set_param("tree", tree)
    for kv in sys.argv[5:]:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    b = bench.SorBench(L, ctx, xyz, int(os.environ.get("PROBE_K", "16")), 1.0)
    b.step(); ctx.synchronize()
    os.environ["GSX_TRACE_LEVELS"] = "1"
    ts = []
    for r in range(reps):
        if r > 0:
            os.environ.pop("GSX_TRACE_LEVELS", None)
        t0 = time.perf_counter(); b.step(); ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ctx.set_timing(True); ctx.reset_timing(); b.step(); ctx.synchronize()
    parts = {nm: ctx.timing(getattr(L, nm))[1] for nm in ("T_SOR_BIN", "T_SOR_KNN", "T_SOR_FALLBACK", "T_SOR_STATS")}
    ctx.set_timing(False)
    mask, stats = b.results()
    print("%s %d tree=%d: step %s ms  parts %s survivors %d thr %r" % (kind, n, tree, ["%.3f" % t for t in ts], {k: round(v, 3) for k, v in parts.items()}, int(mask.sum()), float(stats[2])), flush=True)


def main():
    gsx = importlib.import_module("3dgsconverter_amd")
    L = gsx._lib
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(1 if check(L) else 0)
    kind, n = sys.argv[2], int(sys.argv[3])
    for tree in ([1, 0] if len(sys.argv) < 5 else [int(sys.argv[4])]):
        timeit(L, kind, n, tree)


if __name__ == "__main__":
    main()
