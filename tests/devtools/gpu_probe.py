"""Exploratory timing on the GPU box (not a test, not the bench): per-slot kernel times of the
SOR pipeline for a sweep of cloud sizes / k / grid density.  Usage: python tests/devtools/gpu_probe.py [quick]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib


def uniform(n, extent, seed=0):
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32) * np.float32(extent)


def run(ctx, xyz, k, algo, m=None, reps=3, label=""):
    n = len(xyz)
    if m is not None:
        ctx.set_param("grid_points_per_cell", m)
    cols = [np.ascontiguousarray(xyz[:, a]) for a in range(3)]
    d = [ctx.alloc(4 * n).upload(c) for c in cols]
    out = ctx.alloc(4 * n)
    st = ctx.alloc(16)
    mk = ctx.alloc(n + 4)
    info = ctx.sor_knn(d[0].ptr, d[1].ptr, d[2].ptr, 1, n, 0, n, k, out.ptr, algo=algo, want_info=True)  # warm
    ctx.set_timing(True)
    ctx.reset_timing()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.sor_knn(d[0].ptr, d[1].ptr, d[2].ptr, 1, n, 0, n, k, out.ptr, algo=algo)
        ctx.sor_stats(out.ptr, n, 1.0, st.ptr)
        ctx.sor_mask(out.ptr, n, st.ptr + 8, mk.ptr)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    slots = {name: ctx.timing(i) for i, name in enumerate(["knn", "bin", "fallback", "stats"])}
    ctx.set_timing(False)
    msg = " ".join("%s=%.3fms" % (k_, v[1] / max(v[0], 1) * (v[0] / reps)) for k_, v in slots.items())
    print("%-28s n=%9d k=%2d algo=%d m=%s wall=%.3f ms  %.1f Msplat/s | %s | dims=%s cells=%d bricks=%d fallback=%d exh=%d"
          % (label, n, k, algo, m, wall, n / wall / 1e3, msg, info["grid_dim"], info["n_cells"], info["n_bricks"],
             info["n_fallback"], info["n_exhaustive"]) + (" deferred_bricks=%d refined_pts=%d" % (info["n_deferred_bricks"], info["n_refined"]) if info.get("n_deferred_bricks") else ""), flush=True)
    for a in d + [out, st, mk]:
        a.free()


def host_level(xyz, k, label):
    """PCIe-inclusive: numpy (N,3) in host memory -> bool mask in host memory (gsx_sor_filter)."""
    L.sor_filter(xyz, k, 1.0, want_mean=False)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        L.sor_filter(xyz, k, 1.0, want_mean=False)
    dt = (time.perf_counter() - t0) / reps
    print("%-28s n=%9d k=%2d host->host %.3f ms  %.1f Msplat/s (PCIe inclusive)" % (label, len(xyz), k, dt * 1e3, len(xyz) / dt / 1e6), flush=True)


def shares():
    """What one rank of an N-GPU weak-scaling step computes (N x 1M splats gathered, 1/N of the queries):
    index range of queries (round-1 first design) vs share of the grid's bricks (gsx_sor_knn_share_dev)."""
    ctx = L.Context(0)
    for world in (1, 2, 4, 8):
        n = world * 1_000_000
        xyz = uniform(n, 10.0 * world ** (1.0 / 3.0))
        cols = [np.ascontiguousarray(xyz[:, a]) for a in range(3)]
        d = [ctx.alloc(4 * n).upload(c) for c in cols]
        out = ctx.alloc(4 * n)
        nq = n // world
        for mode in ("index", "share"):
            def step():
                if mode == "index":
                    ctx.sor_knn(d[0].ptr, d[1].ptr, d[2].ptr, 1, n, 3 * nq % n if world > 3 else 0, nq, 16, out.ptr, algo=2)
                else:
                    ctx.sor_knn_share(d[0].ptr, d[1].ptr, d[2].ptr, 1, n, 16, world // 2, world, out.ptr, algo=2)
            step()
            ctx.set_timing(True); ctx.reset_timing(); ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                step()
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / 3 * 1e3
            slots = {name: ctx.timing(i) for i, name in enumerate(["knn", "bin", "fallback", "stats"])}
            ctx.set_timing(False)
            print("world=%d n_total=%8d mode=%-5s per-rank compute %.3f ms | %s" % (
                world, n, mode, wall, " ".join("%s=%.3f" % (k_, v[1] / 3) for k_, v in slots.items())), flush=True)
        for a in d + [out]:
            a.free()
    ctx.close()


def ablate():
    ctx = L.Context(0)
    x10 = uniform(10_000_000, 5.0)
    for mf in (1, 0):
        ctx.set_param("filter_mfma", mf)
        for dbg, name in ((0, "full"), (4, "no epilogue"), (7, "skeleton only"), (15, "skeleton, no output write"),
                          (31, "queue + row tables only"), (5, "phase 1 + cursor walk, no insert"), (36, "phase 1 only, no walk"),
                          (2, "no phase 1")):
            ctx.set_param("debug_skip", dbg)
            run(ctx, x10, 16, 2, 0.0, label="ablate mf=%d: %s" % (mf, name))
    ctx.set_param("debug_skip", 0)
    ctx.set_param("filter_mfma", 1)
    ctx.close()


def sweep():
    ctx = L.Context(0)
    x1 = uniform(1_000_000, 10.0)
    x10 = uniform(10_000_000, 5.0)
    for m in (6.5, 7.0, 7.25, 7.5, 7.75, 8.0):
        run(ctx, x1, 16, 2, m, reps=5, label="sweep 1M")
    for m in (6.5, 7.0, 7.25, 7.5, 7.75, 8.0):
        run(ctx, x10, 16, 2, m, label="sweep 10M")
    for m in (10.0, 11.0, 12.2, 13.5):
        run(ctx, x1, 25, 2, m, reps=5, label="sweep 1M k25")
    for m in (3.5, 4.2, 5.0):
        run(ctx, x1, 8, 2, m, reps=5, label="sweep 1M k8")
    ctx.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ablate":
        return ablate()
    if len(sys.argv) > 1 and sys.argv[1] == "shares":
        return shares()
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        return sweep()
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ctx = L.Context(0)
    x1 = uniform(1_000_000, 10.0)
    for m in (7.0, 8.0, 9.0, 0.0):
        run(ctx, x1, 16, 2, m, label="grid 1M")
    for k in (8, 25, 32, 50):
        run(ctx, x1, k, 2, 0.0, label="grid 1M auto-m")
    run(ctx, x1, 25, 2, 8.0, label="grid 1M k25 m8")
    run(ctx, x1, 32, 2, 10.0, label="grid 1M k32 m10")
    run(ctx, uniform(100_000, 10.0), 16, 1, None, label="brute 100k")
    host_level(x1, 16, "host 1M")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import datasets
    for ad in (0, 1):
        ctx.set_param("adaptive", ad)
        run(ctx, datasets.clustered(200_000, 1), 16, 2, 0.0, reps=1, label="clustered 200k adaptive=%d" % ad)
        run(ctx, datasets.clustered(1_000_000, 1), 16, 2, 0.0, reps=1, label="clustered 1M adaptive=%d" % ad)
        run(ctx, x1, 16, 2, 0.0, reps=5, label="uniform 1M adaptive=%d" % ad)
    ctx.set_param("adaptive", 0)
    if not quick:
        run(ctx, x1, 16, 1, None, reps=1, label="brute 1M")
        x10 = uniform(10_000_000, 5.0)
        for m in (7.0, 8.0, 9.0):
            run(ctx, x10, 16, 2, m, label="grid 10M")
        run(ctx, x10, 25, 2, 0.0, label="grid 10M k25")
        host_level(x10, 16, "host 10M")
    ctx.close()


if __name__ == "__main__":
    main()
