"""Secondary measurements for DESIGN.md (NOT the bench line): BASELINE.json configs[2] and configs[4].

  config 3: 10M splats (L=5), density sensitivity 0.5 then SOR k=16 sigma=1 on the survivors
  config 5: 10M-splat SOG K-Means: 2 scalar codebooks (50k x 1, K=256, 20 it), 64 chunks x
            (156 250 x 45, K=1024, 10 it), quantise 30M scalars x 2
Host-level API (numpy in / numpy out, PCIe included) + per-slot kernel time from the library.
usage: python tests/devtools/bench_configs.py [3] [5] [cpu]
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib


def t(f, reps=1):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t0) / reps, r


def config3(cpu):
    n = 10_000_000
    xyz = np.random.default_rng(0).random((n, 3), dtype=np.float32) * np.float32(5.0)
    arr = np.zeros(n, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4")])
    arr["x"], arr["y"], arr["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]

    def run():
        p = gsx.DataProcessor(arr)
        t0 = time.perf_counter()
        p.apply_density_filter(sensitivity=0.5)
        t1 = time.perf_counter()
        n1 = len(p.data)
        p.remove_flyers(16, 1.0)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, n1, len(p.data)
    run()
    d, s, n1, n2 = run()
    print("config3 GPU (DataProcessor API, structured-array in/out, PCIe + AoS gather + boolean-index compaction included): "
          "density %.1f ms, SOR %.1f ms -> %d -> %d -> %d splats ; %.1f Msplat/s end to end"
          % (d * 1e3, s * 1e3, n, n1, n2, n / (d + s) / 1e6), flush=True)
    cols = (np.ascontiguousarray(arr["x"]), np.ascontiguousarray(arr["y"]), np.ascontiguousarray(arr["z"]))
    dt, occ = t(lambda: L.density_voxels(cols, 1.1, int(n * 0.55 / 100)))
    print("  gsx_density_voxels (host cols in) %.2f ms, %d unique, %d dense" % (dt * 1e3, occ["n_unique"], len(occ["dense_keys"])))
    dt, _ = t(lambda: L.density_mask(cols, 1.1, occ["dense_keys"]))
    print("  gsx_density_mask (host cols in)   %.2f ms" % (dt * 1e3))
    if cpu:
        from oracle import density as oden, sor as osor
        t0 = time.perf_counter()
        dres = oden.density_filter(xyz, sensitivity=0.5)
        t1 = time.perf_counter()
        sres = osor.sor(xyz[dres["mask"]], 16, 1.0)
        t2 = time.perf_counter()
        print("config3 CPU oracle (%d cores): density %.2f s, SOR %.2f s ; survivors %d -> %d"
              % (os.cpu_count(), t1 - t0, t2 - t1, int(dres["mask"].sum()), int(sres["mask"].sum())), flush=True)
        assert int(dres["mask"].sum()) == n1 and int(sres["mask"].sum()) == n2


def config5(cpu):
    n = 10_000_000
    rng = np.random.default_rng(0)
    ctx = L.Context(0)
    ctx.set_timing(True)
    from oracle import kmeans as okm
    plan = okm.sog_sh_plan(n, 2)
    print("config5 plan", plan)
    # (ii) SH palette: 64 independent chunks, data generated chunk by chunk (28 MB each)
    np.random.seed(0)
    total = 0.0
    inertia = []
    for c in range(plan["num_chunks"]):
        data = (rng.standard_normal((plan["chunk_size"], 45)) * 0.1).astype(np.float32)
        init = data[np.random.choice(len(data), plan["k_per_chunk"], replace=False)]
        t0 = time.perf_counter()
        cent, lab = L.kmeans_lloyd(data, init, 10)
        total += time.perf_counter() - t0
        if c == 0:
            inertia.append(okm.inertia(data, cent, lab))
            first = (data, init)
    print("config5 (ii) 64 chunks x (156250 x 45, K=1024, 10 it): %.3f s host-level (%.1f ms/chunk, PCIe included); "
          "inertia chunk0 %.4f" % (total, total / 64 * 1e3, inertia[0]), flush=True)
    # (i) scalar codebooks + (iii) quantise
    scal = (rng.standard_normal(3 * n) - 4).astype(np.float32)
    fit = scal[np.random.choice(len(scal), 50000, replace=False)].reshape(-1, 1)
    dt, (cb, _) = t(lambda: L.kmeans_lloyd(fit, fit[np.random.choice(50000, 256, replace=False)], 20))
    print("config5 (i) 50000 x 1, K=256, 20 it: %.2f ms" % (dt * 1e3))
    cbs = np.sort(cb.ravel())
    dt, idx = t(lambda: L.quantize_sorted_codebook(scal, cbs))
    print("config5 (iii) quantise 30M scalars: %.2f ms host-level (%.0f Mscalar/s)" % (dt * 1e3, 3 * n / dt / 1e6), flush=True)
    if cpu:
        from sklearn.cluster import MiniBatchKMeans
        data, init = first
        t0 = time.perf_counter()
        km = MiniBatchKMeans(n_clusters=1024, max_iter=10, batch_size=16384, n_init="auto", compute_labels=True).fit(data)
        dt = time.perf_counter() - t0
        print("config5 CPU reference fallback (_kmeans_sklearn, gpu_ops.py:48-52) one chunk: %.2f s (x64 = %.1f s), inertia %.4f"
              % (dt, dt * 64, okm.inertia(data, km.cluster_centers_, km.labels_)), flush=True)
        t0 = time.perf_counter()
        ref = okm.quantize_to_codebook(scal, cbs)
        print("config5 CPU quantize_to_codebook 30M: %.2f s ; identical %s" % (time.perf_counter() - t0, bool(np.array_equal(ref, idx))))
    for name, slot in (("assign", L.T_KMEANS_ASSIGN), ("update", L.T_KMEANS_UPDATE)):
        pass
    ctx.close()


if __name__ == "__main__":
    args = sys.argv[1:]
    cpu = "cpu" in args
    if "3" in args or not [a for a in args if a in ("3", "5")]:
        config3(cpu)
    if "5" in args or not [a for a in args if a in ("3", "5")]:
        config5(cpu)
