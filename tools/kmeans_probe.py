"""Device-resident K-Means timing (BASELINE.json configs[4] chunk: 156 250 x 45, K=1024, 10 it)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsx = importlib.import_module("3dgsconverter_amd")
L = gsx._lib
import ctypes as C

def run(n, d, k, iters, reps=3):
    rng = np.random.default_rng(0)
    data = (rng.standard_normal((n, d)) * 0.1).astype(np.float32)
    init = data[rng.choice(n, k, replace=False)]
    ctx = L.Context(0)
    dd = ctx.alloc(data.nbytes).upload(data)
    dc = ctx.alloc(init.nbytes)
    dl = ctx.alloc(4 * n)
    lib = L.load()
    def once():
        dc.upload(init)
        L.check(lib.gsx_kmeans_lloyd_dev(ctx.handle, dd.ptr, n, d, k, iters, dc.ptr, dl.ptr), "kmeans")
    once(); ctx.synchronize()
    ctx.set_timing(True); ctx.reset_timing()
    t0 = time.perf_counter()
    for _ in range(reps): once()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps
    na, ma = ctx.timing(L.T_KMEANS_ASSIGN); nu, mu = ctx.timing(L.T_KMEANS_UPDATE)
    ops = 2.0 * n * k * d
    print("kmeans n=%d d=%d k=%d it=%d: wall %.2f ms | assign %.3f ms/it (%.1f%% of 78.6T lane-op/s) update %.3f ms/it"
          % (n, d, k, iters, wall * 1e3, ma / na, 100 * ops / (ma / na * 1e-3) / 78.6e12, mu / nu), flush=True)
    cent = dc.download(np.float32, k * d).reshape(k, d); lab = dl.download(np.int32, n)
    ctx.close()
    return data, init, cent, lab

if __name__ == "__main__":
    run(156250, 45, 1024, 10)
    run(156250, 24, 1024, 10)
    run(156250, 9, 1024, 10)
    run(50000, 1, 256, 20)
    run(30_000_000, 1, 256, 10, reps=1)   # SOG scales/colours codebook: 10M x 3 flattened
    run(10_000_000, 3, 256, 10, reps=1)
