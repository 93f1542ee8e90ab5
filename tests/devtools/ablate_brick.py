import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
gsx = importlib.import_module("3dgsconverter_amd"); L = gsx._lib
n = 10_000_000
xyz = np.random.default_rng(0).random((n, 3), dtype=np.float32) * np.float32(5.0)
ctx = L.Context(0)
d = ctx.alloc(xyz.nbytes).upload(xyz); out = ctx.alloc(4 * n + 16)
for dbg in (0, 16, 32, 2, 34, 8, 1, 4):
    ctx.set_param("debug_skip", dbg)
    ctx.set_param("timing_mask", 1 << L.T_SOR_KNN); 
    for _ in range(2): ctx.sor_knn(d.ptr, d.ptr + 4, d.ptr + 8, 3, n, 0, n, 16, out.ptr)
    ctx.synchronize(); ctx.set_timing(True); ctx.reset_timing()
    for _ in range(8): ctx.sor_knn(d.ptr, d.ptr + 4, d.ptr + 8, 3, n, 0, n, 16, out.ptr)
    ctx.synchronize()
    cnt, ms = ctx.timing(L.T_SOR_KNN); ctx.set_timing(False)
    print("debug_skip=%2d knn_brick %.4f ms" % (dbg, ms / max(cnt, 1)), flush=True)
