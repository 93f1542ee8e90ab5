"""GPU box: ONE SOG palette chunk (156 250 x 45, K = 1024), a few Lloyd iterations -- small enough for rocprofv3 --pmc
(counter passes serialise every dispatch: never profile the 640-iteration bench with counters).  usage: probe_assign.py [iters]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
gsx = importlib.import_module("3dgsconverter_amd"); L = gsx._lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(0)
x = (rng.standard_normal((156250, 45)) * 0.1).astype(np.float32)
init = x[rng.choice(len(x), 1024, replace=False)]
ctx = L.Context(0)
for kv in sys.argv[2:]:
    ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
d, c, l = ctx.alloc(x.nbytes).upload(x), ctx.alloc(init.nbytes).upload(np.ascontiguousarray(init)), ctx.alloc(4 * len(x) + 16)
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter()
    L.check(ctx.lib.gsx_kmeans_lloyd_dev(ctx.handle, d.ptr, len(x), 45, 1024, iters, c.ptr, l.ptr), "lloyd")
    ctx.synchronize()
    print("lloyd %d iterations: %.3f ms" % (iters, (time.perf_counter() - t0) * 1e3))
