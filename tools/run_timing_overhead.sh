#!/bin/bash
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/timing_overhead.log
import sys, time
sys.path.insert(0, "tests/devtools"); sys.path.insert(0, ".")
import gpu_probe as g
import numpy as np
ctx = g.L.Context(0)
for n, ext, reps in ((1_000_000, 10.0, 200), (10_000_000, 5.0, 20)):
    xyz = g.uniform(n, ext)
    cols = [np.ascontiguousarray(xyz[:, a]) for a in range(3)]
    d = [ctx.alloc(4 * n).upload(c) for c in cols]
    out = ctx.alloc(4 * n); st = ctx.alloc(16); mk = ctx.alloc(n + 4)
    def step():
        ctx.sor_knn(d[0].ptr, d[1].ptr, d[2].ptr, 1, n, 0, n, 16, out.ptr, algo=2)
        ctx.sor_stats(out.ptr, n, 1.0, st.ptr)
        ctx.sor_mask(out.ptr, n, st.ptr + 8, mk.ptr)
    for timing in (False, True, False, True):
        ctx.set_timing(timing); ctx.reset_timing()
        for _ in range(5): step()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): step()
        ctx.synchronize()
        print("n=%d timing_events=%s step %.4f ms" % (n, timing, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    ctx.set_timing(False)
    for a in d + [out, st, mk]: a.free()
ctx.close()
PY
