#!/bin/bash
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/mf_count.log
import sys, os
sys.path.insert(0, "tests/devtools"); sys.path.insert(0, ".")
import gpu_probe as g
ctx = g.L.Context(0)
x1 = g.uniform(1_000_000, 10.0)
for mf in (1, 0):
    ctx.set_param("filter_mfma", mf)
    ctx.set_param("debug_skip", 64)
    g.run(ctx, x1, 16, 2, 0.0, reps=1, label="count mf=%d 1M" % mf)
ctx.close()
PY
