#!/bin/bash
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/probe_50m.log
import sys
sys.path.insert(0, "tests/devtools"); sys.path.insert(0, ".")
import gpu_probe as g
ctx = g.L.Context(0)
x = g.uniform(50_000_000, 10.0)
g.run(ctx, x, 32, 2, 0.0, reps=2, label="50M k=32 (configs[3] on ONE GPU)")
g.run(ctx, x, 16, 2, 0.0, reps=2, label="50M k=16")
ctx.close()
PY
