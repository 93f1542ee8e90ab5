#!/bin/bash
GSX_TRACE_LEVELS=1 timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/adaptive.log
import sys, os, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import gpu_probe as g
from oracle import datasets
ctx = g.L.Context(0)
ctx.set_param("adaptive", 1)
for name, xyz in (("scene+floaters 1M", datasets.scene_with_floaters(1_000_000, 1)),
                  ("scene+floaters 10M", datasets.scene_with_floaters(10_000_000, 1)),
                  ("clustered 1M", datasets.clustered(1_000_000, 1))):
    g.run(ctx, xyz, 16, 2, 0.0, reps=1, label=name)
ctx.close()
PY
