#!/bin/bash
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/adaptive.log
import sys, os, time
sys.path.insert(0, "tests/devtools"); sys.path.insert(0, ".")
import gpu_probe as g
from oracle import datasets
ctx = g.L.Context(0)
clouds = (("scene+floaters 1M", datasets.scene_with_floaters(1_000_000, 1)),
          ("scene+floaters 10M", datasets.scene_with_floaters(10_000_000, 1)),
          ("clustered 1M", datasets.clustered(1_000_000, 1)))
ctx.set_param("adaptive", 1)
for name, xyz in clouds[2:]:
    g.run(ctx, xyz, 16, 2, 0.0, reps=1, label=name)
os.environ.pop("GSX_TRACE_LEVELS", None)
for name, xyz in clouds:
    g.run(ctx, xyz, 16, 2, 0.0, reps=2, label=name)
g.run(ctx, g.uniform(1_000_000, 10.0), 16, 2, 0.0, reps=5, label="uniform 1M adaptive=1")
g.host_level(g.uniform(1_000_000, 10.0), 16, "host 1M (adaptive default)")
ctx.close()
PY
