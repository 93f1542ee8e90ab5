#!/bin/bash
GSX_TRACE_LEVELS=1 timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/adaptive.log
import sys, os, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import gpu_probe as g
from oracle import datasets
ctx = g.L.Context(0)
clouds = (("scene+floaters 1M", datasets.scene_with_floaters(1_000_000, 1)),
          ("scene+floaters 10M", datasets.scene_with_floaters(10_000_000, 1)),
          ("clustered 1M", datasets.clustered(1_000_000, 1)))
os.environ.pop("GSX_TRACE_LEVELS", None)
for ad in (1, 2):
    for dw in (512, 128, 64, 32):
        ctx.set_param("adaptive", ad); ctx.set_param("defer_words", dw)
        for name, xyz in clouds:
            g.run(ctx, xyz, 16, 2, 0.0, reps=1, label="%s clip=%d dw=%d" % (name, ad == 1, dw))
ctx.close()
PY
