#!/bin/bash
# A/B of the knn_brick phase-1 filters on the GPU box: parity subset + 1M/10M timings for filter_mfma = 1, 0
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sor_gpu.py -m gpu -x -q -p no:cacheprovider -k "golden_small or kat_grid or k_buckets or clustered or degenerate" 2>&1 | tail -3
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/mf_ab.log
import sys, os
sys.path.insert(0, "tests/devtools"); sys.path.insert(0, ".")
import gpu_probe as g
ctx = g.L.Context(0)
x1 = g.uniform(1_000_000, 10.0); x10 = g.uniform(10_000_000, 5.0)
for mf in (1, 0, 1, 0):
    ctx.set_param("filter_mfma", mf)
    g.run(ctx, x1, 16, 2, 0.0, reps=10, label="mf=%d 1M" % mf)
    g.run(ctx, x10, 16, 2, 0.0, reps=5, label="mf=%d 10M" % mf)
for mf in (1, 0):
    ctx.set_param("filter_mfma", mf)
    g.run(ctx, x1, 8, 2, 0.0, reps=5, label="mf=%d 1M k8" % mf)
    g.run(ctx, x1, 32, 2, 0.0, reps=5, label="mf=%d 1M k32" % mf)
    from oracle import datasets
    g.run(ctx, datasets.clustered(1_000_000, 1), 16, 2, 0.0, reps=1, label="mf=%d clustered 1M" % mf)
ctx.close()
PY
