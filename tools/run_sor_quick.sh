#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sor_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/pytest_sor.log
tail -4 gpurun_out/pytest_sor.log
timeout 600 python tests/devtools/gpu_probe.py ${1:-} 2>&1 | tee gpurun_out/probe.log
