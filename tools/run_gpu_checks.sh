#!/bin/bash
# One gpurun call: GPU parity tests + exploratory timing.  Outputs land in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 600 python tests/devtools/gpu_probe.py ${1:-} > gpurun_out/probe.log 2>&1
echo "probe exit: $?" >> gpurun_out/probe.log
tail -5 gpurun_out/pytest_gpu.log
cat gpurun_out/probe.log
