#!/bin/bash
# One gpurun call: the SURF-side parity tests, then the fixed micro batch (stage times) -- the loop used while working on a kernel.
#   gpurun --timeout 1200 -- 'bash tools/gpu_check.sh [pytest -k expression]'
mkdir -p gpurun_out/check
K=${1:-"surf or dll or fused or full_size or config4 or dendritic or resident"}
timeout 900 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/check/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/check/pytest.log
tail -4 gpurun_out/check/pytest.log
for v in 1 2; do timeout 200 python tools/microbench.py 16 50 2>&1 | tail -2; done | tee gpurun_out/check/micro.log
