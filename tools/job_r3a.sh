set -x
mkdir -p gpurun_out/r3a
timeout 600 python -m pytest tests -m gpu -x -q -k "surf or dll or fused or full_size or config4 or dendritic or resident" > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -15 gpurun_out/r3a/pytest.log
for v in 1 0 1 0; do echo "== LEGACY=$v"; VFSMS_DESC_LEGACY=$v timeout 200 python tools/microbench.py 16 50 2>&1 | tail -2; done | tee gpurun_out/r3a/micro.log
