#!/bin/bash
# round 6, call A: parity of the rewritten detect-stage kernels, A B A B against the committed tree, LDS-DMA probe
mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
tools/bin/glds_probe > $O/glds_probe.txt 2>&1; cat $O/glds_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "surf or dll or fused or full_size or config4 or dendritic or resident or zirconcl or keypoint or tie" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for L in A B A B; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
echo "== B rows2 off"; VFSMS_HESSIAN_ROWS2=0 VFSMS_LIB=build_ab/B.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2 | tee -a $O/ab.txt
