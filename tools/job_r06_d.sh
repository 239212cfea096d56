#!/bin/bash
# round 6, call D: orientation weight table + NMS rows in flight (E: 3 ahead, F: 6 ahead) + border units on one gather, against call C's build
mkdir -p gpurun_out/r06d
O=gpurun_out/r06d
timeout 900 python -m pytest tests -m gpu -x -q -k "surf or dll or fused or full_size or config4 or dendritic or zirconcl or tie or keypoint" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in C E F C E F; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
