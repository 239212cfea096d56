#!/bin/bash
mkdir -p gpurun_out/r06n
O=gpurun_out/r06n
timeout 900 python -m pytest tests -m gpu -x -q -k "surf or dll or fused or full_size or config4 or dendritic or zirconcl or tie or keypoint or resident" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in CUR NMSW HCOL NOSYNC CUR NMSW HCOL NOSYNC; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
