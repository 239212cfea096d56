#!/bin/bash
mkdir -p gpurun_out/r06k
O=gpurun_out/r06k
timeout 600 python -m pytest tests -m gpu -x -q -k "surf or dll or fused or config4 or dendritic" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in W Y Z W Y Z; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
