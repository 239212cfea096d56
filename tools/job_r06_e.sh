#!/bin/bash
mkdir -p gpurun_out/r06e
O=gpurun_out/r06e
timeout 900 python -m pytest tests -m gpu -x -q -k "bf or surf or fused or config4 or dendritic or zirconcl" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in E H E H; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
