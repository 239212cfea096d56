#!/bin/bash
mkdir -p gpurun_out/r06i
O=gpurun_out/r06i
for L in N T Q U R N T Q U R; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
