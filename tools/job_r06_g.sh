#!/bin/bash
mkdir -p gpurun_out/r06g
O=gpurun_out/r06g
for L in I J K L I J K L; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
