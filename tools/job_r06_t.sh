#!/bin/bash
mkdir -p gpurun_out/r06t
for L in GL R12 R16 GL R12 R16; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee gpurun_out/r06t/ab.txt
