#!/bin/bash
# round 6, call R: BF train tiles by LDS-DMA (GL) against register staging (NOGL): parity of the BF / fused tests, then A B A B at 2048 and 4096
mkdir -p gpurun_out/r06r
O=gpurun_out/r06r
timeout 900 python -m pytest tests -m gpu -x -q -k "bf or fused or surf or config4 or dendritic or zirconcl or dll" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in NOGL GL NOGL GL; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
for L in NOGL GL NOGL GL; do
    echo "== $L 4096"; VFSMS_LIB=build_ab/$L.so timeout 300 python tools/microbench.py 6 6 4096 2>&1 | tail -2
done | tee -a $O/ab.txt
