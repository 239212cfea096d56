#!/bin/bash
# round 6, call B: hess_det form (C) against call A's build (B); per-kernel times and SQ / TA counters of the detect stage on the micro batch
mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 600 python -m pytest tests -m gpu -x -q -k "surf or dll or fused or full_size or config4 or dendritic or zirconcl or tie" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in B C B C; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
bash tools/kprof.sh r06b > $O/kprof.txt 2>&1; head -30 gpurun_out/kprof_r06b.csv
bash tools/pmc.sh r06b "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS_sum" > $O/pmc.txt 2>&1
grep -E "hessian|nms|orientation|bucket|describe|bf_" gpurun_out/pmc_r06b.txt | cut -c1-330
