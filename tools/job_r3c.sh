mkdir -p gpurun_out/r3c
timeout 600 python -m pytest tests -m gpu -x -q -k "surf or dll or fused or full_size or config4 or dendritic or resident" > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log
tail -15 gpurun_out/r3c/pytest.log
for v in 1 0 1 0; do echo "== LEGACY=$v"; VFSMS_DESC_LEGACY=$v timeout 200 python tools/microbench.py 16 50 2>&1 | tail -2; done | tee gpurun_out/r3c/micro.log
VFSMS_DESC_LEGACY=0 bash tools/kprof.sh v3 > gpurun_out/r3c/kprof_v3.txt 2>&1
grep -i "describe\|desc_\|pair_rows" gpurun_out/kprof_v3.csv
VFSMS_DESC_LEGACY=0 bash tools/pmc.sh v3 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" > /dev/null 2>&1
grep -i "describe" gpurun_out/pmc_v3.txt | cut -c1-400
