# kernel trace + PMC of the descriptor kernels, legacy vs row x cell (fixed 16-pair micro batch)
mkdir -p gpurun_out/r3b
VFSMS_DESC_LEGACY=0 bash tools/kprof.sh rc > gpurun_out/r3b/kprof_rc.txt 2>&1
VFSMS_DESC_LEGACY=1 bash tools/kprof.sh legacy > gpurun_out/r3b/kprof_legacy.txt 2>&1
grep -i "describe\|desc_\|pair_rows" gpurun_out/kprof_rc.csv gpurun_out/kprof_legacy.csv
VFSMS_DESC_LEGACY=0 bash tools/pmc.sh rc "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" > /dev/null 2>&1
grep -i "describe" gpurun_out/pmc_rc.txt | cut -c1-400
