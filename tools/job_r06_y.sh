#!/bin/bash
# round 6: LDS-transform phase path -- parity sweep + timing under a few workgroup shapes + per-kernel times
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06y
timeout 600 python tools/phase_ab.py 32 20 2>&1 | grep -v "^W2026" | tail -7
for cfg in "4 4" "8 2" "8 8"; do
  set -- $cfg
  echo "== TDIV=$1 C=$2"
  VFSMS_PHASE_TDIV=$1 VFSMS_PHASE_C=$2 timeout 300 python tools/phase_ab.py 32 20 t 2>&1 | grep "LDS transforms"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r06y/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06y/prof -o ph -- python $R/tools/phase_ab.py 32 20 t > $R/gpurun_out/r06y/ab.log 2>&1
db=$(ls $R/gpurun_out/r06y/prof/*results.db 2>/dev/null | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db $R/gpurun_out/r06y/kernel_stats_phase_ab.csv && head -7 $R/gpurun_out/r06y/kernel_stats_phase_ab.csv
rm -rf $R/gpurun_out/r06y/prof
cd $R
[ "$1" = "notest" ] || timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phase" 2>&1 | tail -5
