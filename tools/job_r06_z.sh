#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06z
cd /tmp && export TMPDIR=/tmp
for v in nopre pre128 pre153; do
  lib=$R/build_ab/$v.so
  rm -rf $R/gpurun_out/r06z/prof
  VFSMS_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06z/prof -o ph -- python $R/tools/phase_ab.py 32 10 t > $R/gpurun_out/r06z/ab_$v.log 2>&1
  grep "LDS transforms" $R/gpurun_out/r06z/ab_$v.log | cut -c1-100
  db=$(ls $R/gpurun_out/r06z/prof/*results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db $R/gpurun_out/r06z/ks_$v.csv > /dev/null && echo "$v:" && grep "k_phase\|k_peak" $R/gpurun_out/r06z/ks_$v.csv | awk -F, '{print "   ", $1, $(NF-1)}' | cut -c1-60
done
rm -rf $R/gpurun_out/r06z/prof
cd $R; for v in pre128 pre153; do VFSMS_LIB=$R/build_ab/$v.so timeout 300 python tools/phase_ab.py 4 2 2>&1 | grep "worst"; done
