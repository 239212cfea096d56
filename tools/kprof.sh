#!/bin/bash
# Quick per-kernel timing under rocprofv3 (GPU box; run through gpurun).
#   bash tools/kprof.sh TAG [cmd...]      default cmd: python tools/microbench.py 16 3     -> gpurun_out/kprof_TAG.csv (+ printed)
TAG=${1:-x}; shift
CMD=${@:-python tools/microbench.py 16 3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kprof_$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/stdout.txt 2> $OUT/trace.err
tail -3 $OUT/stdout.txt | cut -c1-600
python tools/rocpd_summary.py $(ls $OUT/trace/*results.db | head -1) gpurun_out/kprof_$TAG.csv > /dev/null
cut -c1-150 gpurun_out/kprof_$TAG.csv | head -40
rm -rf $OUT/trace
