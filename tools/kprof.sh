#!/bin/bash
# Quick per-kernel timing of the fixed 16-pair micro batch under rocprofv3 (GPU box; run through gpurun).
#   bash tools/kprof.sh TAG [N] [K]      -> gpurun_out/kprof_TAG.csv (+ printed)
TAG=${1:-x}; N=${2:-16}; K=${3:-3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kprof_$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/microbench.py $N $K > $OUT/micro.txt 2> $OUT/trace.err
cat $OUT/micro.txt
python tools/rocpd_summary.py $(ls $OUT/trace/*results.db | head -1) gpurun_out/kprof_$TAG.csv > /dev/null
cut -c1-150 gpurun_out/kprof_$TAG.csv | head -40
rm -rf $OUT/trace
