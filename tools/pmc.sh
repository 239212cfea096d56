#!/bin/bash
# PMC passes over the fixed micro batch (GPU box; through gpurun): each counter set in its own rocprofv3 run, kernel-trace only.
#   bash tools/pmc.sh TAG "SQ_WAVE_CYCLES SQ_BUSY_CYCLES ..." ["second set" ...]   -> gpurun_out/pmc_TAG.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; : > gpurun_out/pmc_$TAG.txt
n=0
for SET in "$@"; do
  n=$((n+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$n -o pmc --output-format csv -- python tools/microbench.py ${PMC_N:-16} 2 > $OUT/p$n.txt 2> $OUT/p$n.err
  python - "$OUT/p$n" >> gpurun_out/pmc_$TAG.txt <<'PY'
import csv, glob, collections, sys
d=sys.argv[1]
a=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
f=glob.glob(d+'/**/*counter_collection.csv', recursive=True)[0]
seen=set()
for row in csv.DictReader(open(f)):
    k=row['Kernel_Name'].split('(')[0]
    a[k][row['Counter_Name']]+=float(row['Counter_Value'])
    key=(row.get('Dispatch_Id'),k)
    if key not in seen: seen.add(key); cnt[k]+=1
for k,v in sorted(a.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES', kv[1].get('GRBM_GUI_ACTIVE',0))):
    if k.startswith('k_') or k.startswith('void k_'):
        print('%-34s n=%-4d '%(k[:34],cnt[k])+' '.join('%s=%.4g'%(c.replace('SQ_','').replace('_sum',''),x/cnt[k]) for c,x in sorted(v.items())))
PY
  rm -rf $OUT/p$n
done
cat gpurun_out/pmc_$TAG.txt | cut -c1-400
