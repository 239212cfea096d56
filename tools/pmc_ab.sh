#!/bin/bash
# PMC passes over the fixed micro batch for ONE build of the library (GPU box; through gpurun): counter sets filtered by what the agent offers,
# each set in its own rocprofv3 run, kernel-trace only.   bash tools/pmc_ab.sh TAG LIB [kernel-regex]   -> gpurun_out/pmcab_TAG.txt
TAG=$1; LIB=$2; KRE=${3:-k_describe}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcab_$TAG; mkdir -p $OUT; : > gpurun_out/pmcab_$TAG.txt
[ -f gpurun_out/pmc_avail.txt ] || timeout 120 rocprofv3-avail list > gpurun_out/pmc_avail.txt 2>&1
SETS=(
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
 "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_BUSY_CYCLES"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
)
n=0
for SET in "${SETS[@]}"; do
  n=$((n+1)); KEEP=""
  for c in $SET; do
    b=${c%_sum}
    if grep -q -w "$c" gpurun_out/pmc_avail.txt || grep -q -w "$b" gpurun_out/pmc_avail.txt; then KEEP="$KEEP $c"; else echo "# not offered: $c" >> gpurun_out/pmcab_$TAG.txt; fi
  done
  [ -n "$KEEP" ] || continue
  VFSMS_LIB=$LIB timeout 200 rocprofv3 --kernel-trace --pmc $KEEP -d $OUT/p$n -o pmc --output-format csv -- python tools/microbench.py ${PMC_N:-16} 2 > $OUT/p$n.txt 2> $OUT/p$n.err || { echo "# pass $n failed: $(tail -2 $OUT/p$n.err | cut -c1-200)" >> gpurun_out/pmcab_$TAG.txt; }
  python - "$OUT/p$n" "$KRE" >> gpurun_out/pmcab_$TAG.txt <<'PY'
import csv, glob, collections, sys, re
d, kre = sys.argv[1], re.compile(sys.argv[2])
a=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.defaultdict(float)
fs=glob.glob(d+'/**/*counter_collection.csv', recursive=True)
if not fs: sys.exit(0)
seen=set()
for row in csv.DictReader(open(fs[0])):
    k=row['Kernel_Name'].split('(')[0]
    if not kre.search(k): continue
    a[k][row['Counter_Name']]+=float(row['Counter_Value'])
    key=(row.get('Dispatch_Id'),k)
    if key not in seen:
        seen.add(key); cnt[k]+=1
        try: dur[k]+=float(row['End_Timestamp'])-float(row['Start_Timestamp'])
        except Exception: pass
for k,v in sorted(a.items()):
    print('%-20s n=%-3d dur_ms=%.4f '%(k[:20],cnt[k],dur[k]/max(cnt[k],1)/1e6)+' '.join('%s=%.5g'%(c.replace('SQ_','').replace('_sum',''),x/cnt[k]) for c,x in sorted(v.items())))
PY
  rm -rf $OUT/p$n
done
cat gpurun_out/pmcab_$TAG.txt | cut -c1-500
