#!/bin/bash
# round 6: per-kernel times of the LDS-transform phase path (tools/phase_ab.py, timing part only) under rocprofv3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r06x
rm -rf $R/gpurun_out/r06x/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06x/prof -o ph -- python $R/tools/phase_ab.py 32 20 t > $R/gpurun_out/r06x/ab.log 2>&1
grep "LDS transforms" $R/gpurun_out/r06x/ab.log
db=$(ls $R/gpurun_out/r06x/prof/*results.db 2>/dev/null | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db $R/gpurun_out/r06x/kernel_stats_phase_ab.csv && head -8 $R/gpurun_out/r06x/kernel_stats_phase_ab.csv
rm -rf $R/gpurun_out/r06x/prof
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $R/gpurun_out/r06x/pmc -o pmc --output-format csv -- python $R/tools/phase_ab.py 8 3 t > $R/gpurun_out/r06x/pmc.log 2>&1
f=$(ls $R/gpurun_out/r06x/pmc/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in acc.items():
    if "phase" in k or "peak" in k:
        print(k, " ".join("%s=%.3g" % (c, v) for c, v in sorted(d.items())))
PY
rm -rf $R/gpurun_out/r06x/pmc
