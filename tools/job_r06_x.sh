#!/bin/bash
# round 6: SQ counters of the LDS-transform phase kernels (tools/phase_ab.py, timing part only)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r06x
summ() {
python - "$1" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, d in acc.items():
    if "phase" in k:
        print(k, "n=%d" % len(n[k]), " ".join("%s=%.4g" % (c, v / len(n[k])) for c, v in sorted(d.items())))
PY
}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  rm -rf $R/gpurun_out/r06x/pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/r06x/pmc -o pmc --output-format csv -- python $R/tools/phase_ab.py 32 3 t > $R/gpurun_out/r06x/pmc.log 2>&1
  f=$(ls $R/gpurun_out/r06x/pmc/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && summ $f || tail -3 $R/gpurun_out/r06x/pmc.log
done
rm -rf $R/gpurun_out/r06x/pmc
