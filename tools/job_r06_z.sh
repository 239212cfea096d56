#!/bin/bash
# per-dispatch durations of the fuse kernels of one mosaic (rocprofv3 --kernel-trace, csv), analytic on / off
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06z
cd /tmp && export TMPDIR=/tmp
for A in 1 0; do
  rm -rf $R/gpurun_out/r06z/prof
  VFSMS_FUSE_ANALYTIC=$A timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06z/prof -o fz -- python $R/bench.py --method fuse --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/r06z/fz_$A.json 2> $R/gpurun_out/r06z/fz_$A.err
  f=$(find $R/gpurun_out/r06z/prof -name "*kernel_trace.csv" | head -1)
  python - "$f" $A <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fuse" in r["Kernel_Name"] or "k_paste" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("analytic", sys.argv[2])
for k, v in by.items():
    v2 = sorted(v)
    print("  %-40s n=%d sum=%.0f us min=%.1f med=%.1f p90=%.1f max=%.1f" % (k[:40], len(v), sum(v), v2[0], v2[len(v2)//2], v2[int(len(v2)*0.9)], v2[-1]))
PY
done
rm -rf $R/gpurun_out/r06z/prof
