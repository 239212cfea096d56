#!/bin/bash
# round 6, call V: the mosaic walk's PMC traffic appended to the committed summary of THIS build (no kernel changed since), then the fuse line
mkdir -p gpurun_out/prof_r06v gpurun_out/r06v
python - <<'PY'
import sys
sys.path.insert(0, ".")
from tools.build_id import build_id
mine = build_id()["src_sha256"]
head = open("profiles/r06_pmc_summary.txt").readline()
assert ("src_sha256=" + mine) in head, (mine, head)
print("summary is of this build:", mine)
PY
[ $? -eq 0 ] || exit 1
cp profiles/r06_pmc_summary.txt gpurun_out/prof_r06v/pmc_summary.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/profile_mosaic_pmc.sh gpurun_out/prof_r06v | tail -8
cp gpurun_out/prof_r06v/pmc_summary.txt profiles/r06_pmc_summary.txt
timeout 400 python bench.py --method fuse --steps 10 --warmup 3 > gpurun_out/r06v/bench_fuse.json 2> gpurun_out/r06v/bench_fuse.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06v/bench_fuse.json") if l.startswith("{")][-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r.get("traffic"), r.get("traffic_over_algorithmic"), r.get("traffic_source"), r.get("pmc_stale"))
PY
