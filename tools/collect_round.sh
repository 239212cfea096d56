#!/bin/bash
# tools/collect_round.sh <tag> <job dir>: copy what tools/job_*_s.sh brought back (gpurun_out/prof_<tag>, gpurun_out/<job dir>) into profiles/<tag>_*
set -e
T=$1; J=gpurun_out/$2; P=gpurun_out/prof_$T
cp $P/pmc_summary.txt profiles/${T}_pmc_summary.txt
for m in "" _orb _phase _fuse; do
  cp $P/kernel_stats$m.csv profiles/${T}_kernel_stats$m.csv
  n=${m#_}; cp $P/trace_bench$m.json profiles/${T}_bench${n:+_$n}_under_rocprof.json
done
for m in default orb phase fuse; do grep '^{' $J/bench_$m.json | tail -1 > profiles/${T}_bench_$m.json; done
python - "$T" <<'PY'
import json, sys
t = sys.argv[1]
d = json.loads(open("profiles/%s_bench_default.json" % t).read())
if d.get("projected_scaling"):
    json.dump(d["projected_scaling"], open("profiles/%s_projected_scaling.json" % t, "w"), indent=1)
PY
python tools/build_id.py | head -3; head -1 profiles/${T}_pmc_summary.txt
