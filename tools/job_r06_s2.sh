#!/bin/bash
# round 6, final call: job S (GPU suite, profiles stamped with this build, default / orb / phase / fuse lines) + the mosaic PMC section + configs[4]
bash tools/job_r06_s.sh
timeout 400 python bench.py --method fuse --steps 10 --warmup 3 > gpurun_out/r06s/bench_fuse.json 2> gpurun_out/r06s/bench_fuse.err     # again, now with the mosaic traffic of this build in the summary
mkdir -p gpurun_out/r06w2
timeout 900 python bench.py --rows 32 --cols 32 --tile 4096 --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg --no-cold-leg --prior same --also-fuse > gpurun_out/r06w2/bench_config4_surf.json 2> gpurun_out/r06w2/bench_config4_surf.err
python - <<'PY'
import json
for l in open("gpurun_out/r06w2/bench_config4_surf.json"):
    if l.startswith("{"):
        d = json.loads(l); print("config4", d["metric"][:28], d["value"], d["ms_per_step"], d.get("ms_per_step_with_stage_events"), (d.get("roofline") or {}).get("frac"))
d = json.loads([l for l in open("gpurun_out/r06s/bench_fuse.json") if l.startswith("{")][-1]); r = d["roofline"]
print("fuse", d["value"], d["ms_per_step"], r["frac"], r.get("traffic_over_algorithmic"), r.get("pmc_stale"))
PY
