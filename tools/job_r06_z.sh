#!/bin/bash
# configs[4] on one GPU with smaller speculation windows (batch working set against the 256 MB infinity cache)
mkdir -p gpurun_out/r06z
O=gpurun_out/r06z
for W in 16 24; do
  timeout 700 python bench.py --rows 32 --cols 32 --tile 4096 --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg --no-cold-leg --prior same --window $W > $O/bench_config4_w$W.json 2> $O/bench_config4_w$W.err
  tail -1 $O/bench_config4_w$W.err
  python - $O/bench_config4_w$W.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d.get("attempts_per_step"), d.get("batches_per_step"), {k: v["ms_per_launch"] for k, v in d["stages"].items() if k in ("describe", "bf_mfma", "hessian", "orientation", "bf_verify")})
PY
done
