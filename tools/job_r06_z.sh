#!/bin/bash
mkdir -p gpurun_out/r06z
O=gpurun_out/r06z
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuse or mosaic or fused_from or canvas or driver or main_py or line_scan or hand_off or golden or random_placements or phase" 2>&1 | tail -4
for A in 1 1; do
  timeout 300 python bench.py --method fuse --steps 10 --warmup 3 --cpu-sample 0 > $O/bench_fuse_a$A.json 2> $O/bench_fuse_a$A.err
  python - $O/bench_fuse_a$A.json $A <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("fuse", d["value"], d["unit"], d["ms_per_step"], d["ms_per_step_with_stage_events"], d["roofline"]["frac"])
PY
done
