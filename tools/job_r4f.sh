#!/bin/bash
# round 4, sixth GPU call: ORB shape runs / level plan (parity + bench), phase and fuse benches, full GPU suite
mkdir -p gpurun_out/r4f
O=gpurun_out/r4f
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for m in orb phase fuse surf_full; do
  timeout 300 python bench.py --method $m --steps 5 --warmup 2 --cpu-sample 0 > $O/bench_$m.json 2> $O/bench_$m.err
  python - "$O/bench_$m.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["unit"], d["ms_per_step"], d.get("attempts_per_step"), d.get("batches_per_step"), d.get("value_cold_path"), {k:v.get('ms_per_launch') for k,v in (d.get('stages') or {}).items()})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
timeout 300 python bench.py --from-files --decode-threads 16 --steps 5 > $O/ff.json 2> $O/ff.err
timeout 300 python bench.py --from-files --color --decode-threads 16 --steps 5 > $O/ff_color.json 2> $O/ff_color.err
for f in $O/ff.json $O/ff_color.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["decode_only_ms_per_step"], d["registration_only_ms_per_step"], d["end_to_end_over_slower_stage"], d.get("ingest_thread_ms_per_tile"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
