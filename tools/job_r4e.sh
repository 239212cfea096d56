#!/bin/bash
# round 4, fifth GPU call: shape-run launches (parity + bench), decode pool with the Pillow block cache at 16 / 24 / 32 threads
mkdir -p gpurun_out/r4e
O=gpurun_out/r4e
timeout 1200 python -m pytest tests -m gpu -x -q -k "not orb and not phase and not fuse" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 400 python bench.py --steps 10 --warmup 2 --cpu-sample 0 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4e/bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], d['ms_per_step'], d['attempts_per_step'], d['batches_per_step'], 'cold', d['value_cold_path'], 'host', d['value_host_resident_tiles'], 'err', d['max_abs_offset_error_px'])
print({k:v['ms_per_launch'] for k,v in d['stages'].items()})
PY
for t in 16 24 32; do for c in "" "--color"; do
  timeout 300 python bench.py --from-files $c --decode-threads $t --steps 5 > $O/ff${c}_$t.json 2> $O/ff${c}_$t.err
  python - "$O/ff${c}_$t.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["decode_only_ms_per_step"], d["registration_only_ms_per_step"], d["end_to_end_over_slower_stage"], d.get("ingest_thread_ms_per_tile"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done; done
