#!/bin/bash
# round 4 final check: full GPU suite, smoke, default bench line (as the driver runs it)
mkdir -p gpurun_out/final
O=gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], d['ms_per_step'], d['attempts_per_step'], d['batches_per_step'], 'cold', d['value_cold_path'], 'host', d['value_host_resident_tiles'], 'err', d['max_abs_offset_error_px'], 'frac', d['roofline']['frac'], d['roofline']['valu_issued_over_lower_bound'], d['cpu_baseline']['value'])
PY
