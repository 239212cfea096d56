mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/full/pytest.log
tail -4 gpurun_out/full/pytest.log
timeout 600 python bench.py --steps 10 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/full/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','max_abs_offset_error_px','pairs_failed','attempts_per_step','batches_per_step','value_host_resident_tiles')})
print({k:v['ms_per_launch'] for k,v in d['stages'].items()}); print(d['roofline']['frac'], d['roofline']['valu_issued_over_lower_bound'], d['cpu_baseline']['value'])
PY
