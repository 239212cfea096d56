#!/bin/bash
# round 4, fourth GPU call: path memory -- default bench (hot + cold legs), from-files, N > 1 rehearsals over gloo, grid-related GPU tests
mkdir -p gpurun_out/r4d
O=gpurun_out/r4d
timeout 900 python -m pytest tests -m gpu -x -q -k "grid or dendritic or ingest or main_py or colour_mode or incremental or line_scan or real" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 400 python bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4d/bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], d['ms_per_step'], d['attempts_per_step'], d['batches_per_step'], 'cold', d['value_cold_path'], d['cold_path'], 'host', d['value_host_resident_tiles'], 'err', d['max_abs_offset_error_px'])
print({k:v['ms_per_launch'] for k,v in d['stages'].items()})
PY
for c in "" "--color"; do
  timeout 300 python bench.py --from-files $c --decode-threads 16 --steps 5 > $O/ff${c}.json 2> $O/ff${c}.err
  python - "$O/ff${c}.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["decode_only_ms_per_step"], d["registration_only_ms_per_step"], d["end_to_end_over_slower_stage"], d.get("ingest_thread_ms_per_tile"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
export VFSMS_DIST_BACKEND=gloo
for n in 8 2; do
  for h in "" "--no-path-memory"; do
    timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 3 --warmup 1 --cpu-sample 0 --no-host-leg $h > $O/rehearsal_n${n}${h}.json 2> $O/rehearsal_n${n}${h}.err
    python - "$O/rehearsal_n${n}${h}.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'],d['ms_per_step'],d['attempts_per_step'],[(r['pairs'],r['attempts_per_step'],r['batches_per_step'],r['gpu_ms_per_step']) for r in d['per_rank']], d['collective'].get('prediction_repair_rounds'), d['max_abs_offset_error_px'])
except Exception as e: print('ERR',e)
PY
  done
done
