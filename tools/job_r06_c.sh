#!/bin/bash
# round 6, call C: whole GPU suite on the new detect stage; the default bench with the projected 2 / 4 / 8-rank steps; the other method lines with cpu_baseline
mkdir -p gpurun_out/r06c
O=gpurun_out/r06c
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --project-shards 2,4,8 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.err
python - <<'PY'
import json
for l in open('gpurun_out/r06c/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); print('surf', d['value'], d['ms_per_step'], 'cold', d['value_cold_path'], 'host', d['value_host_resident_tiles'])
        print({k:v['ms_per_launch'] for k,v in d['stages'].items()})
        for k,v in (d.get('projected_scaling') or {}).items():
            if k!='note': print(k, v['projected_pairs_per_s'], v['projected_efficiency_vs_this_run_at_1'], 'slowest', v['slowest_rank_ms'], 'tail', v['tail_ms_gather_assemble_learn'], [(r['attempts_per_step'], r['batches_per_step'], r['wall_ms'], r['gpu_ms']) for r in v['ranks']])
PY
for M in orb phase fuse; do
  timeout 400 python bench.py --method $M --steps 10 --warmup 3 > $O/bench_$M.json 2> $O/bench_$M.err; tail -c 400 $O/bench_$M.err
  python - $M <<'PY'
import json,sys
for l in open('gpurun_out/r06c/bench_%s.json'%sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], d['value'], d['unit'], d['ms_per_step'], 'cpu', d.get('cpu_baseline'), d.get('pairs_off_truth_note'))
PY
done
