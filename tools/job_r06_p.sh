#!/bin/bash
# round 6, call P: the round's bench lines on the final kernels (every method with cpu_baseline; projection; from-files; e2e dataset; line scan; dendritic25)
mkdir -p gpurun_out/r06p
O=gpurun_out/r06p
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --project-shards 2,4,8 > $O/bench_default.json 2> $O/bench_default.err
for M in orb phase fuse surf_full; do timeout 400 python bench.py --method $M --steps 10 --warmup 3 > $O/bench_$M.json 2> $O/bench_$M.err; done
timeout 300 python bench.py --workload dendritic25 --steps 10 --warmup 3 > $O/bench_dendritic25.json 2> $O/bench_dendritic25.err
timeout 300 python bench.py --from-files --steps 5 --warmup 1 > $O/bench_from_files_gray.json 2> $O/ffg.err
timeout 300 python bench.py --from-files --color --steps 5 --warmup 1 > $O/bench_from_files_color.json 2> $O/ffc.err
timeout 300 python tools/e2e_dataset.py > $O/e2e_dataset.json 2> $O/e2e.err
timeout 300 python bench.py --force-dist --steps 10 --warmup 3 --cpu-sample 0 --no-host-leg --no-cold-leg > $O/bench_force_dist.json 2> $O/fd.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06p/*.json')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); print(f.split('/')[-1], d.get('value'), d.get('unit'), d.get('ms_per_step'), (d.get('cpu_baseline') or {}).get('value'), d.get('seconds_per_dataset'))
PY
