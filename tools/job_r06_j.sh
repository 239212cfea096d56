#!/bin/bash
mkdir -p gpurun_out/r06j
O=gpurun_out/r06j
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in N W X N W X; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
timeout 600 python bench.py --steps 10 --warmup 3 --project-shards 8 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
for l in open('gpurun_out/r06j/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); print('surf', d['value'], d['ms_per_step'], 'cold', d['value_cold_path'], 'host', d['value_host_resident_tiles'])
        print({k:v['ms_per_launch'] for k,v in d['stages'].items()})
        v=d['projected_scaling']['N=8']; print(v['projected_pairs_per_s'], v['projected_efficiency_vs_this_run_at_1'])
PY
