#!/bin/bash
mkdir -p gpurun_out/r06h
O=gpurun_out/r06h
for L in L M N O P L M N O P; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.txt
timeout 300 python bench.py --method fuse --steps 10 --warmup 3 > $O/bench_fuse.json 2> $O/bench_fuse.err
python - <<'PY'
import json
for l in open('gpurun_out/r06h/bench_fuse.json'):
    if l.startswith('{'):
        d=json.loads(l); print('fuse', d['value'], d['ms_per_step'], d['ms_per_step_with_stage_events'], d['roofline']['frac'], d['cpu_baseline']['value'])
PY
