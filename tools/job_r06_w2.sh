#!/bin/bash
# round 6, last call: the ancillary lines on the FINAL build (line scan, dendritic25, decode inclusive, the dataset end to end with the colour mosaic,
# RCCL at world size 1, configs[4] on one GPU)
mkdir -p gpurun_out/r06w2
O=gpurun_out/r06w2
timeout 400 python bench.py --method surf_full --steps 10 --warmup 3 > $O/bench_surf_full.json 2> $O/bench_surf_full.err
timeout 300 python bench.py --workload dendritic25 --steps 10 --warmup 3 > $O/bench_dendritic25.json 2> $O/bench_dendritic25.err
timeout 300 python bench.py --from-files --steps 5 --warmup 1 > $O/bench_from_files_gray.json 2> $O/ffg.err
timeout 300 python bench.py --from-files --color --steps 5 --warmup 1 > $O/bench_from_files_color.json 2> $O/ffc.err
timeout 300 python tools/e2e_dataset.py > $O/e2e_dataset.json 2> $O/e2e.err
timeout 300 python bench.py --force-dist --steps 10 --warmup 3 --cpu-sample 0 --no-host-leg --no-cold-leg > $O/bench_force_dist.json 2> $O/fd.err
timeout 900 python bench.py --rows 32 --cols 32 --tile 4096 --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg --no-cold-leg --prior same --also-fuse > $O/bench_config4_surf.json 2> $O/bench_config4_surf.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06w2/*.json')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); print(f.split('/')[-1], d.get('metric', '')[:30], d.get('value'), d.get('unit'), d.get('ms_per_step'), d.get('seconds_per_dataset'))
PY
