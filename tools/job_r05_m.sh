#!/bin/bash
mkdir -p gpurun_out/final7
O=gpurun_out/final7
timeout 200 python bench.py --force-dist --cpu-sample 0 --no-host-leg --no-cold-leg > $O/bench_force_dist.json 2> $O/fd.err; python -c "
import json
for l in open('$O/bench_force_dist.json'):
    if l.startswith('{'):
        d=json.loads(l); print('force-dist', d['value'], d['ms_per_step'], d['collective']['backend'], d['collective']['world_size'])"
timeout 200 python bench.py --from-files --steps 5 --warmup 1 > $O/bench_from_files_gray.json 2> $O/ffg.err; python -c "
import json
for l in open('$O/bench_from_files_gray.json'):
    if l.startswith('{'):
        d=json.loads(l); print('from-files gray', d['value'], d['ms_per_step'], 'decode only', d['decode_only_tiles_per_s'], 'reg only', d['registration_only_pairs_per_s'])"
