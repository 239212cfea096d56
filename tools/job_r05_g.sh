#!/bin/bash
# round 5, GPU call 7: R3 (one-run small staging + hoisted INTER_AREA tail weights) against R1; configs[4] surf + fuse on one GPU;
# one dataset end to end; decode-inclusive colour ingest
mkdir -p gpurun_out/r05g
O=gpurun_out/r05g
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "surf or dll or full_width or fused or config4_tile or zircon or edge or parameter" > $O/pytest_surf.log 2>&1; tail -3 $O/pytest_surf.log
for L in R1 R3 R1 R3; do
  echo "== $L" >> $O/ab.txt; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py 40 30 2>&1 | tail -2 >> $O/ab.txt
done
cat $O/ab.txt | cut -c1-330
timeout 300 python tools/e2e_dataset.py > $O/e2e_dataset.json 2> $O/e2e.err; cut -c1-1800 $O/e2e_dataset.json; tail -3 $O/e2e.err
timeout 200 python bench.py --from-files --color --steps 5 --warmup 1 > $O/bench_from_files_color.json 2> $O/ff.err; python -c "
import json
for l in open('$O/bench_from_files_color.json'):
    if l.startswith('{'):
        d=json.loads(l); print('from-files colour', d['value'], d['ms_per_step'], 'decode only', d['decode_only_tiles_per_s'], 'reg only', d['registration_only_pairs_per_s'])"
timeout 560 python bench.py --rows 32 --cols 32 --tile 4096 --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg --no-cold-leg --prior same --also-fuse > $O/bench_config4.json 2> $O/bench_config4.err; tail -c 900 $O/bench_config4.err
python - $O/bench_config4.json <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); print(d['metric'][:50], d['value'], d['unit'], d['ms_per_step'], 'err', d.get('max_abs_offset_error_px'), d.get('pairs_failed'))
        print('   ', {k: round(v['ms']/d['steps'],1) for k,v in d['stages'].items()})
PY
