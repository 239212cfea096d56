#!/bin/bash
# round 5, GPU call 5: descriptor work list (records) and the asynchronous next-keypoint prefetch: parity, A B A B, bench
mkdir -p gpurun_out/r05e
O=gpurun_out/r05e
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "surf or dll or full_width or fused or config4_tile or zircon or edge or parameter or keypoint_greater or featureless" > $O/pytest_surf.log 2>&1; tail -4 $O/pytest_surf.log
for L in X_new R1 R2 X_new R1 R2; do
  echo "== $L" >> $O/ab.txt; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py 16 60 2>&1 | tail -2 >> $O/ab.txt
done
cat $O/ab.txt | cut -c1-330
for L in X_new R1 R2 X_new R1 R2; do
  echo "== $L (40 pairs)" >> $O/ab40.txt; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py 40 30 2>&1 | tail -2 >> $O/ab40.txt
done
cat $O/ab40.txt | cut -c1-330
timeout 200 python bench.py --cpu-sample 0 --no-host-leg --no-cold-leg --prior same --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); print('bench', d['value'], d['ms_per_step'], 'err', d['max_abs_offset_error_px'], d['pairs_failed'])
        print('   ', {k: round(v['ms']/d['steps'],2) for k,v in d['stages'].items()})
PY
tail -3 $O/bench.err
