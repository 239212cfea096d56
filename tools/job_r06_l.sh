#!/bin/bash
# round 6, call L: configs[4]-size strips (4096^2 tiles, 37 k keypoints per 819 x 4096 ROI): BF train splits by train count (BA: 10240, BC: 5120)
# against the wave-count rule alone (BB); parity of the BF / config4 tests on the new default; then configs[4] itself with the projection
mkdir -p gpurun_out/r06l
O=gpurun_out/r06l
timeout 600 python -m pytest tests -m gpu -x -q -k "bf or config4 or fused or surf" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in BB BA BC BB BA BC; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 300 python tools/microbench.py 6 6 4096 2>&1 | tail -2
done | tee $O/ab_bf_split_4096.txt
for L in BB BA; do
    echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 200 python tools/microbench.py 16 40 2>&1 | tail -2
done | tee -a $O/ab_bf_split_4096.txt
timeout 1500 python bench.py --rows 32 --cols 32 --tile 4096 --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg --no-cold-leg --prior same --project-shards 8 > $O/bench_config4_surf.json 2> $O/bench_config4_surf.err; tail -c 1200 $O/bench_config4_surf.err
python - <<'PY'
import json
for l in open('gpurun_out/r06l/bench_config4_surf.json'):
    if l.startswith('{'):
        d=json.loads(l); print('config4', d['metric'], d['value'], d['ms_per_step'], d['max_abs_offset_error_px'], d['pairs_failed'], d['attempts_per_step'], d['batches_per_step'])
        print({k:v['ms_per_launch'] for k,v in d['stages'].items()})
        v=d['projected_scaling']['N=8']; print(v['projected_pairs_per_s'], v['projected_efficiency_vs_this_run_at_1'], [(r['attempts_per_step'], r['wall_ms']) for r in v['ranks']])
PY
