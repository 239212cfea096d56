#!/bin/bash
# round 5, GPU call 3: what bounds k_describe (PMC, round-4 staging vs balanced staging); BF || detect on two streams (A/B on the bench); new tests
mkdir -p gpurun_out/r05c
O=gpurun_out/r05c
bash tools/pmc_ab.sh old build_ab/S_old.so > $O/pmc_old.log 2>&1
bash tools/pmc_ab.sh big build_ab/S_big.so > $O/pmc_big.log 2>&1
cat gpurun_out/pmcab_old.txt gpurun_out/pmcab_big.txt | cut -c1-420
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mode_vote or registrar or dendritic_path or incremental or driver or fused or full_width" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for ov in 0 1 0 1; do
  VFSMS_OVERLAP=$ov timeout 200 python bench.py --cpu-sample 0 --no-host-leg --no-cold-leg --prior same --steps 10 --warmup 2 > $O/bench_ov$ov.json 2> $O/bench_ov$ov.err
  python - $O/bench_ov$ov.json $ov <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); print('overlap', sys.argv[2], d['value'], d['ms_per_step'], 'stages sum', d.get('stages_sum_ms_per_step'), 's2', d.get('stages_second_stream_ms_per_step'), 'err', d['max_abs_offset_error_px'], d['pairs_failed'])
        print('   ', {k: round(v['ms']/d['steps'],2) for k,v in d['stages'].items()})
PY
done
for pct in 50 85; do
  VFSMS_OVERLAP_PCT=$pct timeout 200 python bench.py --cpu-sample 0 --no-host-leg --no-cold-leg --prior same --steps 10 --warmup 2 2> /dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('overlap pct $pct', d['value'], d['ms_per_step'])"
done
