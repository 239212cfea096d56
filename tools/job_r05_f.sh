#!/bin/bash
# round 5, GPU call 6: the whole GPU suite, smoke, the round's rocprofv3 evidence (kernel trace + PMC passes), then the default bench line
# reading THAT evidence (pmc_stale must come out false)
mkdir -p gpurun_out/final
O=gpurun_out/final
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_round.sh r05 pmc > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log | cut -c1-300
P=gpurun_out/prof_r05
cp $P/pmc_summary.txt profiles/r05_pmc_summary.txt
cp $P/kernel_stats.csv profiles/r05_kernel_stats.csv
for m in orb phase fuse; do cp $P/kernel_stats_$m.csv profiles/r05_kernel_stats_$m.csv; cp $P/trace_bench_$m.json profiles/r05_bench_${m}_under_rocprof.json; done
cp $P/trace_bench.json profiles/r05_bench_under_rocprof.json
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('default', d['value'], d['ms_per_step'], d['attempts_per_step'], d['batches_per_step'], 'cold', d['value_cold_path'], 'host', d['value_host_resident_tiles'], 'err', d['max_abs_offset_error_px'], d['pairs_failed'])
print('roofline frac', r['frac'], 'at eff clock', r.get('frac_at_effective_clock'), r.get('effective_clock_ghz'), 'issued/lower', r['valu_issued_over_lower_bound'], 'traffic', r['traffic'], 'compulsory', r['compulsory_bytes_per_launch'], 'stale', r.get('pmc_stale'), 'cpu', d['cpu_baseline']['value'])
print({k: round(v['ms']/d['steps'],2) for k,v in d['stages'].items()})
PY
