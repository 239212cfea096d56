#!/bin/bash
# round 5, GPU call 8 (final sources): rocprofv3 evidence of the round (kernel trace + PMC passes, stamped with the build), then the default
# bench line reading it, then the plain lines of the other methods
mkdir -p gpurun_out/final2
O=gpurun_out/final2
bash tools/profile_round.sh r05 pmc > $O/profile_round.log 2>&1; tail -4 $O/profile_round.log | cut -c1-300
P=gpurun_out/prof_r05
cp $P/pmc_summary.txt profiles/r05_pmc_summary.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final2/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('default', d['value'], d['ms_per_step'], d['attempts_per_step'], d['batches_per_step'], 'cold', d['value_cold_path'], 'host', d['value_host_resident_tiles'], 'err', d['max_abs_offset_error_px'], d['pairs_failed'])
print('roofline frac', r['frac'], 'at eff clock', r.get('frac_at_effective_clock'), r.get('effective_clock_ghz'), 'issued/lower', r['valu_issued_over_lower_bound'], 'traffic', r['traffic'], 'compulsory', r['compulsory_bytes_per_launch'], 'stale', r.get('pmc_stale'), 'cpu', d['cpu_baseline']['value'])
print({k: round(v['ms']/d['steps'],2) for k,v in d['stages'].items()})
PY
for m in orb phase fuse surf_full; do timeout 200 python bench.py --method $m --cpu-sample 0 > $O/bench_$m.json 2> $O/bench_$m.err; python -c "
import json
for l in open('$O/bench_$m.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$m', d['value'], d['unit'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done
timeout 200 python bench.py --workload dendritic25 --cpu-sample 0 > $O/bench_dendritic25.json 2> $O/bench_d25.err; python -c "
import json
for l in open('$O/bench_dendritic25.json'):
    if l.startswith('{'):
        d=json.loads(l); print('dendritic25', d['value'], d['ms_per_step'], d['attempts_per_step'], d['batches_per_step'])"
