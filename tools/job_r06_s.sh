#!/bin/bash
# round 6, call S: the round's evidence on the FINAL kernels: GPU suite, rocprofv3 kernel trace + PMC passes (stamped), then the default line with
# the projection, the ORB / phase / fuse lines, configs[4]
mkdir -p gpurun_out/r06s
O=gpurun_out/r06s
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
bash tools/profile_round.sh r06 pmc > gpurun_out/prof_r06_stdout.txt 2>&1
cp gpurun_out/prof_r06/pmc_summary.txt profiles/r06_pmc_summary.txt     # bench.py reads the newest committed summary: make it this build's before the final line
timeout 600 python bench.py --steps 20 --warmup 5 --project-shards 2,4,8 > $O/bench_default.json 2> $O/bench_default.err
for M in orb phase fuse; do timeout 400 python bench.py --method $M --steps 10 --warmup 3 > $O/bench_$M.json 2> $O/bench_$M.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06s/*.json')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); r=d.get('roofline') or {}
            print(f.split('/')[-1], d.get('value'), d.get('ms_per_step'), r.get('frac'), r.get('pmc_stale'), (d.get('cpu_baseline') or {}).get('value'))
            if 'projected_scaling' in d and d['projected_scaling']:
                print({k:(v['projected_pairs_per_s'], v['projected_efficiency_vs_this_run_at_1']) for k,v in d['projected_scaling'].items() if k!='note'})
                print({k:v['ms_per_launch'] for k,v in d['stages'].items()})
PY
