#!/bin/bash
# round 5, last GPU minutes: units BESIDE the image sampled by the nearest-pixel path (R4) against the final kernels (R3).  Adopted only if
# the parity subset passes AND the describe stage gains >= 1 %: then the round's evidence is taken again on R4 (profile + bench); otherwise
# the gloo rehearsals of the sharded bench are refreshed on R3.
mkdir -p gpurun_out/final5
O=gpurun_out/final5
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "surf or dll or full_width or fused or config4_tile or zircon or edge or parameter or work_list" > $O/pytest_surf.log 2>&1; echo "rc=$?" >> $O/pytest_surf.log; tail -3 $O/pytest_surf.log
for L in R3 R4 R3 R4; do
  echo "== $L" >> $O/ab.txt; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py 40 30 2>&1 | tail -2 >> $O/ab.txt
done
cat $O/ab.txt | cut -c1-330
DEC=$(python - <<'PY'
import re
t=open('gpurun_out/final5/ab.txt').read().split('== ')[1:]
d={}
for blk in t:
    name=blk.split()[0]; m=re.search(r'describe=([0-9.]+)', blk)
    if m: d.setdefault(name,[]).append(float(m.group(1)))
ok='rc=0' in open('gpurun_out/final5/pytest_surf.log').read()
r3=sum(d['R3'])/len(d['R3']); r4=sum(d['R4'])/len(d['R4'])
print('ADOPT' if ok and r4 < 0.99*r3 else 'KEEP', r3, r4)
PY
)
echo "decision: $DEC" | tee $O/decision.txt
if [[ "$DEC" == ADOPT* ]]; then
  bash tools/profile_round.sh r05 pmc > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log | cut -c1-200
  cp gpurun_out/prof_r05/pmc_summary.txt profiles/r05_pmc_summary.txt
  timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
  python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default', d['value'], d['ms_per_step'], 'cold', d['value_cold_path'], 'host', d['value_host_resident_tiles'], 'err', d['max_abs_offset_error_px'], d['pairs_failed'], 'frac', r['frac'], r['valu_issued_over_lower_bound'], 'stale', r['pmc_stale'])"
else
  export VFSMS_LIB=build_ab/R3.so VFSMS_DIST_BACKEND=gloo
  for N in 2 8; do
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 5 --warmup 1 --cpu-sample 0 --no-host-leg > $O/rehearsal_gloo_n$N.json 2> $O/rehearsal_n$N.err
    python -c "
import json
for l in open('$O/rehearsal_gloo_n$N.json'):
    if l.startswith('{'):
        d=json.loads(l); print('gloo N=$N', d['value'], d['ms_per_step'], [(r['attempts_per_step'], r['batches_per_step']) for r in d['per_rank']], d['collective']['prediction_repair_rounds'])"
  done
fi
