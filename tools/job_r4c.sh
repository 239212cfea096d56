#!/bin/bash
# round 4, third GPU call: hessian variants (parity + A/B), N > 1 rehearsal over gloo with / without the scan-pattern hint
mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
timeout 900 python -m pytest tests -m gpu -x -q -k "surf or keypoint or full_size or config4 or parameter_variants or dendritic or fused or edge" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in E1 H1 E1 H1; do
  echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.log
export VFSMS_DIST_BACKEND=gloo
for n in 8 2; do
  for h in "" "--no-path-hint"; do
    timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2960$n bench.py --gpus $n --steps 3 --warmup 1 --cpu-sample 0 --no-host-leg $h > $O/rehearsal_n${n}_${h:-hint}.json 2> $O/rehearsal_n${n}_${h:-hint}.err
    tail -c 1500 $O/rehearsal_n${n}_${h:-hint}.json | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n','$h',d['value'],d['ms_per_step'],d['attempts_per_step'],[ (r['pairs'],r['attempts_per_step'],r['batches_per_step'],r['gpu_ms_per_step']) for r in d['per_rank']], d['collective'].get('hint_repair_rounds'))
except Exception as e: print('ERR',e)
"
  done
done
