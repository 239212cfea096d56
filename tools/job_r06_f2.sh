#!/bin/bash
# round 6, last rehearsal of the N-rank bench on the one-GPU box (gloo) with the final bench.py
mkdir -p gpurun_out/r06f2
O=gpurun_out/r06f2
export VFSMS_DIST_BACKEND=gloo
for N in 2 8; do
    HL=""; [ $N = 8 ] && HL="--no-host-leg"
    timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2954$N bench.py --gpus $N --steps 5 --warmup 1 --cpu-sample 0 $HL > $O/rehearsal_gloo_n$N.json 2> $O/rehearsal_n$N.err
    echo "N=$N rc=$?"
    python - $N <<'PY'
import json,sys
N=sys.argv[1]
for l in open('gpurun_out/r06f2/rehearsal_gloo_n%s.json'%N):
    if l.startswith('{'):
        d=json.loads(l); print('N', N, d['value'], d['ms_per_step'], d['max_abs_offset_error_px'], d['pairs_failed'], d['n_gpus'], d['scaling'])
PY
done
