#!/bin/bash
# round 6, call F: the sharded bench with SHARD-LOCAL tiles rehearsed on the one-GPU box (gloo: ranks share the device; what is checked is the code
# path -- per-rank tile sets, re-fetch when the learned split moves, host leg -- and the work split), and --force-dist with the projection
mkdir -p gpurun_out/r06f
O=gpurun_out/r06f
export VFSMS_DIST_BACKEND=gloo
for N in 2 8; do
    HL=""; [ $N = 8 ] && HL="--no-host-leg"
    timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2953$N bench.py --gpus $N --steps 5 --warmup 1 --cpu-sample 0 $HL > $O/rehearsal_gloo_n$N.json 2> $O/rehearsal_n$N.err
    echo "N=$N rc=$?"; tail -c 600 $O/rehearsal_n$N.err
    python - $N <<'PY'
import json,sys
N=sys.argv[1]
for l in open('gpurun_out/r06f/rehearsal_gloo_n%s.json'%N):
    if l.startswith('{'):
        d=json.loads(l); print('N', N, d['value'], d['ms_per_step'], d['max_abs_offset_error_px'], d['pairs_failed'], d['tiles_per_rank'])
        for r in d['per_rank']: print(r)
PY
done
unset VFSMS_DIST_BACKEND
timeout 400 python bench.py --force-dist --steps 10 --warmup 3 --cpu-sample 0 --no-host-leg --no-cold-leg --project-shards 8 > $O/bench_force_dist.json 2> $O/fd.err; tail -c 300 $O/fd.err
python - <<'PY'
import json
for l in open('gpurun_out/r06f/bench_force_dist.json'):
    if l.startswith('{'):
        d=json.loads(l); print('force-dist', d['value'], d['collective']['backend']); v=d['projected_scaling']['N=8']; print(v['projected_pairs_per_s'], v['projected_efficiency_vs_this_run_at_1'], v['tail_ms_gather_assemble_learn'], v['gather'])
PY
