#!/bin/bash
# round 5, GPU call 1: evidence first -- PC sampling of the descriptor kernels; RCCL at world size 1; bench --force-dist; configs[4] on one GPU
mkdir -p gpurun_out/r05a
O=gpurun_out/r05a
bash tools/pcsamp.sh r05 'k_describe' 16 6 > $O/pcsamp.log 2>&1; tail -60 $O/pcsamp.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rccl" > $O/pytest_rccl.log 2>&1; tail -15 $O/pytest_rccl.log
timeout 400 python bench.py --force-dist --cpu-sample 0 --steps 10 --warmup 2 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -c 600 $O/bench_force_dist.err; cut -c1-400 $O/bench_force_dist.json
timeout 900 python bench.py --rows 32 --cols 32 --tile 4096 --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg --no-cold-leg > $O/bench_config4_surf.json 2> $O/bench_config4_surf.err; tail -c 1500 $O/bench_config4_surf.err; cut -c1-1500 $O/bench_config4_surf.json
