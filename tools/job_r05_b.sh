#!/bin/bash
# round 5, GPU call 2: balanced descriptor staging -- parity, A/B against the round-4 form, phase cycle counters of both; VALU probe; configs[4]
mkdir -p gpurun_out/r05b
O=gpurun_out/r05b
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "surf or dll or full_width or fused or config4_tile or zircon or edge" > $O/pytest_surf.log 2>&1; tail -6 $O/pytest_surf.log
for L in S_old S_new S_big S_small S_old S_new S_big S_small; do
  echo "== $L" >> $O/ab.txt; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py 16 60 2>&1 | tail -2 >> $O/ab.txt
done
cat $O/ab.txt | cut -c1-330
echo "== round-4 staging (timing build)" > $O/desc_cycles.txt; VFSMS_TIMING_LIB=build_ab/T_old.so timeout 120 python tools/desc_timing.py >> $O/desc_cycles.txt 2>&1
echo "== balanced staging (timing build)" >> $O/desc_cycles.txt; timeout 120 python tools/desc_timing.py >> $O/desc_cycles.txt 2>&1
cat $O/desc_cycles.txt | cut -c1-250
timeout 200 tools/bin/valu_peak > $O/valu_peak.txt 2>&1; cut -c1-420 $O/valu_peak.txt
timeout 480 python bench.py --rows 32 --cols 32 --tile 4096 --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg --no-cold-leg --prior same > $O/bench_config4_surf.json 2> $O/bench_config4_surf.err; tail -c 1500 $O/bench_config4_surf.err; cut -c1-1500 $O/bench_config4_surf.json
