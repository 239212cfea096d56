#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06z
for ch in 32 16 8 4; do
  echo "== CHUNK=$ch"
  VFSMS_PHASE_CHUNK=$ch timeout 300 python tools/phase_ab.py 32 20 t 2>&1 | grep "LDS transforms"
done
timeout 600 python bench.py --method phase > $R/gpurun_out/r06z/bench_phase.json 2> $R/gpurun_out/r06z/bench_phase.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06z/bench_phase.json").read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], d.get("roofline"))
PY
VFSMS_PHASE_LDS_FFT=0 timeout 600 python bench.py --method phase --cpu-sample 0 > $R/gpurun_out/r06z/bench_phase_rocfft.json 2> $R/gpurun_out/r06z/bench_phase_rocfft.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06z/bench_phase_rocfft.json").read().strip().splitlines()[-1])
print("rocfft:", d["value"], d["unit"], d["ms_per_step"])
PY
