#!/bin/bash
# round 4, second GPU call: describe parity, EXP variants A/B, phase timing, ingest with pinned staging
mkdir -p gpurun_out/r4b
O=gpurun_out/r4b
timeout 600 python -m pytest tests -m gpu -x -q -k "surf or config4 or ingest or colour_mode or full_size" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for L in E3 E0 E1 E2 E3 E0; do
  echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py 16 60 2>&1 | tail -2
done | tee $O/ab.log
timeout 200 python tools/desc_timing.py > $O/desc_timing.txt 2>&1; cat $O/desc_timing.txt
for t in 16 32; do
  timeout 300 python bench.py --from-files --decode-threads $t --steps 5 > $O/ff_gray_$t.json 2> $O/ff_gray_$t.err
  timeout 300 python bench.py --from-files --color --decode-threads $t --steps 5 > $O/ff_color_$t.json 2> $O/ff_color_$t.err
done
for f in $O/ff_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["decode_only_ms_per_step"], d["registration_only_ms_per_step"], d["end_to_end_over_slower_stage"], d.get("ingest_thread_ms_per_tile"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
