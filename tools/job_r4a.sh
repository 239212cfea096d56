#!/bin/bash
# round 4, first GPU call: new tests, VALU peak probe, decode-thread sweep, colour / gray from-files, default bench
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
tools/bin/valu_peak > $O/valu_peak.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for t in 16 32 64; do
  timeout 300 python bench.py --from-files --decode-threads $t --steps 5 > $O/ff_gray_$t.json 2> $O/ff_gray_$t.err
  timeout 300 python bench.py --from-files --color --decode-threads $t --steps 5 > $O/ff_color_$t.json 2> $O/ff_color_$t.err
done
timeout 300 python bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err
head -c 600 $O/bench_default.json; echo
for f in $O/ff_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["decode_only_ms_per_step"], d["registration_only_ms_per_step"], d["end_to_end_over_slower_stage"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
cat $O/valu_peak.txt
