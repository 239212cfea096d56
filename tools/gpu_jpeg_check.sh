#!/bin/bash
# the library's own JPEG decoder against the Pillow hand-over: full GPU suite, then the decode-inclusive bench with either decoder
mkdir -p gpurun_out/jpeg
O=gpurun_out/jpeg
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
run() { # name, env value, extra args
  VFSMS_NATIVE_JPEG=$2 timeout 150 python bench.py --from-files ${@:3} > $O/$1.json 2> $O/$1.err || tail -c 300 $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d['value'], 'ms', d['ms_per_step'], 'decode-only tiles/s', d['decode_only_tiles_per_s'], 'one', d['decode_one_tile_one_thread_ms'], 'reg-only', d['registration_only_pairs_per_s'], d['ingest_thread_ms_per_tile'], d['config']['decoder'][:12], 'err', d['max_abs_offset_error_px'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run color_native16 1 --color
run color_pillow16 0 --color
run gray_native16 1
run gray_pillow16 0
run color_native32 1 --color --decode-threads 32
run gray_native32 1 --decode-threads 32
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/jpeg/bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], d['ms_per_step'], d['attempts_per_step'], 'cold', d['value_cold_path'], 'err', d['max_abs_offset_error_px'], 'frac', d['roofline']['frac'])
PY
