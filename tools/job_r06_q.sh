#!/bin/bash
# round 6, call Q: the device stage of the JPEG decode (raw 4:2:0 planes -> k_ingest_420): parity, then decode-inclusive colour ingest A B A B
mkdir -p gpurun_out/r06q
O=gpurun_out/r06q
timeout 900 python -m pytest tests -m gpu -x -q -k "jpeg or ingest or colour or color or driver or main_py or break" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for V in 0 1 0 1; do
  VFSMS_JPEG_RAW420=$V timeout 300 python bench.py --from-files --color --steps 5 --warmup 1 > $O/ffc_$V.json 2> $O/ffc_$V.err
  python - $V <<'PY'
import json,sys
for l in open('gpurun_out/r06q/ffc_%s.json'%sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('raw420=%s'%sys.argv[1], d['value'], d['ms_per_step'], 'decode only', d['decode_only_tiles_per_s'], 'reg only', d['registration_only_pairs_per_s'], d.get('ingest_thread_ms_per_tile'))
PY
done
timeout 300 python tools/e2e_dataset.py > $O/e2e_dataset.json 2> $O/e2e.err; python -c "
import json; d=json.load(open('$O/e2e_dataset.json')); print('e2e', d['default_streamed_pinned_bands']['seconds'], d['default_streamed_pinned_bands_again']['seconds'])"
