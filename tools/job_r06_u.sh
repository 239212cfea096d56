#!/bin/bash
# timing experiment (wrong mosaics on purpose): what the statistics kernel of the fuse spends on its fences / on the ramps tail
mkdir -p gpurun_out/r06u
for L in FREF FNOFENCE FNOTAIL FBOTH FREF FNOFENCE FNOTAIL FBOTH; do
  VFSMS_LIB=build_ab/$L.so timeout 200 python bench.py --method fuse --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r06u/fuse_$L.json 2> gpurun_out/r06u/fuse_$L.err
  python - $L <<'PY'
import json,sys
ok=False
for l in open('gpurun_out/r06u/fuse_%s.json'%sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], d['ms_per_step'], d['stages']); ok=True
if not ok: print(sys.argv[1], 'no line', open('gpurun_out/r06u/fuse_%s.err'%sys.argv[1]).read()[-300:])
PY
done
