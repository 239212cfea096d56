#!/bin/bash
# round 6, call V: the fuse without agent fences (FNEW) against the fenced build (FREF): parity of every fuse / mosaic / canvas test, a stress loop
# with a second process loading the GPU, then the timing
mkdir -p gpurun_out/r06v
O=gpurun_out/r06v
timeout 900 python -m pytest tests -m gpu -x -q -k "fuse or mosaic or canvas or stitch or golden or blend or driver or main_py or walk" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
VFSMS_LIB=build_ab/FREF.so timeout 200 python tools/fuse_stress.py 3 | tee $O/stress.txt
timeout 200 python tools/fuse_stress.py 40 | tee -a $O/stress.txt
( for k in 1 2 3 4 5 6; do timeout 100 python tools/microbench.py 16 120 > /dev/null 2>&1; done ) &
BG=$!
sleep 20
timeout 300 python tools/fuse_stress.py 60 | tee -a $O/stress.txt
timeout 300 python tools/fuse_stress.py 12 10 9 2048 | tee -a $O/stress.txt
wait $BG
VFSMS_LIB=build_ab/FREF.so timeout 200 python tools/fuse_stress.py 2 10 9 2048 | tee -a $O/stress.txt
for L in FREF FNEW FREF FNEW; do
  VFSMS_LIB=build_ab/$L.so timeout 200 python bench.py --method fuse --steps 10 --warmup 3 --cpu-sample 0 > $O/fuse_$L.json 2> $O/fuse_$L.err
  python -c "
import json
for l in open('$O/fuse_$L.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$L', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done | tee $O/ab.txt
