#!/bin/bash
# round 5, GPU call 4: XCD-affine ROI-major descriptor tickets (and the software-pipelined staging at occupancy 4) against round 4's ticket order
mkdir -p gpurun_out/r05d
O=gpurun_out/r05d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "surf or dll or full_width or fused or config4_tile or zircon or edge" > $O/pytest_surf.log 2>&1; tail -4 $O/pytest_surf.log
for L in X_oldtickets X_new P4 P4_oldtickets X_oldtickets X_new P4 P4_oldtickets; do
  echo "== $L" >> $O/ab.txt; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py 16 60 2>&1 | tail -2 >> $O/ab.txt
done
cat $O/ab.txt | cut -c1-330
# the same on a launch of the bench's size (40 pairs = 80 ROIs)
for L in X_oldtickets X_new X_oldtickets X_new; do
  echo "== $L (40 pairs)" >> $O/ab40.txt; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py 40 30 2>&1 | tail -2 >> $O/ab40.txt
done
cat $O/ab40.txt | cut -c1-330
for L in X_oldtickets X_new; do
  VFSMS_LIB=build_ab/$L.so timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f_$L -o pmc --output-format csv -- python tools/microbench.py 40 2 > /dev/null 2> $O/f_$L.err
  python - $O/f_$L $L <<'PY'
import csv, glob, collections, sys
a=collections.defaultdict(float); n=collections.Counter(); seen=set()
for row in csv.DictReader(open(glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True)[0])):
    k=row['Kernel_Name'].split('(')[0]
    if 'describe' not in k: continue
    a[k]+=float(row['Counter_Value']); key=(row['Dispatch_Id'],k)
    if key not in seen: seen.add(key); n[k]+=1
for k in a: print(sys.argv[2], k, 'FETCH_SIZE raw %.4g B/launch (x2 rule %.4g)' % (a[k]*1024/n[k], 2*a[k]*1024/n[k]))
PY
  rm -rf $O/f_$L
done
timeout 200 python bench.py --cpu-sample 0 --no-host-leg --no-cold-leg --prior same --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); print('bench', d['value'], d['ms_per_step'], 'err', d['max_abs_offset_error_px'], d['pairs_failed'])
        print('   ', {k: round(v['ms']/d['steps'],2) for k,v in d['stages'].items()})
PY
