#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun); outputs under gpurun_out/prof_$TAG.
#   bash tools/profile_round.sh r01
# pass 1: --kernel-trace --stats equivalent (rocpd database -> per-kernel CSV by tools/rocpd_summary.py afterwards)
# pass 2..4: PMC passes, each in its own run (wave-cycle breakdown; FETCH_SIZE; WRITE_SIZE), kernel-trace only.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --cpu-sample 0"
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES -d $OUT/pmc_sq -o pmc --output-format csv -- $CMD > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- $CMD > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
python tools/rocpd_summary.py $(ls $OUT/trace/*results.db | head -1) $OUT/kernel_stats.csv
python - <<PY
import csv, glob, collections, json
def agg(path):
    a=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for row in csv.DictReader(open(path)):
        k=row['Kernel_Name'].split('(')[0]
        a[k][row['Counter_Name']]+=float(row['Counter_Value'])
    return a
out=open('$OUT/pmc_summary.txt','w')
sq=agg(glob.glob('$OUT/pmc_sq/*counter_collection.csv')[0])
out.write('# wave-cycle breakdown per kernel (rocprofv3 --pmc SQ_*; percentages of SQ_WAVE_CYCLES)\n')
for k,v in sq.items():
    if k.startswith('k_'):
        w=v.get('SQ_WAVE_CYCLES',1)
        out.write('%-18s waves=%.3g wave_cycles=%.3g ' % (k, v.get('SQ_WAVES',0), w) + ' '.join('%s=%.1f%%'%(c.replace('SQ_',''),100*x/w) for c,x in sorted(v.items()) if c not in ('SQ_WAVE_CYCLES','SQ_WAVES')) + '\n')
# dispatch counts from the trace
disp=collections.Counter()
for row in csv.DictReader(open(glob.glob('$OUT/pmc_fetch/*kernel_trace.csv')[0])):
    disp[row['Kernel_Name'].split('(')[0]]+=1
f=agg(glob.glob('$OUT/pmc_fetch/*counter_collection.csv')[0]); w=agg(glob.glob('$OUT/pmc_write/*counter_collection.csv')[0])
out.write('\n# HBM traffic per launch (MI355X_MICROARCH.md HBM section): bytes = FETCH_SIZE*1024*2 (gfx950 reports half of a wide\n# coalesced read stream; scattered/narrow accesses uncalibrated) + WRITE_SIZE*1024; separate --pmc passes\n')
for k in sorted(f):
    if k.startswith('k_'):
        n=max(disp.get(k,1),1)
        fb=f[k].get('FETCH_SIZE',0)*1024; wb=w.get(k,{}).get('WRITE_SIZE',0)*1024
        out.write('%-18s launches=%d fetch_raw=%.4g B/launch fetch_x2=%.4g B/launch write=%.4g B/launch total(x2 rule)=%.4g B/launch\n' % (k,n,fb/n,2*fb/n,wb/n,(2*fb+wb)/n))
out.close()
print(open('$OUT/pmc_summary.txt').read())
PY
