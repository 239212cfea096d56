cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d gpurun_out/pmc2 -o pmc2 --output-format csv -- python tools/microbench.py 8 3 > gpurun_out/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/pmc2/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for row in csv.DictReader(open(f[0])):
    k=row['Kernel_Name'].split('(')[0]
    agg[k][row['Counter_Name']]+=float(row['Counter_Value'])
for k,v in agg.items():
    if k.startswith('k_'):
        w=v.get('SQ_WAVE_CYCLES',1)
        print(k, 'wave_cycles=%.3g'%w, ' '.join('%s=%.1f%%'%(c.replace('SQ_',''),100*x/w) for c,x in sorted(v.items()) if c!='SQ_WAVE_CYCLES'))
PY
