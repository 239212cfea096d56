cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d gpurun_out/pmc1 -o pmc1 --output-format csv -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/pmc1.log 2>&1
ls gpurun_out/pmc1
python - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/pmc1/*counter_collection.csv')
print(f)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for row in csv.DictReader(open(f[0])):
    k=row['Kernel_Name'].split('(')[0]
    agg[k][row['Counter_Name']]+=float(row['Counter_Value'])
    if row['Counter_Name']=='SQ_WAVES': cnt[k]+=1
for k,v in agg.items():
    w=max(v.get('SQ_WAVES',1),1)
    print(k, 'dispatches',cnt[k],'waves %.3g'%w, ' '.join('%s/wave=%.1f'%(c.replace('SQ_',''),x/w) for c,x in v.items() if c!='SQ_WAVES'))
PY
