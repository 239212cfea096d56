#!/bin/bash
# tools/profile_mosaic_pmc.sh <out dir>: append the mosaic walk's HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) to
# <out dir>/pmc_summary.txt (called by tools/profile_round.sh; the summary's build stamp must be this build's)
OUT=$1
# ---- the mosaic walk's kernels: HBM traffic per launch, and how many launches one mosaic is ----
FZCMD="python bench.py --method fuse --steps 2 --warmup 1 --cpu-sample 0"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fz_fetch -o pmc --output-format csv -- $FZCMD > $OUT/pmc_fz_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_fz_write -o pmc --output-format csv -- $FZCMD > $OUT/pmc_fz_write.log 2>&1
python - <<PY
import csv, glob, collections
def agg(pat):
    a=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    f=glob.glob(pat, recursive=True)
    if not f: return a, {}
    for row in csv.DictReader(open(f[0])):
        k=row['Kernel_Name'].split('(')[0]
        a[k][row['Counter_Name']]+=float(row['Counter_Value']); n[k].add(row.get('Dispatch_Id'))
    return a, {k: len(v) for k, v in n.items()}
f,nf=agg('$OUT/pmc_fz_fetch/**/*counter_collection.csv'); w,nw=agg('$OUT/pmc_fz_write/**/*counter_collection.csv')
out=open('$OUT/pmc_summary.txt','a')
mosaics=max(nf.get('k_paste',0),1)
out.write('\n# the mosaic walk (separate passes of bench.py --method fuse --steps 2 --warmup 1): per launch, by the x2 rule; one mosaic = launches / mosaics launches of a kernel\n# (k_paste runs once per mosaic: the first tile)\n')
for k in sorted(f):
    if 'k_fuse' in k or 'k_paste' in k:
        n=max(nf.get(k,1),1)
        fb=f[k].get('FETCH_SIZE',0)*1024/n; wb=w.get(k,{}).get('WRITE_SIZE',0)*1024/max(nw.get(k,n),1)
        out.write('%-32s launches=%d mosaics=%d fetch_x2=%.4g B/launch write=%.4g B/launch total(x2 rule)=%.4g B/launch\n' % (k[:32].replace('void ',''),n,mosaics,2*fb,wb,2*fb+wb))
out.close()
print(open('$OUT/pmc_summary.txt').read()[-900:])
PY
rm -rf $OUT/pmc_fz_fetch $OUT/pmc_fz_write
