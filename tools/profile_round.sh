#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun); outputs under gpurun_out/prof_$TAG.
#   bash tools/profile_round.sh r02 [pmc]
# pass 1: --kernel-trace --stats of the DEFAULT bench command (rocpd database -> per-kernel CSV by tools/rocpd_summary.py)
# passes 2..4 (only with "pmc"): PMC passes, each in its own run, kernel-trace only (SQ instruction mix / wave-cycle breakdown;
#   FETCH_SIZE; WRITE_SIZE).  PMC serialises kernels, so these run ONE step of the same workload without the steady-clock warmup.
TAG=${1:-r02}; PMC=${2:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# (VFSMS_BENCH_PRIME=1: the scan pattern is known from the first step on, so that EVERY launch in the trace is a steady-state launch and the
#  per-kernel averages are those of the timed region -- the unprimed default run mixes in the 13 small launches of the first, cold step;
#  --no-cold-leg for the same reason.  The JSON written under the profiler says so: path_memory_primed_for_profiling.)
export VFSMS_BENCH_PRIME=1
CMD="python bench.py --cpu-sample 0 --no-cold-leg"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.err
python tools/rocpd_summary.py $(ls $OUT/trace/*results.db | head -1) $OUT/kernel_stats.csv
for m in orb phase fuse; do
  VFSMS_BENCH_PRIME=0 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_$m -o trace -- python bench.py --method $m --cpu-sample 0 --no-cold-leg > $OUT/trace_bench_$m.json 2> $OUT/trace_$m.err
  python tools/rocpd_summary.py $(ls $OUT/trace_$m/*results.db | head -1) $OUT/kernel_stats_$m.csv
done
rm -rf $OUT/trace $OUT/trace_orb $OUT/trace_phase $OUT/trace_fuse
[ "$PMC" = "pmc" ] || exit 0
export VFSMS_BENCH_MIN_WARM=0
PCMD="python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-host-leg --no-cold-leg"
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o pmc --output-format csv -- $PCMD > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- $PCMD > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- $PCMD > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
timeout 500 rocprofv3 --kernel-trace --pmc TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS_sum -d $OUT/pmc_ta -o pmc --output-format csv -- $PCMD > $OUT/pmc_ta_bench.json 2> $OUT/pmc_ta.err
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/pmc_mfma -o pmc --output-format csv -- $PCMD > $OUT/pmc_mfma_bench.json 2> $OUT/pmc_mfma.err
python - <<PY
import csv, glob, collections
dur=collections.defaultdict(float)          # kernel -> summed (End - Start) ns over the dispatches of the pass agg() read last
def agg(pat):
    a=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); seen=set()
    f=glob.glob(pat, recursive=True)
    dur.clear()
    if not f: return a, n
    for row in csv.DictReader(open(f[0])):
        k=row['Kernel_Name'].split('(')[0]
        a[k][row['Counter_Name']]+=float(row['Counter_Value'])
        key=(row.get('Dispatch_Id'),k)
        if key not in seen:
            seen.add(key); n[k]+=1
            try: dur[k]+=float(row['End_Timestamp'])-float(row['Start_Timestamp'])
            except (KeyError, ValueError): pass
    return a, n
out=open('$OUT/pmc_summary.txt','w')
# what was profiled: bench.py compares this with the library it loads and reports "pmc_stale" when the kernels changed behind the profile
import sys
sys.path.insert(0, '.')
from tools.build_id import build_id
out.write('# build: ' + ' '.join('%s=%s' % kv for kv in sorted(build_id().items())) + '\n')
sq,nsq=agg('$OUT/pmc_sq/**/*counter_collection.csv')
sqdur=dict(dur)
out.write('# SQ counters per launch (rocprofv3 --pmc, one step of the default bench workload, kernels serialised); quad-cycle units for\\n# WAVE_CYCLES / WAIT_ANY / ACTIVE_INST_ANY; INSTS_* = wave-instructions (one VALU wave-instruction occupies its SIMD for 4 cycles)\\n')
for k,v in sorted(sq.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',0)):
    if 'k_' in k:
        n=max(nsq[k],1)
        # eff_clock_ghz: SQ_BUSY_CYCLES (summed over the 32 shader engines) / 32 = cycles the launch lasted, over its traced duration in
        # the same pass: the shader clock the kernel actually ran at (the roofline's peak constant assumes 2.4 GHz)
        clk=(v.get('SQ_BUSY_CYCLES',0)/32.0)/sqdur[k] if sqdur.get(k) else 0.0
        out.write('%-28s launches=%d '%(k[:28],n)+' '.join('%s=%.4g'%(c.replace('SQ_',''),x/n) for c,x in sorted(v.items()))+(' eff_clock_ghz=%.3f dur_ms=%.4g'%(clk,sqdur[k]/n/1e6) if clk else '')+'\\n')
f,nf=agg('$OUT/pmc_fetch/**/*counter_collection.csv'); w,nw=agg('$OUT/pmc_write/**/*counter_collection.csv')
out.write('\\n# HBM traffic per launch (MI355X_MICROARCH.md HBM section): bytes = FETCH_SIZE*1024*2 (gfx950 reports half of a wide\\n# coalesced read stream; scattered/narrow accesses uncalibrated) + WRITE_SIZE*1024; separate --pmc passes\\n')
for k in sorted(f):
    if 'k_' in k:
        n=max(nf[k],1)
        fb=f[k].get('FETCH_SIZE',0)*1024; wb=w.get(k,{}).get('WRITE_SIZE',0)*1024; nn=max(nw.get(k,n),1)
        out.write('%-28s launches=%d fetch_raw=%.4g B/launch fetch_x2=%.4g B/launch write=%.4g B/launch total(x2 rule)=%.4g B/launch\\n' % (k[:28],n,fb/n,2*fb/n,wb/nn,2*fb/n+wb/nn))
ta,nta=agg('$OUT/pmc_ta/**/*counter_collection.csv'); mf,nmf=agg('$OUT/pmc_mfma/**/*counter_collection.csv')
out.write('\n# texture-address path and matrix pipe per launch (separate passes): TA_TA_BUSY summed over the 256 CUs (/256 = busy cycles of one\n# TA; the launch lasts SQ_BUSY_CYCLES/32 cycles); SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs\n')
for k in sorted(ta, key=lambda k:-ta[k].get('TA_TA_BUSY',0)):
    if 'k_' in k:
        n=max(nta[k],1); m=mf.get(k,{}); nm=max(nmf.get(k,n),1)
        busy=m.get('SQ_BUSY_CYCLES',0)/nm/32.0
        out.write('%-28s launches=%d TA_BUSY=%.4g TA_READ_WAVEFRONTS=%.4g ta_busy_frac=%.3f MFMA_BUSY_CYCLES=%.4g mfma_busy_frac=%.3f WAIT_INST_ANY=%.4g ACTIVE_INST_VALU=%.4g\n' % (
            k[:28],n,ta[k].get('TA_TA_BUSY',0)/n,ta[k].get('TA_FLAT_READ_WAVEFRONTS_sum',ta[k].get('TA_FLAT_READ_WAVEFRONTS',0))/n,
            (ta[k].get('TA_TA_BUSY',0)/n/256.0)/busy if busy else 0, m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/nm,
            (m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/nm/1024.0)/busy if busy else 0, m.get('SQ_WAIT_INST_ANY',0)/nm, m.get('SQ_ACTIVE_INST_VALU',0)/nm))
# what the PMC runs described, so that counter TOTALS can be set against an algorithmic count whatever the launch sizes were: the SQ pass ran
# (warm-up-0 step + 1 timed step) of the bench line stored next to it, plus one single-ROI call (the samples-per-keypoint probe)
import json
try:
    b=json.loads(open('$OUT/pmc_sq_bench.json').read().strip().splitlines()[-1])
    steps=b['steps']+max(b['warmup'],1)
    pt=b.get('process_totals_through_timed_steps')
    if pt:      # counted by the run itself (incl. the registration of the prior instance): + the one-ROI samples-per-keypoint probe
        kps=pt['keypoints']+b['keypoints_per_roi']; att=pt['attempts']
    else:
        kps=2.0*b['attempts_per_step']*b['keypoints_per_roi']*steps+b['keypoints_per_roi']; att=b['attempts_per_step']*steps
    out.write('\n# the SQ pass as a whole (for ratios of counter totals to algorithmic counts)\n')
    out.write('pmc_run steps=%d attempts=%.0f keypoints=%.0f samples_per_keypoint=%.1f\n' % (steps, att, kps, b['roofline']['samples_per_keypoint']))
except Exception as e:
    out.write('# pmc_run: %s\n' % e)
out.close()
print(open('$OUT/pmc_summary.txt').read())
PY
# ---- the phase path's own kernels (csrc/phase_kernels.hip, LDS transforms): batches of 32 attempts of the bench's two strip shapes ----
PHCMD="python tools/phase_ab.py 32 2 t"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES -d $OUT/pmc_ph_sq -o pmc --output-format csv -- $PHCMD > $OUT/pmc_ph_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_ph_fetch -o pmc --output-format csv -- $PHCMD > $OUT/pmc_ph_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_ph_write -o pmc --output-format csv -- $PHCMD > $OUT/pmc_ph_write.log 2>&1
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
sq,nsq=agg('$OUT/pmc_ph_sq/**/*counter_collection.csv'); f,nf=agg('$OUT/pmc_ph_fetch/**/*counter_collection.csv'); w,nw=agg('$OUT/pmc_ph_write/**/*counter_collection.csv')
out=open('$OUT/pmc_summary.txt','a')
out.write('\n# the phase path (separate passes of tools/phase_ab.py 32 2 t: batches of 32 attempts, 409 x 2048 strips and 2048 x 409 strips (transposed first)):\n# per launch; valu_busy_frac = INSTS_VALU x 4 cycles / (BUSY_CYCLES / 32 x 1024 SIMDs); traffic by the same x2 rule as above\n')
for k in sorted(sq):
    if 'k_phase' in k or 'k_peak' in k:
        n=max(nsq.get(k,1),1); v=sq[k]
        busy=v.get('SQ_BUSY_CYCLES',0)/n/32.0
        fb=f.get(k,{}).get('FETCH_SIZE',0)*1024/max(nf.get(k,n),1); wb=w.get(k,{}).get('WRITE_SIZE',0)*1024/max(nw.get(k,n),1)
        out.write('%-24s launches=%d attempts_per_launch=32 '%(k[:24],n)+' '.join('%s=%.4g'%(c.replace('SQ_',''),x/n) for c,x in sorted(v.items()))+
                  ' valu_busy_frac=%.3f fetch_x2=%.4g B/launch write=%.4g B/launch total(x2 rule)=%.4g B/launch\n' % ((v.get('SQ_INSTS_VALU',0)/n*4.0)/(busy*1024.0) if busy else 0.0, 2*fb, wb, 2*fb+wb))
out.close()
print(open('$OUT/pmc_summary.txt').read()[-2500:])
PY
bash tools/profile_mosaic_pmc.sh $OUT
rm -rf $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_ta $OUT/pmc_mfma $OUT/pmc_ph_sq $OUT/pmc_ph_fetch $OUT/pmc_ph_write
