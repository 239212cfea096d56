#!/bin/bash
# PC sampling of the descriptor kernels (GPU box; through gpurun): rocprofv3 --pc-sampling-beta-enabled over the fixed micro batch with the
# line-table build of the library (make -C imagestitch_amd/csrc lines -> tools/libvfsms_lines.so: same instructions, PCs map to csrc lines).
#   bash tools/pcsamp.sh TAG [kernel-regex] [N pairs] [K reps]   -> gpurun_out/pcsamp_TAG/{summary.txt, samples*.csv.gz, avail.txt}
# stochastic (hardware) sampling first -- it carries issue / stall reasons --, host_trap as the fallback.  No --pmc, no other trace domain
# than the kernel trace (needed to name the dispatches).
TAG=${1:-x}; KRE=${2:-k_describe}; N=${3:-16}; K=${4:-6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pcsamp_$TAG; mkdir -p $OUT
export VFSMS_LIB=${PCSAMP_LIB:-tools/libvfsms_lines.so}
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=on timeout 120 rocprofv3-avail info --pc-sampling > $OUT/avail.txt 2>&1
head -40 $OUT/avail.txt
ok=0
IFS=';' read -ra CFGS <<< "${PCSAMP_CFGS:-stochastic cycles 65536;stochastic cycles 1048576;host_trap time 100;host_trap time 1000}"
for CFG in "${CFGS[@]}"; do
  set -- $CFG
  rm -rf $OUT/raw
  echo "== trying $CFG"
  timeout 400 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 \
      -d $OUT/raw -o pcs --output-format csv -- python tools/microbench.py $N $K > $OUT/run_$1_$3.txt 2> $OUT/run_$1_$3.err
  rc=$?
  tail -2 $OUT/run_$1_$3.txt; tail -5 $OUT/run_$1_$3.err | cut -c1-300
  f=$(find $OUT/raw -name '*pc_sampling*csv' | head -1)
  if [ $rc -eq 0 ] && [ -n "$f" ] && [ $(wc -l < "$f") -gt 100 ]; then ok=1; echo "$CFG" > $OUT/config.txt; break; fi
done
[ $ok -eq 1 ] || { echo "pc sampling: no configuration worked"; ls -R $OUT | head -50; exit 0; }
python tools/pcsamp_summary.py $OUT/raw "$KRE" $OUT > $OUT/summary.txt
head -150 $OUT/summary.txt | cut -c1-220
rm -rf $OUT/raw
