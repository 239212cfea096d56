#!/bin/bash
# round 6, call M: the rocprofv3 evidence of the round on the final kernels (kernel trace of the default command and the three other methods,
# then the SQ / FETCH / WRITE / TA / MFMA counter passes), stamped with the build id
bash tools/profile_round.sh r06 pmc > gpurun_out/prof_r06_stdout.txt 2>&1
tail -5 gpurun_out/prof_r06_stdout.txt
ls gpurun_out/prof_r06
head -12 gpurun_out/prof_r06/kernel_stats.csv
