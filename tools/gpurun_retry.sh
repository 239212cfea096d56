#!/bin/bash
# tools/gpurun_retry.sh TIMEOUT 'command' : gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged)
T=$1; shift
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do
    /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 90
done
exit 3
