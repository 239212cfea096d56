#!/bin/bash
# A/B comparison of two builds of libvfsms.so on the fixed micro batch (stage times repeat within ~0.2 % inside ONE gpurun call).
#   tools/ab.sh build        here (no GPU): build_ab/A.so = the committed tree (git stash), build_ab/B.so = the working tree
#   tools/ab.sh run [N] [K]  on the GPU box (through gpurun): A B A B over tools/microbench.py N K (default 8 pairs, 100 repetitions)
# e.g.  tools/ab.sh build && gpurun --timeout 400 -- 'bash tools/ab.sh run'
set -e
cd "$(dirname "$0")/.."
case "${1:-run}" in
build)
    mkdir -p build_ab
    make -C imagestitch_amd/csrc -j8 -s && cp imagestitch_amd/lib/libvfsms.so build_ab/B.so
    git stash -q && { make -C imagestitch_amd/csrc -j8 -s; cp imagestitch_amd/lib/libvfsms.so build_ab/A.so; git stash pop -q; }
    make -C imagestitch_amd/csrc -j8 -s
    ls -la build_ab ;;
run)
    for L in A B A B; do
        echo "== $L"; VFSMS_LIB=build_ab/$L.so timeout 120 python tools/microbench.py ${2:-8} ${3:-100} 2>&1 | tail -2
    done ;;
esac
