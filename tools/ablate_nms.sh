for a in 0 1 2 3; do echo "VFSMS_NMS_ABLATE=$a"; VFSMS_NMS_ABLATE=$a python tools/microbench.py 16 4 2>&1 | tail -1; done
