python tools/microbench.py 8 5 2>&1 | tail -2
python tools/microbench.py 16 4 | tail -2
