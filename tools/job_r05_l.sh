#!/bin/bash
mkdir -p gpurun_out/final6
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "driver or ingest or colour or jpeg or main_py or line_scan" > gpurun_out/final6/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/final6/pytest.log; tail -4 gpurun_out/final6/pytest.log
timeout 200 python tools/e2e_dataset.py > gpurun_out/final6/e2e.json 2> gpurun_out/final6/e2e.err; python -c "
import json
for l in open('gpurun_out/final6/e2e.json'):
    if l.startswith('{'):
        e=json.loads(l)
        for k,v in e.items():
            if isinstance(v,dict): print(k, v['seconds'])"
