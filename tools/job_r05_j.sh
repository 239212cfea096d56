#!/bin/bash
mkdir -p gpurun_out/final4
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "work_list or mode_vote or rccl or full_width" > gpurun_out/final4/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/final4/pytest.log; tail -15 gpurun_out/final4/pytest.log
