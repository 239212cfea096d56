#!/bin/bash
# round 5, last GPU call: the whole GPU suite and smoke on the final tree
mkdir -p gpurun_out/final3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/final3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final3/pytest.log; tail -3 gpurun_out/final3/pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/final3/smoke.log 2>&1; tail -1 gpurun_out/final3/smoke.log
