#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=15 > gpurun_out/r03_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/r03_pytest_gpu.log
tail -25 gpurun_out/r03_pytest_gpu.log
timeout 600 python tools/e2e_host_propagate.py > gpurun_out/r03_e2e_host_propagate.log 2>&1
cat gpurun_out/r03_e2e_host_propagate.log | grep -v amdgpu.ids
