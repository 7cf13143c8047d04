#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 900 python tools/bench_aggregators.py > gpurun_out/agg.log 2>&1; echo "exit $?" >> gpurun_out/agg.log
tail -25 gpurun_out/pytest_gpu.log; grep -E "nafs|gather|d=147" gpurun_out/agg.log
