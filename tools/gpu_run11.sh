#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 900 python tools/papers_shard_bench.py > gpurun_out/papers_shard.log 2>&1; echo "exit $?" >> gpurun_out/papers_shard.log
tail -4 gpurun_out/pytest_gpu.log; grep -E "PAPERS|exit|Error" gpurun_out/papers_shard.log
