#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
timeout 600 python bench.py --force-sharded --steps 5 > gpurun_out/bench_sharded1.log 2>&1; echo "exit $?" >> gpurun_out/bench_sharded1.log
timeout 600 python bench.py --strict --steps 5 --no-cpu-baseline > gpurun_out/bench_strict.log 2>&1
tail -4 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log; tail -2 gpurun_out/bench_sharded1.log; tail -1 gpurun_out/bench_strict.log
