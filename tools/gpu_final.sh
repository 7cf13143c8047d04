#!/bin/bash
# what the driver does at round end, in one session: GPU parity suite, smoke(), default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
tail -2 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; wc -l gpurun_out/bench.json; cut -c1-260 gpurun_out/bench.json; tail -1 gpurun_out/bench.err
