#!/bin/bash
# round-3 evidence run: full GPU suite, smoke, default bench line, kernel stats + PMC of the S1 command, sharded path on one rank
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --maxfail=10 > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03_pytest_gpu.log
tail -6 gpurun_out/r03_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/r03_smoke.log; tail -2 gpurun_out/r03_smoke.log
timeout 700 python bench.py > gpurun_out/r03_bench_S1.json 2> gpurun_out/r03_bench_S1.err; echo "bench exit $?" >> gpurun_out/r03_bench_S1.err
cut -c1-300 gpurun_out/r03_bench_S1.json; tail -1 gpurun_out/r03_bench_S1.err
timeout 900 tools/gpu_profile.sh S1_products > gpurun_out/r03_gpu_profile.log 2>&1; tail -3 gpurun_out/r03_gpu_profile.log
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 5 > gpurun_out/r03_bench_sharded1.json 2> gpurun_out/r03_bench_sharded1.err
cut -c1-200 gpurun_out/r03_bench_sharded1.json; tail -2 gpurun_out/r03_bench_sharded1.err
timeout 600 python tools/bench_aggregators.py > gpurun_out/r03_aggregators.log 2>&1; grep -c AGG gpurun_out/r03_aggregators.log
