#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
timeout 600 python bench.py --force-sharded --steps 5 > gpurun_out/bench_sharded1.log 2>&1
timeout 600 python bench.py --workload S0_pubmed --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_S0.log 2>&1
timeout 600 python bench.py --workload S2_gamlp --steps 5 --no-cpu-baseline > gpurun_out/bench_S2.log 2>&1
timeout 900 python tools/pasca_sweep.py > gpurun_out/pasca_sweep.log 2>&1; echo "exit $?" >> gpurun_out/pasca_sweep.log
for f in bench bench_sharded1 bench_S0 bench_S2; do tail -1 gpurun_out/$f.log | cut -c1-420; done
grep -v "^/opt" gpurun_out/pasca_sweep.log | tail -40
