#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
timeout 600 python bench.py --workload S0_pubmed --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_S0.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
for f in bench bench_S0; do grep -v exit gpurun_out/$f.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['config']['workload'][:12], round(j['ms_per_step'],4),'ms/step', round(j['value']/1e12,3),'e12 frac', round(j['roofline']['frac'],3))"; done
