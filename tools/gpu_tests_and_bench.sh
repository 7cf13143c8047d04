#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
for W in S2_gamlp S0_pubmed S1_products; do timeout 600 python bench.py --workload $W --steps 5 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_$W.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['config']['workload'][:12], round(j['ms_per_step'],4),'ms/step', round(j['value']/1e12,3),'e12 frac', round(j['roofline']['frac'],3))"; done
