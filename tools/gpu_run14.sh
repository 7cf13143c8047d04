#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
for cfg in "--exchange p2p" "--exchange push"; do
  echo "== $cfg"; timeout 600 python bench.py --force-sharded --steps 5 $cfg 2>gpurun_out/err.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],3),'ms/step', j['config']['plan'])"; grep -v "^/opt\|socket" gpurun_out/err.log | tail -3
done
timeout 600 python bench.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('single', round(j['ms_per_step'],3),'ms/step')"
