#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "--pieces 1 --col-chunks 1" "--pieces 4 --col-chunks 1" "--pieces 1 --col-chunks 2" "--pieces 2 --col-chunks 2"; do
  echo "== $cfg"; timeout 600 python bench.py --force-sharded --steps 5 $cfg 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],3),'ms/step', round(j['value']/1e12,3),'e12')"
done
