#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --force-sharded --steps 5 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],3),'ms/step', j['config']['diagnostics'], j['config']['plan'])"
timeout 600 python bench.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],3),'ms/step', j['config']['diagnostics'])"
