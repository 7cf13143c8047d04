#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/bench_aggregators.py > gpurun_out/agg.log 2>&1; echo "exit $?" >> gpurun_out/agg.log
grep -v "^/opt" gpurun_out/agg.log | tail -60
