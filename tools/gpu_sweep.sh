#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python tools/sweep_spmm.py --exp knobs,plan,dsweep > gpurun_out/sweep2.log 2>&1
echo "sweep exit $?" >> gpurun_out/sweep2.log
cat gpurun_out/sweep2.log | grep -v "^/opt" | tail -120
