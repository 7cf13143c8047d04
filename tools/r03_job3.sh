#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "gate or learnable or fuzz or weighted or models or model" > gpurun_out/r03_pytest_gate.log 2>&1
tail -15 gpurun_out/r03_pytest_gate.log
timeout 600 python tools/bench_aggregators.py > gpurun_out/r03_aggregators.log 2>&1
grep -i "gate\|jk\| sum \|nafs" gpurun_out/r03_aggregators.log
timeout 900 python tools/scale_model.py --papers > gpurun_out/r03_scale_model.md 2> gpurun_out/r03_scale_model.err
grep -A16 "Pipelining" gpurun_out/r03_scale_model.md; tail -3 gpurun_out/r03_scale_model.err
