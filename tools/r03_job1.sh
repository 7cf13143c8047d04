#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/probe_tlb.py --sizes 1,57 > gpurun_out/r03_probe_tlb.log 2>&1
tail -40 gpurun_out/r03_probe_tlb.log
timeout 900 tools/tlb_pmc.sh torch,contiguous,vmm2m 1,57
timeout 600 python tools/bench_aggregators.py > gpurun_out/r03_aggregators.log 2>&1
tail -5 gpurun_out/r03_aggregators.log
