#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "two_ranks or need_aware or papers_section or auto_selects or row_gather or gather_rows" > gpurun_out/r03_pytest_halo.log 2>&1
tail -5 gpurun_out/r03_pytest_halo.log
timeout 900 python tools/scale_model.py --papers > gpurun_out/r03_scale_model.md 2> gpurun_out/r03_scale_model.err
tail -30 gpurun_out/r03_scale_model.md; tail -5 gpurun_out/r03_scale_model.err
timeout 600 python tools/probe_tlb.py --sizes 0.25,0.5,1,2,4,8,16,32,57 --modes torch,vmm2m --reps 1 > gpurun_out/r03_probe_tlb_sizes.log 2>&1
tail -20 gpurun_out/r03_probe_tlb_sizes.log
