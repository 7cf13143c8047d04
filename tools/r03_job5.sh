#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 -k "reorder or two_plans or gate or host_output or hip_graph or fuzz or learnable" > gpurun_out/r03_pytest_sel.log 2>&1
tail -12 gpurun_out/r03_pytest_sel.log
timeout 600 python tools/sweep_spmm.py --exp dsweep > gpurun_out/r03_dsweep.log 2>&1
grep "unroll=0" gpurun_out/r03_dsweep.log
timeout 600 python bench.py --no-papers --no-cpu-baseline --steps 10 2>gpurun_out/r03_bench_quick.err | tail -1 > gpurun_out/r03_bench_quick.json
python -c "import json;j=json.load(open('gpurun_out/r03_bench_quick.json'));print(j['ms_per_step'], j['roofline']['frac'])"
