#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for t in bench_nafs_pipeline bench_setup_path bench_reorder pcie_shim_bench layout_shares; do
  timeout 900 python tools/$t.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_$t.log; echo "$t exit $?"; tail -3 gpurun_out/r03_$t.log
done
