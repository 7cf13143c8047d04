#!/bin/bash
# per-kernel time table of EVERY kernel of the library (PaSca sweep exercises normalisation, SpMM, all aggregators)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_all
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_all -o all -- python $R/tools/pasca_sweep.py > $R/gpurun_out/prof_all.log 2>&1
cd $R; find gpurun_out/prof_all -name "*_kernel_trace.csv" -delete; find gpurun_out -name "*.db" -delete
ls gpurun_out/prof_all
