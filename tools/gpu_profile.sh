#!/bin/bash
# rocprofv3 evidence for a bench command: kernel-trace stats (csv) + separate PMC passes (never combined with sys-trace).
#   tools/gpu_profile.sh <workload> [extra bench flags]      e.g.  tools/gpu_profile.sh S1_products
# Outputs under gpurun_out/prof_<workload>/ ; tools/summarize_profiles.py copies the summaries into profiles/.
WL=${1:-S1_products}; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/prof_$WL
rm -rf $O; mkdir -p $O
# the papers100M-shaped section of an S1 run launches the same kernel on another graph: kept out of the profiled command
BENCH="python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-papers --no-extras $*"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- $BENCH > $O/rocprof_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$T -o pmc -- $BENCH > $O/rocprof_pmc_$T.log 2>&1
done
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*_kernel_trace.csv" -size +5M -delete 2>/dev/null
ls $O | head -30
