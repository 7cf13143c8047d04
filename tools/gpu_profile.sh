#!/bin/bash
# rocprofv3 evidence for the bench command: kernel-trace stats (csv) + separate PMC passes (never combined with sys-trace)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stats gpurun_out/prof_pmc_*
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r1 -- $BENCH > $R/gpurun_out/rocprof_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/prof_pmc_$T -o pmc -- $BENCH > $R/gpurun_out/rocprof_pmc_$T.log 2>&1
done
cd $R
find gpurun_out -name "*.db" -delete 2>/dev/null
find gpurun_out -name "*_kernel_trace.csv" -size +5M -delete 2>/dev/null
grep -v exit gpurun_out/bench.log | tail -1 | cut -c1-300; ls gpurun_out | head -30
