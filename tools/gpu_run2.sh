#!/bin/bash
# second GPU session: tests again, rocprof (csv) kernel stats + PMC passes, tuning sweep
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r1 -- $BENCH > $R/gpurun_out/rocprof_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/prof_pmc_$T -o pmc -- $BENCH > $R/gpurun_out/rocprof_pmc_$T.log 2>&1
done
rocprofv3 -L > $R/gpurun_out/counters_full.txt 2>&1; grep -oE "^\s*(Name|Counter_Name)\s*:\s*\w+|\b(TCC|TCP|SQ|GRBM|TA|TD)_[A-Z0-9_a-z]+" $R/gpurun_out/counters_full.txt | sort -u | head -600 > $R/gpurun_out/counters.txt; rm -f $R/gpurun_out/counters_full.txt
cd $R
timeout 1200 python tools/sweep_spmm.py --exp knobs,plan,colblock,relabel,pad > gpurun_out/sweep.log 2>&1
echo "sweep exit $?" >> gpurun_out/sweep.log
find gpurun_out -name "*.db" -delete 2>/dev/null
find gpurun_out -name "*_kernel_trace.csv" -size +5M -delete 2>/dev/null
du -sh gpurun_out; tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/sweep.log | tail -60
