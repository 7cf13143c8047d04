#!/bin/bash
# Translation-cache counters of the bare gather probe (and nothing else) per table size / allocation path:
# separate rocprofv3 --pmc passes over `tools/probe_tlb.py --pmc` (one probe dispatch per case, in the printed order).
#   tools/tlb_pmc.sh [modes] [sizes]
MODES=${1:-torch,contiguous,vmm2m}; SIZES=${2:-1,57}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r03_tlb_pmc
rm -rf $O; mkdir -p $O
CMD="python $R/tools/probe_tlb.py --pmc --modes $MODES --sizes $SIZES"
cd /tmp
for C in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
         "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-48)
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$T -o pmc -- $CMD > $O/$T.log 2>&1
  grep 'TLB-PMC' $O/$T.log | head -20 > $O/cases.txt
done
cd $R
find $O -name "*.db" -delete 2>/dev/null
python $R/tools/summarize_tlb_pmc.py $O > $R/gpurun_out/r03_papers_tlb_pmc.md 2>&1
cat $R/gpurun_out/r03_papers_tlb_pmc.md | head -60
