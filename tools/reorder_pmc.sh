#!/bin/bash
# L2 hit rate of the SpMM on the shuffled community graph before / after the plan-time ordering (PMC pass, kernel-trace only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/prof_reorder
rm -rf $O; mkdir -p $O; cd /tmp
REORDER_BLOCKS=2048 timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O -o pmc -- python $R/tools/bench_reorder.py > $O/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/prof_reorder/**/pmc_counter_collection.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "spmm_kernel" in r["Kernel_Name"]]
disp = {}
for r in rows:
    disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
print("REORDER_PMC spmm dispatches:", len(ids))
for i in ids:
    h, m = disp[i].get("TCC_HIT_sum", 0), disp[i].get("TCC_MISS_sum", 0)
    print(f"REORDER_PMC dispatch {i}: L2 hit {h:.3e} miss {m:.3e} hit_rate {h / max(h + m, 1):.3f}")
PY
grep EXP $O/run.log
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +3M -delete
