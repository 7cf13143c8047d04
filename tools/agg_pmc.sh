#!/bin/bash
# PMC counters of the register-resident row kernels next to the streaming sum kernel (bench_aggregators shapes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r03_agg_pmc
rm -rf $O; mkdir -p $O
cd /tmp
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE MemUnitStalled OccupancyPercent"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  AGG_N=2449029 timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$T -o pmc -- python $R/tools/bench_aggregators.py > $O/$T.log 2>&1
done
cd $R
find $O -name "*.db" -delete 2>/dev/null
python - <<'PY'
import csv, glob, collections, os
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","r03_agg_pmc")
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for key in ("nafs_fused_kernel<32, 1, 12>","nafs_fused_kernel<32, 1, 6>","gate_fused_kernel<32, 1, 12>","gate_fused_kernel<32, 1, 6>","hop_reduce_kernel<0, 4>"):
            if key in k:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names=sorted({c for k in agg for c in agg[k]})
print("| kernel | "+" | ".join(names)+" |"); print("|---|"+"---|"*len(names))
for k in agg:
    print(f"| {k} | "+" | ".join(f"{sum(agg[k][c])/max(len(agg[k][c]),1):.4g}" if agg[k][c] else "-" for c in names)+" |")
PY
