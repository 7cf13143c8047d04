#!/bin/bash
# PMC counters of the register-resident row kernels next to the streaming sum kernel (tools/bench_aggregators.py shapes).
#   tools/agg_pmc.sh TAG "dxH[,dxH...]" "kernel key;kernel key;..." [ENV=VALUE ...]
# e.g. tools/agg_pmc.sh r04_agg_pmc_narrow 147x6 "gate_fused_kernel<8, 5, 6>;nafs_fused_kernel<8, 5, 6>;hop_reduce_kernel<0, 4>" AGG_NARROW=1
# Every counter group is its own rocprofv3 pass (--kernel-trace + --pmc only); the table is written to gpurun_out/TAG.md.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; TAG=$1; SHAPES=$2; KEYS=$3; shift 3
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" \
         "GRBM_GUI_ACTIVE MemUnitStalled OccupancyPercent" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  AGG_N=${AGG_N:-2449029} AGG_SHAPES=$SHAPES timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$T -o pmc -- python $R/tools/bench_aggregators.py > $O/$T.log 2>&1
done
cd $R
find $O -name "*.db" -delete 2>/dev/null
AGG_PMC_DIR=$O AGG_PMC_KEYS="$KEYS" python - > $R/gpurun_out/$TAG.md <<'PY'
import csv, glob, collections, os
O = os.environ["AGG_PMC_DIR"]
keys = [k for k in os.environ["AGG_PMC_KEYS"].split(";") if k]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
res = {}
for f in glob.glob(O + "/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in keys:
            if key in k:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                res[key] = (r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"))
names = sorted({c for k in agg for c in agg[k]})
print("| kernel | VGPR | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k in keys:
    if k in agg:
        print(f"| {k} | {res[k][0]} | " + " | ".join(f"{sum(agg[k][c]) / max(len(agg[k][c]), 1):.4g}" if agg[k][c] else "-" for c in names) + " |")
PY
cat $R/gpurun_out/$TAG.md
