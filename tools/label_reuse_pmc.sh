#!/bin/bash
# rocprofv3 PMC passes over examples/gamlp_label_reuse_synthetic.py: what the 48-column launches of the column delta move
# (config.delta_propagate) next to the 147-column launches of the first call.   -> gpurun_out/r06_label_reuse_S2_pmc.md
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/prof_label_reuse_pmc
rm -rf $O; mkdir -p $O
CMD="python $R/examples/gamlp_label_reuse_synthetic.py --epochs 1"
cd /tmp
for C in "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$T -o pmc -- $CMD > $O/rocprof_pmc_$T.log 2>&1
done
cd $R
find $O -name "*.db" -delete 2>/dev/null
python - <<'PY' > gpurun_out/r06_label_reuse_S2_pmc.md
import csv, glob, os, collections, re
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_label_reuse_pmc"
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(O + "/pmc_*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(spmm_kernel<[^>]*>|col_signature_kernel)", row["Kernel_Name"])
        if m:
            k = (m.group(1), row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
nnz = 126167309
print("| kernel | counter | mean per launch | launches | per non-zero |\n|---|---|---|---|---|")
for (kn, cn), (s, c) in sorted(acc.items()):
    print(f"| `{kn}` | {cn} | {s / c:.5e} | {c} | {s / c / nnz:.3f} |" if "spmm" in kn else f"| `{kn}` | {cn} | {s / c:.5e} | {c} | |")
PY
cat gpurun_out/r06_label_reuse_S2_pmc.md
