#!/bin/bash
# PMC passes (separate, kernel-trace only) over the narrow-slice SpMM: HBM-side bytes per launch for 16- and 32-float rows
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf gpurun_out/slice_pmc_*
cd /tmp
for W in 16 32; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    T=$(echo $C | tr ' ' '_' | cut -c1-30)
    SGL_SLICE_WIDTH=$W timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/slice_pmc_${W}_$T -o pmc -- python $R/tools/slice_pmc.py > $R/gpurun_out/slice_pmc_${W}_$T.log 2>&1
  done
done
cd $R
find gpurun_out -name "*.db" -delete 2>/dev/null
python - <<'PY'
import csv, glob, collections
for w in (16, 32):
    m = collections.OrderedDict()
    for f in sorted(glob.glob(f"gpurun_out/slice_pmc_{w}_*/pmc_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "spmm_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in agg.items():
            m[c] = sum(v) / len(v)
    print(f"SLICE width={w}", " ".join(f"{c}={v:.6g}" for c, v in m.items()),
          f"hbm_bytes_per_launch={(2 * m.get('FETCH_SIZE', 0) + m.get('WRITE_SIZE', 0)) * 1024:.4g}")
PY
