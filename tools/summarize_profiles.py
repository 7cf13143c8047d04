#!/usr/bin/env python3
"""Copy the rocprofv3 evidence of a gpurun session from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/summarize_profiles.py r01            # -> profiles/r01_*.{csv,md,json}

kernel stats: the `--kernel-trace --stats` summary CSV, trimmed to our own kernels + top entries
PMC:          one markdown table of per-launch counter means for spmm_kernel (separate --pmc passes)
traffic.json: HBM bytes per spmm launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of
              /opt/skills/guides/MI355X_MICROARCH.md section HBM (FETCH_SIZE tallies 128-B requests at 64 B)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "S1_products"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

src = os.path.join(src, f"prof_{workload}") if os.path.isdir(os.path.join(src, f"prof_{workload}")) else src
kernel_avg_ms = None
stats = glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True) or \
    glob.glob(os.path.join(src, "prof_stats", "*_kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    for r in rows:
        if "spmm_kernel" in r["Name"] and kernel_avg_ms is None:
            kernel_avg_ms = float(r["AverageNs"]) / 1e6
    with open(os.path.join(dst, f"{tag}_{workload}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            r["Name"] = r["Name"][:160]
            w.writerow(r)

means = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "pmc_counter_collection.csv"), recursive=True) or
                glob.glob(os.path.join(src, "prof_pmc_*", "pmc_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "spmm_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kname = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split(">(")[0] + ">"
            regs = (r["VGPR_Count"], r["SGPR_Count"], r["Grid_Size"], r["Workgroup_Size"])
    for c, v in agg.items():
        means[c] = (sum(v) / len(v), len(v))
if means:
    fetch = means.get("FETCH_SIZE", (0, 0))[0]
    write = means.get("WRITE_SIZE", (0, 0))[0]
    hbm = (2 * fetch + write) * 1024
    with open(os.path.join(dst, f"{tag}_{workload}_pmc_spmm_kernel.md"), "w") as f:
        f.write(f"# rocprofv3 PMC counters, {kname}, workload {workload}\n\n")
        f.write("Command per pass: `rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python bench.py "
                f"--workload {workload} --steps 3 --warmup 1 --no-cpu-baseline --no-papers --no-extras` (one pass per counter group; means over "
                "the launches)\n\n")
        f.write(f"VGPR {regs[0]}, SGPR {regs[1]}, grid {regs[2]} threads, workgroup {regs[3]}\n\n")
        f.write("| counter | mean per launch | launches |\n|---|---|---|\n")
        for c, (m, n) in means.items():
            f.write(f"| {c} | {m:.6g} | {n} |\n")
        f.write(f"\nHBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = {hbm:.4g} B "
                "(gfx950: FETCH_SIZE counts 128-B requests at 64 B; cross-check: TCC_EA0_RDREQ_sum x 128 B = "
                f"{means.get('TCC_EA0_RDREQ_sum', (0, 0))[0] * 128:.4g} B)\n")
        if "TCC_HIT_sum" in means:
            h, m_ = means["TCC_HIT_sum"][0], means["TCC_MISS_sum"][0]
            f.write(f"\nL2 hit rate = {h / (h + m_):.3%}\n")
    tfile = os.path.join(dst, "traffic.json")
    try:
        table = json.load(open(tfile))
        if "workload" in table:                       # round-1 layout: a single entry
            table = {table["workload"]: table}
    except Exception:  # noqa: BLE001
        table = {}
    table[workload] = dict(table.get(workload, {}), **{"workload": workload, "hbm_bytes_per_launch": hbm, "fetch_size_kb": fetch, "write_size_kb": write,
                       "source": f"profiles/{tag}_{workload}_pmc_spmm_kernel.md", "kernel": kname,
                       "kernel_avg_ms_rocprof": kernel_avg_ms,
                       "kernel_stats": f"profiles/{tag}_{workload}_kernel_stats.csv"})    # other keys of the entry are kept
    json.dump(table, open(tfile, "w"), indent=1)
for name in ("sweep.log", "bench.log"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
print(f"{tag} {workload}: summaries written to profiles/")
