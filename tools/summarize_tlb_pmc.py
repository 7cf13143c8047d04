#!/usr/bin/env python3
"""Markdown table from the passes of tools/tlb_pmc.sh: one row per probe dispatch (= per table size / allocation path, in
the order of cases.txt), one column per counter."""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
cases = [ln.split("case ", 1)[1].strip() for ln in open(os.path.join(src, "cases.txt")) if "case " in ln]
cols = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(src, "**", "pmc_counter_collection.csv"), recursive=True)):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "probe_gather_kernel" not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], collections.OrderedDict())
        did = int(r["Dispatch_Id"])
        per[r["Counter_Name"]][did] = per[r["Counter_Name"]].get(did, 0.0) + float(r["Counter_Value"])
    for c, by in per.items():
        cols[c] = [v for _, v in sorted(by.items())]
print("# Translation-cache counters of the bare row-gather probe (96 Mi gathers of 512-byte rows, 16 in flight per lane)\n")
print("`tools/tlb_pmc.sh`: separate `rocprofv3 --kernel-trace --pmc` passes over `tools/probe_tlb.py --pmc`; one probe dispatch per case.\n")
names = list(cols)
print("| case | " + " | ".join(names) + " |")
print("|---|" + "---|" * len(names))
for i, case in enumerate(cases):
    print(f"| {case} | " + " | ".join(f"{cols[c][i]:.6g}" if i < len(cols[c]) else "-" for c in names) + " |")
if "TCP_UTCL1_TRANSLATION_HIT_sum" in cols and "TCP_UTCL1_TRANSLATION_MISS_sum" in cols:
    print("\n| case | UTCL1 miss rate | misses per gathered row |\n|---|---|---|")
    for i, case in enumerate(cases):
        h, m = cols["TCP_UTCL1_TRANSLATION_HIT_sum"][i], cols["TCP_UTCL1_TRANSLATION_MISS_sum"][i]
        print(f"| {case} | {m / max(h + m, 1):.4f} | {m / (96 << 20):.3f} |")
