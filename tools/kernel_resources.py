#!/usr/bin/env python3
"""Compact per-kernel register/occupancy table from hipcc -Rpass-analysis=kernel-resource-usage.
usage: python tools/kernel_resources.py sgl_amd/csrc/sgl_spmm.hip [filter-substring]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-x", "hip", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in err.splitlines():
    m = re.search(r"remark:\s*(.+?): (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        if cur:
            rows.append(cur)
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
    else:
        cur[k] = v
if cur:
    rows.append(cur)
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'occ':>4s} {'LDS':>6s}")
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    n = re.sub(r"\(.*\)$", "", n).replace("void ", "")
    if flt and flt not in n:
        continue
    print(f"{n[:70]:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>6s}")
