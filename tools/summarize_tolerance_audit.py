#!/usr/bin/env python3
"""Summarise a parity audit (SGL_PARITY_AUDIT=<file> python -m pytest tests -m gpu): every oracle.parity_report call that was
given a relaxation of the SURVEY 8(c) criterion (scale=|A||X| in the row norm, or rowwise=False) is recorded with the verdict of
the UNRELAXED criterion.  Prints a markdown table: per test, how many comparisons used a relaxation and how many NEEDED it
(unrelaxed row criterion > tol), with the worst unrelaxed row error and the output widths involved.
usage: python tools/summarize_tolerance_audit.py gpurun_out/r05_tolerance_audit.jsonl > profiles/r05_tolerance_audit.md"""
import collections
import json
import sys

rows = [json.loads(line) for line in open(sys.argv[1]) if line.strip()]
by = collections.OrderedDict()
for r in rows:
    test = r["test"].split(" ")[0].replace("tests/test_gpu_parity.py::", "").replace("tests/", "")
    e = by.setdefault((test, r["relaxation"].strip()), {"n": 0, "need": 0, "worst": 0.0, "widths": set(), "tol": r["tol"], "used": 0.0})
    e["n"] += 1
    e["used"] = max(e["used"], r["row_l2_rel_used"])
    if r["needs_relaxation"]:
        e["need"] += 1
        e["worst"] = max(e["worst"], r["row_l2_rel_unrelaxed"])
        e["widths"].add(r["shape"][-1] if r["shape"] else 1)
total, need = sum(e["n"] for e in by.values()), sum(e["need"] for e in by.values())
print("# Tolerance audit: which comparisons need the relaxed row criterion\n")
print(f"{total} comparisons were made with a relaxation (`scale=` = condition-aware row norm, or `rowwise=False`); "
      f"**{need} of them would fail the unrelaxed SURVEY 8(c) row criterion** (`max_rows |d_row|_2 / |ref_row|_2 <= tol`); the global "
      "max-norm and the allclose criteria are never relaxed.  Strict-order (bit-exact) comparisons do not appear: they use array_equal.\n")
print("| test | relaxation | comparisons | need it | worst unrelaxed row error | tol | widths (columns) of those that need it |")
print("|---|---|---|---|---|---|---|")
for (test, relax), e in by.items():
    w = ", ".join(str(x) for x in sorted(e["widths"])) if e["widths"] else "-"
    print(f"| `{test}` | {relax} | {e['n']} | {e['need']} | {e['worst']:.2e} | {e['tol']:g} | {w} |")
