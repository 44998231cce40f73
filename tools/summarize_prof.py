#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary for profiles/."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print(f"== kernel stats ({os.path.relpath(f, out)}) ==")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  {:60s} calls={:>6s} total_ns={:>14s} avg_ns={:>12s} pct={:>6s}".format(r.get("Name", "")[:60], r.get("Calls", ""), r.get("TotalDurationNs", ""), r.get("AverageNs", ""), r.get("Percentage", "")))
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:48]
        c = r.get("Counter_Name", "")
        acc[k][c] += float(r.get("Counter_Value", 0) or 0)
        cnt[k][c] += 1
    print(f"== PMC ({os.path.relpath(f, out)}) : per-dispatch averages ==")
    for k in acc:
        if "ggemm" in k or "multi_" in k or "axpy_" in k or "slab_" in k:
            print("  " + k + ": " + ", ".join(f"{c}={acc[k][c] / cnt[k][c]:.4g} (n={cnt[k][c]})" for c in sorted(acc[k])))
