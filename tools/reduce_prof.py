#!/usr/bin/env python3
"""Reduce rocprofv3 csv outputs under <dir>/prof_* to per-kernel summaries (mean/sum per kernel
name): kernel-trace durations and PMC counter values."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.replace("irn::(anonymous namespace)::", "")
    return name[:90]


for d in sorted(glob.glob(os.path.join(root, "prof_*"))):
    print("==", d)
    for f in sorted(glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)):
        base = os.path.basename(f)
        try:
            rows = list(csv.DictReader(open(f)))
        except Exception as e:
            print("  ", base, "unreadable", e)
            continue
        if not rows:
            continue
        cols = rows[0].keys()
        if "kernel_stats" in base or "stats" in base:
            print("  --", base)
            for r in rows[:14]:
                print("    ", {k: (v[:70] if isinstance(v, str) else v) for k, v in r.items()})
        elif "counter_collection" in base:
            agg = defaultdict(lambda: defaultdict(list))
            for r in rows:
                agg[short(r.get("Kernel_Name", "?"))][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0)))
            print("  --", base)
            for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
                for cn, vals in cs.items():
                    print("     %-92s %-14s n=%5d mean=%.6g sum=%.6g" % (k, cn, len(vals), sum(vals) / len(vals), sum(vals)))
        elif "kernel_trace" in base:
            agg = defaultdict(list)
            for r in rows:
                agg[short(r.get("Kernel_Name", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            print("  --", base)
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                print("     %-92s n=%5d mean=%.2f us sum=%.2f ms" % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
