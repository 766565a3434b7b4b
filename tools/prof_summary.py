#!/usr/bin/env python3
"""Summarise tools/prof.sh output: per-kernel avg duration and mean PMC counters per dispatch."""
import csv, glob, os, sys, collections
d = sys.argv[1]
def short(n):
    for k in ("k_skin_fix", "k_skin2", "k_skin", "k_pose_wave", "k_propagate_quad", "k_aabb_reduce", "k_sample_local", "k_propagate_lane", "k_palette", "k_fill_mw"):
        if k in n: return k
    return n[:40]
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    for r in csv.DictReader(open(f)):
        print(f"  {short(r['Name']):18s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):>12.0f} total%={r['Percentage']}")
for p in sorted(glob.glob(os.path.join(d, "pmc*"))):
    if not os.path.isdir(p): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.basename(p))
    for k, cs in acc.items():
        print("  ", k, {c: f"{sum(v)/len(v):.4g}" for c, v in cs.items()}, "dispatches", len(next(iter(cs.values()))))
