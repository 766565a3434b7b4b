#!/usr/bin/env python3
"""Summarise tools/prof.sh output: per-kernel duration (from the kernel trace) and mean PMC counters
per dispatch.  Dispatches of the skin kernels that precede the first pose kernel belong to the
placement search at crowd creation (all morph weights zero, see api.hip tune_output_placement):
they are not frames and are left out.  `--csv` prints the per-kernel table as CSV."""
import csv, glob, os, sys, collections
d = sys.argv[1]
as_csv = "--csv" in sys.argv
KERNELS = ("k_classify", "k_export_interleaved", "k_export_skin_uniform", "k_sample_weights", "k_skin_fix", "k_skin2", "k_skin", "k_pose_wave", "k_propagate_quad", "k_aabb_reduce", "k_sample_local",
           "k_palette", "k_fill_mw")
def short(n):
    for k in KERNELS:
        if k in n: return k
    return n[:40]
def frames_only(rows, name_key, id_key):
    """drop skin-kernel dispatches issued before the first pose dispatch"""
    rows = sorted(rows, key=lambda r: int(r[id_key]))
    first_pose = next((int(r[id_key]) for r in rows if short(r[name_key]) in ("k_pose_wave", "k_sample_local")), None)
    if first_pose is None: return rows
    return [r for r in rows if not (short(r[name_key]).startswith("k_skin") and int(r[id_key]) < first_pose)]
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = frames_only(list(csv.DictReader(open(f))), "Kernel_Name", "Dispatch_Id")
    acc = collections.defaultdict(list)
    for r in rows: acc[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in acc.values())
    if as_csv:
        print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            print(f"{k},{len(v)},{sum(v)},{sum(v)/len(v):.0f},{min(v)},{max(v)},{100.0*sum(v)/tot:.3f}")
        sys.exit(0)
    print("== kernel durations (frames only)", f)
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {k:18s} calls={len(v):5d} avg_ns={sum(v)/len(v):12.0f} min_ns={min(v):10d} max_ns={max(v):10d} total%={100.0*sum(v)/tot:.2f}")
    skin = [acc.get("k_skin2", []), acc.get("k_skin_fix", []), acc.get("k_skin", [])]
    per_frame = sum(sum(v) / len(v) for v in skin if v)
    print(f"  sum of the skin kernels' average durations (k_skin2 + k_skin_fix + k_skin): {per_frame / 1e6:.3f} ms")
    # k_skin_fix runs NEXT TO k_skin2 (side stream): the stage is the span from its first kernel's start
    # (k_classify, or k_skin2 for rigs without morph targets) to the last end of its kernels, per frame
    stage = ("k_classify", "k_skin2", "k_skin_fix", "k_skin")
    evs = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows if short(r["Kernel_Name"]) in stage))
    has_classify = any(e[2] == "k_classify" for e in evs)
    spans, cur = [], None
    for st, en, name in evs:
        if name == ("k_classify" if has_classify else "k_skin2"):
            if cur: spans.append(cur[1] - cur[0])
            cur = [st, en]
        elif cur:
            cur[1] = max(cur[1], en)
    if cur: spans.append(cur[1] - cur[0])
    if spans:
        print(f"  skin stage span per frame (first start -> last end of k_classify / k_skin2 / k_skin_fix): avg {sum(spans)/len(spans)/1e6:.3f} ms, min {min(spans)/1e6:.3f}, max {max(spans)/1e6:.3f} over {len(spans)} frames")
        timed = int(os.environ.get("PROF_TIMED", "0"))  # bench.py's --steps: the frames its own timers cover
        if 0 < timed < len(spans):
            t = spans[-timed:]
            print(f"  ... of which the last {timed} (bench.py's timed steps, after its warm-up): avg {sum(t)/len(t)/1e6:.3f} ms, min {min(t)/1e6:.3f}, max {max(t)/1e6:.3f}")
        print("  spans, ms: " + " ".join(f"{x/1e6:.2f}" for x in spans))
for p in sorted(glob.glob(os.path.join(d, "pmc*"))):
    if not os.path.isdir(p): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for r in frames_only(list(csv.DictReader(open(f))), "Kernel_Name", "Dispatch_Id"):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.basename(p))
    for k, cs in acc.items():
        print("  ", k, {c: f"{sum(v)/len(v):.4g}" for c, v in cs.items()}, "dispatches", len(next(iter(cs.values()))))
