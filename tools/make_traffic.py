#!/usr/bin/env python3
"""HBM bytes per frame of the skin stage (k_skin2 + k_skin_fix + k_skin) from the FETCH_SIZE / WRITE_SIZE passes of tools/prof.sh, corrected
as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: both counters are
in KiB; FETCH_SIZE tallies 128-B read requests at 64 B, so it is doubled; WRITE_SIZE is exact.
Prints {"<workload>:<instances>": bytes} for profiles/traffic.json (read by bench.py)."""
import csv, glob, json, os, sys
d = sys.argv[1]
args = sys.argv[2:]
wl = args[args.index("--workload") + 1] if "--workload" in args else "c3"
inst = {"c2": 4096, "c3": 10000, "c4": 100000, "c5": 1000}[wl]
if "--instances" in args: inst = int(args[args.index("--instances") + 1])
def mean(passname, counter):
    """per frame: the skin stage's kernels (k_skin2 + k_skin_fix + k_skin) added up; dispatches before the
    first pose kernel are the placement search at crowd creation, not frames"""
    rows = []
    for f in glob.glob(os.path.join(d, passname, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    first_pose = next((int(r["Dispatch_Id"]) for r in rows if "k_pose" in r["Kernel_Name"] or "k_sample" in r["Kernel_Name"]), 0)
    per = {}
    for r in rows:
        if "k_skin" in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r["Dispatch_Id"]) > first_pose:
            key = "k_skin_fix" if "k_skin_fix" in r["Kernel_Name"] else "k_skin2" if "k_skin2" in r["Kernel_Name"] else "k_skin"
            per.setdefault(key, []).append(float(r["Counter_Value"]))
    return sum(sum(v) / len(v) for v in per.values())
fetch = mean("pmc3", "FETCH_SIZE") * 1024 * 2
write = mean("pmc4", "WRITE_SIZE") * 1024
print(json.dumps({f"{wl}:{inst}": fetch + write, f"{wl}:{inst}:read": fetch, f"{wl}:{inst}:write": write}))
