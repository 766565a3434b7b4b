#!/usr/bin/env python3
"""HBM bytes per k_skin launch from the FETCH_SIZE / WRITE_SIZE passes of tools/prof.sh, corrected
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
    v = []
    for f in glob.glob(os.path.join(d, passname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_skin" in r["Kernel_Name"] and r["Counter_Name"] == counter: v.append(float(r["Counter_Value"]))
    return sum(v) / len(v)
fetch = mean("pmc3", "FETCH_SIZE") * 1024 * 2
write = mean("pmc4", "WRITE_SIZE") * 1024
print(json.dumps({f"{wl}:{inst}": fetch + write, f"{wl}:{inst}:read": fetch, f"{wl}:{inst}:write": write}))
