import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    pos = None; acc = collections.defaultdict(list); seen = set()
    for r in rows:
        n = r["Kernel_Name"]
        key = (r["Dispatch_Id"])
        if "k_pose_wave" in n: pos = 0; continue
        if pos is None: continue
        if "k_active_mask" in n and key not in seen: pos += 1
        seen.add(key)
        for k in ("k_skin2", "k_skin_fix"):
            if k in n: acc[(pos, k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for key in sorted(acc):
        v = acc[key][2:]
        if v: print(key, f"{sum(v)/len(v):.4g}")
