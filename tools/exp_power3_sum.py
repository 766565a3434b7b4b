import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("k_classify", "k_active_mask", "k_skin_fix", "k_skin2", "k_pose_wave", "k_sample_nodes", "k_aabb_reduce", "fillBuffer"):
        if k in n: return k
    return n[:30]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
# group by skin stage: index since last k_pose_wave
pos = None; acc = collections.defaultdict(list)
first_frame = True; prev_end = None
for st, en, name in ev:
    if name == "k_pose_wave": pos = 0; prev_end = en; continue
    if pos is None: continue
    if name == "k_active_mask": pos += 1
    if name in ("k_skin2", "k_skin_fix", "k_classify", "k_active_mask", "k_aabb_reduce"):
        acc[(pos, name)].append(en - st)
    if name == "k_skin2":
        acc[(pos, "gap_before_skin2")].append(st - prev_end)
    prev_end = max(prev_end, en) if prev_end else en
for key in sorted(acc):
    v = acc[key][2:]
    if v: print(key, f"{sum(v)/len(v)/1e3:.1f} us  n={len(v)}")
