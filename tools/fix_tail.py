"""Per frame: when k_skin_fix ends relative to k_skin2 (a tail of the side-stream kernel is time the stage pays for).
usage: python3 tools/fix_tail.py gpurun_out/<trace dir made by tools/trace.sh>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
s2 = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_skin2" in r["Kernel_Name"]])
fx = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_skin_fix" in r["Kernel_Name"]])
n = min(len(s2), len(fx))
tails, leads, f_d, s_d = [], [], [], []
for (a0, a1), (b0, b1) in list(zip(s2, fx))[-min(n, 20):]:
    tails.append((b1 - a1) / 1e6); leads.append((a0 - b0) / 1e6); f_d.append((b1 - b0) / 1e6); s_d.append((a1 - a0) / 1e6)
avg = lambda v: sum(v) / len(v)
print(f"frames {len(tails)}: k_skin2 {avg(s_d):.3f} ms, k_skin_fix {avg(f_d):.3f} ms; k_skin_fix starts {avg(leads):+.3f} ms before k_skin2 and ends {avg(tails):+.3f} ms after it "
      f"(min {min(tails):+.3f}, max {max(tails):+.3f})")
