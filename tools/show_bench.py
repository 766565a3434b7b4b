#!/usr/bin/env python3
"""One line per bench.py JSON file: frame, stages, roofline fraction, placement, extras.  usage: tools/show_bench.py a.json b.json ..."""
import json
import sys

for path in sys.argv[1:]:
    try:
        j = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:  # an empty or truncated file
        print(f"{path}: unreadable ({e})")
        continue
    st, rf, pl = j["stage_ms"], j["roofline"], j.get("output_placement", {})
    extra = ""
    if j.get("frame_loop"):
        extra += f"  loop adv {j['frame_loop']['ms_per_step_advancing_clocks']:.3f} / same {j['frame_loop']['ms_per_step_evaluate_only']:.3f}"
    if st.get("skin_morph_plain_placement"):
        extra += f"  plain-placement skin {st['skin_morph_plain_placement']:.3f}"
    if j.get("cpu_baseline"):
        extra += f"  cpu {j['cpu_baseline']['value']:.3g} ({j['cpu_baseline']['cores']}) / {j['cpu_baseline_mt']['value']:.3g} ({j['cpu_baseline_mt']['cores']})"
    print(f"{path}: {j['config']['workload']} x{j['config']['instances_total']}  value {j['value']:.4g} {j['unit']}  frame {j['ms_per_step']:.3f} ms  "
          f"pose {st['pose']:.3f} skin {st['skin_morph']:.3f} aabb {st['aabb_reduce']:.3f}  frac {rf['frac']:.3f} ({rf['achieved']:.0f} GB/s)  "
          f"traffic {rf.get('traffic')}  placement {pl.get('strategy')} gain {pl.get('spread_gain', 0):.2f} filler {pl.get('filler_GiB', 0):.0f} GiB{extra}")
