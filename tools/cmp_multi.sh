#!/bin/bash
# C3's avatar as one skin and cut into three skins + a rigid prop (bench.py --multi-skin), same box, same crowd:
# stage times, roofline fraction and what the rig compiler made of the two.   usage: tools/cmp_multi.sh [workload]
set -u
WL=${1:-c3}
for kind in multi single; do
  flag=""; [ $kind = multi ] && flag="--multi-skin"
  timeout -k 10 300 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-extras $flag > gpurun_out/r3_${WL}_$kind.json 2> gpurun_out/r3_${WL}_$kind.err || { echo "$kind failed"; tail -5 gpurun_out/r3_${WL}_$kind.err; exit 1; }
done
python3 - <<PY
import json
for f in ("single", "multi"):
    j = json.load(open("gpurun_out/r3_${WL}_%s.json" % f))
    c = j["config"]
    print(f"{f:6s} frame {j['ms_per_step']:.3f} ms  pose {j['stage_ms']['pose']:.3f}  skin {j['stage_ms']['skin_morph']:.3f}  frac {j['roofline']['frac']:.3f}  "
          f"skins {c['skins']} sections {c['mesh_sections']} palette {c['palette_entries']} targets {c['morph_targets']}  "
          f"tiles {j['rig_compile']['tiles']} max tile joints {j['rig_compile']['max_tile_joints']} placement {j['output_placement']['strategy']}")
PY
