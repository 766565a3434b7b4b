"""Output-array placement (placement.hip) under conditions the steering does not control (VERDICT r2 "weak 4",
ADVICE r2): a second crowd while the first lives, a crowd created after a large foreign allocation, create / destroy /
create, a capped filler.  Placement must never show in the results (bit-identical to output_placement = 1), and
skinrt_placement_info.strategy must say what the MEASUREMENT found, not what was intended: 1 exactly when
spread_gain >= 1.15, 2 when the search found nothing, 3 when it ran out of room.  The gains themselves are
printed (and written to gpurun_out/r3_placement.json): what the allocator does in each situation is a property of the
box, recorded in DESIGN.md, not asserted."""
import json
import os

import numpy as np
import pytest

from renderer_amd import abi, api, synth
from tests.helpers import DeviceBuffers, assert_equal

pytestmark = pytest.mark.gpu

ROWS = []


def _run(ctx, rig_h, crowd_state, n):
    c = api.Crowd(ctx, rig_h, n)
    c.set_clocks(crowd_state.tau, crowd_state.clip_id)
    c.set_morph_weights(crowd_state.morph_weights)
    c.evaluate(abi.STAGE_ALL, sync=True)
    return c


def _check(tag, c, ref_bounds, ref_pos, max_filler=None):
    p = c.placement
    ROWS.append(dict(case=tag, **p))
    print(f"[placement] {tag:34s} strategy {p['strategy']:24s} filler {p['filler_GiB']:6.1f} GiB  spread_gain {p['spread_gain']:.3f}  probe {p['probe_GBps']:.0f} GB/s")
    assert p["strategy"] in ("spread", "searched-none-found", "search-out-of-room")
    assert (p["strategy"] == "spread") == (p["spread_gain"] >= 1.15)
    if max_filler is not None:
        assert p["filler_GiB"] <= max_filler
    assert_equal(c.aabbs().view(np.uint32), ref_bounds.view(np.uint32), f"{tag}: bounds of all instances")
    for i, want in ref_pos.items():
        assert_equal(c.positions(i, 1).view(np.uint32), want.view(np.uint32), f"{tag}: positions of instance {i}")


def test_placement_never_shows_in_results_and_reports_what_was_measured():
    spec = synth.scaled(synth.CONFIGS["c3"], instances=2000)   # 3 x 1.44 GB of outputs: large enough to be steered
    rig = synth.make_rig(spec)
    n = spec.instances
    crowd = synth.make_crowd(spec, rig, instances=n)
    samples = [0, 79, 80, n - 1]
    # the reference: plain placement
    with api.Context(output_placement=1) as ctx:
        r = api.Rig(ctx, rig)
        c = _run(ctx, r, crowd, n)
        assert c.placement["strategy"] == "plain" and c.placement["spread_gain"] == 0
        ref_bounds = c.aabbs().copy()
        ref_pos = {i: c.positions(i, 1).copy() for i in samples}
        c.close(); r.close()
    dev = DeviceBuffers()
    with api.Context() as ctx:
        r = api.Rig(ctx, rig)
        a = _run(ctx, r, crowd, n)
        _check("first crowd of the process", a, ref_bounds, ref_pos)
        b = _run(ctx, r, crowd, n)
        _check("second crowd, first alive", b, ref_bounds, ref_pos)
        a.close(); b.close()
        c = _run(ctx, r, crowd, n)
        _check("create / destroy / create", c, ref_bounds, ref_pos)
        c.close()
        # a foreign holder of 120 GiB (another library's buffers: the renderer's textures, a second engine)
        big = dev.empty(120 << 30)
        d = _run(ctx, r, crowd, n)
        _check("after a foreign 120 GiB allocation", d, ref_bounds, ref_pos)
        d.close()
        dev.close()
        e = _run(ctx, r, crowd, n)
        _check("after the foreign allocation is gone", e, ref_bounds, ref_pos)
        e.close(); r.close()
    with api.Context(placement_max_filler_gib=1) as ctx:
        r = api.Rig(ctx, rig)
        c = _run(ctx, r, crowd, n)
        _check("filler capped at 1 GiB", c, ref_bounds, ref_pos, max_filler=1.0)
        c.close(); r.close()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r3_placement.json", "w") as f:
        json.dump(ROWS, f, indent=1)
