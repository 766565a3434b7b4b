"""skinrt_node_*: one crowd sharded over several devices of ONE process, behind the C ABI (VERDICT r1
"multi-GPU behind the C ABI").  The GPU box has one device, so the node lists device 0 several times: every shard has
its own context, stream, rig copy and crowd, exactly as with distinct devices — only the hardware under them is shared.
Checked against the oracle run on the WHOLE crowd: shard g must hold the rows skinrt_shard_range assigns it."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import abi, api, synth
from tests.helpers import assert_equal, run_oracle

pytestmark = pytest.mark.gpu

GETTERS = {"world": "world", "palette": "palette", "key_index": "keyframe_index", "position": "positions",
           "normal": "normals", "tangent": "tangents"}


def _check_node(node, ref, instances, tag):
    covered = 0
    for g in range(node.shard_count):
        crowd, first, count = node.shard(g)
        assert (first, count) == api.shard_range(instances, g, node.shard_count)
        assert first == covered
        covered += count
        if crowd is None:
            assert count == 0
            continue
        for key, getter in GETTERS.items():
            assert_equal(getattr(crowd, getter)(), ref[key][first:first + count], f"{tag}shard {g}: {key}")
    assert covered == instances
    assert_equal(node.aabbs(), ref["aabb"], tag + "gathered aabb")


def test_node_of_three_shards_equals_the_whole_crowd(oracle32):
    spec = synth.scaled(synth.CONFIGS["c3"], verts=3000, targets=8, keys=5)
    rig = synth.make_rig(spec)
    I = 101                                  # shards of 34, 34, 33
    with api.Node([0, 0, 0], rig, I, exact_arithmetic=1) as node:
        assert node.shard_count == 3
        for frame in range(2):               # the second frame reuses every shard's state
            crowd = synth.make_crowd(synth.scaled(spec, seed=spec.seed + frame), rig, instances=I, with_root=True)
            node.set_clocks(crowd.tau, crowd.clip_id)
            node.set_morph_weights(crowd.morph_weights)
            node.set_root_transforms(crowd.root)
            node.evaluate()                  # returns with all three shards enqueued
            node.synchronize()
            _check_node(node, run_oracle(oracle32, rig, crowd), I, f"frame {frame}: ")
        # defaults again: NULL weights / NULL roots mean what they mean for one crowd
        node.set_morph_weights(None)
        node.set_root_transforms(None)
        node.evaluate(sync=True)
        crowd.morph_weights, crowd.root = None, None
        _check_node(node, run_oracle(oracle32, rig, crowd), I, "defaults: ")


def test_node_shards_a_rig_with_several_skins_and_props(oracle32):
    """The rig copy every shard uploads carries skins[] and sections[] (ABI 6): a three-skin avatar with a rigid prop,
    two shards, every output of every shard against the oracle on the whole crowd."""
    spec = synth.tiny_spec(seed=33, verts=500, joints=12, nodes=20, clips=2, keys=5, targets=4, sets=2)
    base = synth.make_rig(spec)
    rig, tmap = synth.make_multi_rig(base, seed=9)
    I = 37
    crowd = synth.make_crowd(spec, base, instances=I, with_root=True)
    crowd.morph_weights = np.ascontiguousarray(crowd.morph_weights[:, tmap])
    with api.Node([0, 0], rig, I, exact_arithmetic=1) as node:
        node.set_clocks(crowd.tau, crowd.clip_id)
        node.set_morph_weights(crowd.morph_weights)
        node.set_root_transforms(crowd.root)
        node.evaluate(sync=True)
        _check_node(node, run_oracle(oracle32, rig, crowd), I, "multi-skin: ")


def test_node_with_more_devices_than_instances(oracle32):
    spec = synth.tiny_spec(seed=12)
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=2)
    with api.Node([0, 0, 0, 0], rig, 2, exact_arithmetic=1) as node:
        node.set_clocks(crowd.tau, crowd.clip_id)
        node.set_morph_weights(crowd.morph_weights)
        node.evaluate(sync=True)
        _check_node(node, run_oracle(oracle32, rig, crowd), 2, "tiny: ")
        assert [node.shard(g)[2] for g in range(4)] == [1, 1, 0, 0]


def test_node_errors():
    spec = synth.tiny_spec(seed=12)
    rig = synth.make_rig(spec)
    with pytest.raises(api.SkinrtError) as e:
        api.Node([0, 7777], rig, 10)          # no such device: the whole node fails, nothing is left behind
    assert e.value.code == abi.E_DEVICE and "shard 1" in e.value.message
    with pytest.raises(api.SkinrtError) as e:
        api.Node([], rig, 10)
    assert e.value.code == abi.E_INVALID
    with api.Node([0, 0], rig, 10) as node:
        with pytest.raises(api.SkinrtError) as e:
            node.shard(2)
        assert e.value.code == abi.E_RANGE
        bad = np.full(10, 99, np.uint32)      # clip id out of range on every shard: the first shard reports it
        with pytest.raises(api.SkinrtError) as e:
            node.set_clocks(np.zeros(10, np.float32), bad)
        assert e.value.code == abi.E_RANGE
        with pytest.raises(api.SkinrtError) as e:
            node.evaluate(abi.STAGE_AABB)
        assert e.value.code == abi.E_INVALID
