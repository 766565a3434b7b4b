"""Known-answer tests of the CPU oracle (the reference holds no tests or fixtures
for this path, so parity is UNPINNED; these hand-derived answers pin the oracle to the reference's
formulas instead).  Each case names the reference lines whose arithmetic it restates.  CPU only."""
import numpy as np
import pytest

from renderer_amd import abi
from renderer_amd.abi import Channel, MorphTarget, RigData

S = np.float32(np.sqrt(0.5))


def mat(cols):
    """column-major [16] from 4 columns"""
    return np.array(cols, dtype=np.float64).reshape(16)


def rig_of(parent, t=None, q=None, s=None, joint_node=(0,), ibm=None, pos=((0, 0, 0),), nrm=None, tan=None,
           joints=None, weights=None, clips=(), targets=(), default_weights=None, matrix=None, has_matrix=None):
    n = len(parent)
    v = len(pos)
    t = np.zeros((n, 3)) if t is None else t
    q = np.tile([0, 0, 0, 1.0], (n, 1)) if q is None else q
    s = np.ones((n, 3)) if s is None else s
    nrm = np.tile([0, 0, 1.0], (v, 1)) if nrm is None else nrm
    tan = np.tile([1.0, 0, 0], (v, 1)) if tan is None else tan
    joints = [np.zeros((v, 4), np.uint16)] if joints is None else joints
    weights = [np.tile([1.0, 0, 0, 0], (v, 1))] if weights is None else weights
    return RigData(parent=np.array(parent), translation=t, rotation=q, scale=s, joint_node=np.array(joint_node),
                   inverse_bind=ibm, position=np.array(pos, dtype=np.float32), normal=nrm, tangent=tan, joints=joints,
                   weights=weights, clips=[list(c) for c in clips], targets=list(targets),
                   default_weights=default_weights, matrix=matrix, has_matrix=has_matrix)


# ---- a2 / a3: Interpolate (renderer-asset/src/animation.rs:51-63) ---------------------------------------------

def test_linear_known_answer(oracle32):
    assert oracle32.linear([1, 2, 3], [3, 2, 1], 0.25).tolist() == [1.5, 2.0, 2.5]


def test_linear_is_not_slerp(oracle32):
    # rotations take the same component-wise formula: no normalisation, no neighbourhood flip
    a, b = [0, 0, 0, 1.0], [0, 0, 0, -1.0]
    assert oracle32.linear(a, b, 0.5).tolist() == [0, 0, 0, 0]


def test_cubic_spline_quirk_and_spec(oracle32):
    # vk=1, bk=2 (out tangent of k), vk1=3, ak1=4 (in tangent of k+1), t=0.5, td=2
    # spec Hermite: .5*1 + .125*2*2 + .5*3 - .125*2*4 = 1.5
    # as written (animation.rs:57, t2 = t^3): .875*1 + .375*2*2 + .125*3 + 0 = 2.75
    assert oracle32.cubic_spline([1], [2], [3], [4], 0.5, 2.0, quirk=0).tolist() == [1.5]
    assert oracle32.cubic_spline([1], [2], [3], [4], 0.5, 2.0, quirk=1).tolist() == [2.75]


def test_cubic_spline_endpoints(oracle32):
    for quirk in (0, 1):
        assert oracle32.cubic_spline([1, 5], [2, 2], [3, 7], [4, 4], 0.0, 2.0, quirk).tolist() == [1, 5]
        assert oracle32.cubic_spline([1, 5], [2, 2], [3, 7], [4, 4], 1.0, 2.0, quirk).tolist() == [3, 7]


# ---- a4: keyframe search (build rule) -----------------------------------------------------------------------------

@pytest.mark.parametrize("tau,k,u", [(-1.0, 0, 0.0), (0.0, 0, 0.0), (0.5, 0, 0.5), (1.0, 1, 0.0), (3.0, 2, 0.5),
                                     (4.0, 3, 0.0), (9.0, 3, 0.0)])
def test_find_key(oracle32, tau, k, u):
    kk, uu, _ = oracle32.find_key([0.0, 1.0, 2.0, 4.0], tau)
    assert (kk, uu) == (k, u)


def test_find_key_single_key(oracle32):
    assert oracle32.find_key([0.7], 123.0)[0] == 0
    assert oracle32.find_key([0.7], -5.0)[0] == 0


# ---- a5: TRS -> Mat4 (renderer-asset/src/node.rs:47-53, glam from_quat) --------------------------------------

def test_trs_identity(oracle32):
    assert oracle32.trs_to_mat4([0, 0, 0], [0, 0, 0, 1], [1, 1, 1]).tolist() == np.eye(4).reshape(16).tolist()


def test_trs_half_turn_exact(oracle32):
    # 180 degrees about Z, q = (0,0,1,0): every product is exact in fp32
    m = oracle32.trs_to_mat4([1, 2, 3], [0, 0, 1, 0], [2, 3, 4]).reshape(4, 4)  # [col][row]
    assert m[0].tolist() == [-2, 0, 0, 0]
    assert m[1].tolist() == [0, -3, 0, 0]
    assert m[2].tolist() == [0, 0, 4, 0]
    assert m[3].tolist() == [1, 2, 3, 1]


def test_trs_quarter_turn(oracle32):
    # 90 degrees about Z: x axis -> +y, y axis -> -x (column-major, columns are the images of the axes)
    m = oracle32.trs_to_mat4([0, 0, 0], [0, 0, S, S], [1, 1, 1]).reshape(4, 4)
    np.testing.assert_allclose(m[0], [0, 1, 0, 0], atol=1e-7)
    np.testing.assert_allclose(m[1], [-1, 0, 0, 0], atol=1e-7)
    np.testing.assert_allclose(m[2], [0, 0, 1, 0], atol=0)


def test_trs_uses_quaternion_unnormalised(oracle32):
    # glam only debug-asserts the length; a length-2 quaternion scales the off-identity terms by 4
    m = oracle32.trs_to_mat4([0, 0, 0], [0, 0, 0, 2], [1, 1, 1]).reshape(4, 4)
    assert m[0].tolist() == [1, 0, 0, 0]   # 1 - (yy + zz) with y = z = 0
    m = oracle32.trs_to_mat4([0, 0, 0], [0, 0, 2, 0], [1, 1, 1]).reshape(4, 4)
    assert m[0].tolist() == [1 - 8, 0, 0, 0]  # zz = z * (z + z) = 8


def test_decompose_round_trip(oracle32):
    m = oracle32.trs_to_mat4([1, 2, 3], [0, 0, 1, 0], [2, 3, 4])
    t, q, s = oracle32.decompose(m)
    assert t.tolist() == [1, 2, 3] and s.tolist() == [2, 3, 4]
    # q and -q are the same rotation
    assert np.abs(q).tolist() == [0, 0, 1, 0]


# ---- a7: world propagation, glam Mat4 * Mat4 ------------------------------------------------------------------------

def test_mat4_mul_translations_add(oracle32):
    a = oracle32.trs_to_mat4([1, 2, 3], [0, 0, 0, 1], [1, 1, 1])
    b = oracle32.trs_to_mat4([10, 20, 30], [0, 0, 0, 1], [1, 1, 1])
    assert oracle32.mat4_mul(a, b).reshape(4, 4)[3].tolist() == [11, 22, 33, 1]


def test_two_node_chain_world(oracle32):
    # parent: half turn about Z at (5,0,0); child: translation (1,0,0) -> world child origin = (4,0,0)
    rig = rig_of(parent=[-1, 0], t=[[5, 0, 0], [1, 0, 0]], q=[[0, 0, 1, 0], [0, 0, 0, 1]], joint_node=(1,))
    r = oracle32.evaluate(rig, tau=[0.0], want=("world",))
    w = r.world.reshape(1, 2, 4, 4)
    assert w[0, 0, 3].tolist() == [5, 0, 0, 1]
    assert w[0, 1, 3].tolist() == [4, 0, 0, 1]
    assert w[0, 1, 0].tolist() == [-1, 0, 0, 0]


def test_dfs_preorder_is_required_and_used(oracle32):
    # nodes 0 -> (1 -> 2), 3 under 0: a pre-order walk; world[3] must use world[0] only
    rig = rig_of(parent=[-1, 0, 1, 0], t=[[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 5]], joint_node=(3,))
    r = oracle32.evaluate(rig, tau=[0.0], want=("world",))
    w = r.world.reshape(4, 4, 4)
    assert w[2, 3].tolist() == [1, 1, 1, 1]
    assert w[3, 3].tolist() == [1, 0, 5, 1]


# ---- a8 / a9: joint palette (renderer/src/renderer/uniform/skin.rs:12-50) -------------------------------------

def test_identity_pose_palette_is_identity(oracle32):
    rig = rig_of(parent=[-1, 0], joint_node=(0, 1), ibm=None)
    r = oracle32.evaluate(rig, tau=[0.0], want=("palette",))
    e = r.palette[0]
    for j in range(2):
        assert e[j, :16].tolist() == np.eye(4).reshape(16).tolist()
        assert e[j, 16:].reshape(3, 4).tolist() == [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]]


def test_skin_matrix_layout_112_bytes(oracle32):
    # SkinMatrix { transform: [[f32;4];4], normal: [[f32;4];3] } = 28 floats; columns first, pad lane = 0
    rig = rig_of(parent=[-1], t=[[7, 8, 9]], s=[[2, 4, 8]], joint_node=(0,))
    r = oracle32.evaluate(rig, tau=[0.0], want=("palette",))
    e = r.palette[0, 0]
    assert e.shape == (28,) and e.astype(np.float32).nbytes == 112
    assert e[:16].reshape(4, 4).tolist() == [[2, 0, 0, 0], [0, 4, 0, 0], [0, 0, 8, 0], [7, 8, 9, 1]]
    assert e[16:].reshape(3, 4).tolist() == [[0.5, 0, 0, 0], [0, 0.25, 0, 0], [0, 0, 0.125, 0]]


def test_joint_matrix_is_world_times_ibm(oracle32):
    ibm = oracle32.trs_to_mat4([-1, -2, -3], [0, 0, 0, 1], [1, 1, 1]).astype(np.float32)[None]
    rig = rig_of(parent=[-1], t=[[1, 2, 3]], joint_node=(0,), ibm=ibm)
    r = oracle32.evaluate(rig, tau=[0.0], want=("palette",))
    assert r.palette[0, 0, :16].tolist() == np.eye(4).reshape(16).tolist()


def test_inverse_transpose_non_uniform_scale(oracle32):
    m = mat([[2, 0, 0, 0], [0, 4, 0, 0], [0, 0, 8, 0], [3, 3, 3, 1]])
    n = oracle32.normal_transform(m).reshape(3, 4)
    assert n.tolist() == [[0.5, 0, 0, 0], [0, 0.25, 0, 0], [0, 0, 0.125, 0]]


def test_mat4_inverse_general(oracle32, oracle64):
    rng = np.random.default_rng(5)
    m = oracle64.trs_to_mat4(rng.normal(size=3), [0.1, 0.2, 0.3, 0.9], [1.5, 0.7, 1.1])
    inv = oracle64.mat4_inverse(m)
    np.testing.assert_allclose(oracle64.mat4_mul(m, inv).reshape(4, 4), np.eye(4), atol=1e-12)
    inv32 = oracle32.mat4_inverse(m)
    np.testing.assert_allclose(inv32, inv, rtol=1e-5, atol=1e-6)


# ---- a10: linear blend skinning (renderer/src/shader/shader.wgsl:127-141,154-165) -------------------------------

def test_rigid_single_joint_skin_is_matrix_times_point(oracle32):
    rig = rig_of(parent=[-1], t=[[1, 2, 3]], q=[[0, 0, 1, 0]], s=[[2, 2, 2]], joint_node=(0,),
                 pos=[(1, 0, 0), (0, 1, 0), (0.5, 0.25, -1)], nrm=[(1, 0, 0)] * 3, tan=[(0, 1, 0)] * 3)
    r = oracle32.evaluate(rig, tau=[0.0], want=("position", "normal", "tangent", "aabb"))
    assert r.position[0].tolist() == [[-1, 2, 3], [1, 0, 3], [0, 1.5, 1]]
    # normal matrix = inverse transpose = R / 2 ; outputs are NOT normalised in the main shader variants
    assert r.normal[0].tolist() == [[-0.5, 0, 0]] * 3
    assert r.tangent[0].tolist() == [[0, -0.5, 0]] * 3
    assert r.aabb[0].tolist() == [[-1, 0, 1], [1, 2, 3]]


def test_outline_variant_normalises(oracle32):
    rig = rig_of(parent=[-1], s=[[2, 2, 2]], joint_node=(0,), pos=[(1, 0, 0)], nrm=[(1, 0, 0)], tan=[(0, 1, 0)])
    r = oracle32.evaluate(rig, tau=[0.0], normalize_outputs=1, want=("normal", "tangent"))
    assert r.normal[0, 0].tolist() == [1, 0, 0] and r.tangent[0, 0].tolist() == [0, 1, 0]


def test_two_joint_blend_weights_used_as_given(oracle32):
    # joints at x = 0 and x = 10; weights (0.25, 0.5): no renormalisation -> S = .25*M0 + .5*M1
    rig = rig_of(parent=[-1, -1], t=[[0, 0, 0], [10, 0, 0]], joint_node=(0, 1), pos=[(1, 1, 1)],
                 joints=[np.array([[0, 1, 0, 0]], np.uint16)], weights=[np.array([[0.25, 0.5, 0, 0]], np.float32)])
    r = oracle32.evaluate(rig, tau=[0.0], want=("position",))
    # S = diag(.75) with translation 5, w row = .75: p' = .75 * p + 5 (xyz of S * (p, 1))
    assert r.position[0, 0].tolist() == [5.75, 0.75, 0.75]


def test_second_weight_set_adds(oracle32):
    rig = rig_of(parent=[-1, -1], t=[[0, 0, 0], [4, 0, 0]], joint_node=(0, 1), pos=[(0, 0, 0)],
                 joints=[np.array([[0, 0, 0, 0]], np.uint16), np.array([[1, 0, 0, 0]], np.uint16)],
                 weights=[np.array([[0.5, 0, 0, 0]], np.float32), np.array([[0.5, 0, 0, 0]], np.float32)])
    r = oracle32.evaluate(rig, tau=[0.0], want=("position",))
    assert r.position[0, 0].tolist() == [2, 0, 0]


# ---- a13: morph blend before skinning (build rule; data model renderer-asset/src/primitive.rs:33-38) ---------------

def test_morph_blend_order_and_missing_attributes(oracle32):
    t0 = MorphTarget(position=[[1, 0, 0], [0, 0, 0]], normal=[[0, 1, 0], [0, 0, 0]], tangent=None)
    t1 = MorphTarget(position=[[0, 2, 0]], normal=None, tangent=[[0, 0, 1]], indices=[1])  # compacted: vertex 1 only
    rig = rig_of(parent=[-1], joint_node=(0,), pos=[(0, 0, 0), (1, 1, 1)], targets=[t0, t1])
    mw = np.array([[0.5, 2.0], [0.0, 0.0]], np.float32)
    r = oracle32.evaluate(rig, tau=[0.0, 0.0], morph_weights=mw, want=("position", "normal", "tangent"))
    assert r.position[0].tolist() == [[0.5, 0, 0], [1, 5, 1]]
    assert r.normal[0].tolist() == [[0, 0.5, 1], [0, 0, 1]]
    assert r.tangent[0].tolist() == [[1, 0, 0], [1, 0, 2]]
    assert r.position[1].tolist() == [[0, 0, 0], [1, 1, 1]]


def test_default_morph_weights_apply_when_none_given(oracle32):
    t0 = MorphTarget(position=[[1, 0, 0]])
    rig = rig_of(parent=[-1], joint_node=(0,), pos=[(0, 0, 0)], targets=[t0], default_weights=[0.25])
    r = oracle32.evaluate(rig, tau=[0.0], want=("position",))
    assert r.position[0, 0].tolist() == [0.25, 0, 0]


# ---- a1 / a6: channels overwrite one of T / R / S; STEP / LINEAR / CUBICSPLINE ----------------------------------

def test_channel_sampling_paths(oracle32):
    lin = Channel(0, abi.PATH_TRANSLATION, abi.INTERP_LINEAR, [0, 2], [[0, 0, 0], [4, 8, 12]])
    step = Channel(1, abi.PATH_SCALE, abi.INTERP_STEP, [0, 1], [[1, 1, 1], [3, 3, 3]])
    cub = Channel(2, abi.PATH_TRANSLATION, abi.INTERP_CUBICSPLINE, [0, 2],
                  [[[9, 9, 9], [1, 1, 1], [2, 2, 2]], [[4, 4, 4], [3, 3, 3], [9, 9, 9]]])  # (in, value, out) per key
    rig = rig_of(parent=[-1, -1, -1], joint_node=(0, 1, 2), clips=[[lin, step, cub]])
    r = oracle32.evaluate(rig, tau=[0.5, 1.0], clip_id=[0, 0], cubic_quirk=0, want=("world", "key_index"))
    w = r.world.reshape(2, 3, 4, 4)
    assert w[0, 0, 3].tolist() == [1, 2, 3, 1]          # linear, u = .25
    assert w[0, 1, 0, 0] == 1 and w[1, 1, 0, 0] == 3    # step: value[k]
    assert w[1, 2, 3, :3].tolist() == [1.5, 1.5, 1.5]   # spec Hermite of the known-answer case above
    assert r.key_index[0].tolist() == [0, 0, 0] and r.key_index[1].tolist() == [0, 1, 0]
    rq = oracle32.evaluate(rig, tau=[1.0], clip_id=[0], cubic_quirk=1, want=("world",))
    assert rq.world.reshape(3, 4, 4)[2, 3, :3].tolist() == [2.75, 2.75, 2.75]


def test_weights_channel_known_answers(oracle32):
    """glTF "weights" channels (SKINRT_PATH_WEIGHTS; the reference's loader: todo!(), loader/gltf/mod.rs:1312):
    the channel's sample replaces the instance's morph weights, component by component with Interpolate::linear /
    cubic_spline (animation.rs:51-63) and the a4 key rule; instances on a clip without one keep the host's weights."""
    t0 = MorphTarget(position=[[1, 0, 0]])
    t1 = MorphTarget(position=[[0, 2, 0]])
    lin = Channel(0, abi.PATH_WEIGHTS, abi.INTERP_LINEAR, [0, 2], [[0, 1], [4, 0]])
    step = Channel(0, abi.PATH_WEIGHTS, abi.INTERP_STEP, [1, 2], [[1, 1], [3, 0.5]])
    # per key (in, value, out), each [M]: component 0 is the known-answer Hermite case (vk 1, bk 2, vk1 3, ak1 4, td 2)
    cub = Channel(0, abi.PATH_WEIGHTS, abi.INTERP_CUBICSPLINE, [0, 2],
                  [[[9, 9], [1, 0], [2, 0]], [[4, 0], [3, 0], [9, 9]]])
    none = Channel(0, abi.PATH_TRANSLATION, abi.INTERP_STEP, [0], [[0, 0, 5]])
    rig = rig_of(parent=[-1], joint_node=(0,), pos=[(0, 0, 0)], targets=[t0, t1], clips=[[lin], [step], [cub], [none]],
                 default_weights=[0.25, 0.25])
    host = np.tile(np.array([[10.0, 10.0]], np.float32), (7, 1))
    r = oracle32.evaluate(rig, tau=[0.5, 0.0, 1.5, 2.5, 1.0, 1.0, 3.0], clip_id=[0, 1, 1, 1, 2, 3, 0], morph_weights=host,
                          cubic_quirk=0, want=("position", "key_index"))
    p = r.position[:, 0]
    assert p[0].tolist() == [1.0, 1.5, 0.0]      # linear u = .25: w = (0*.75 + 4*.25, 1*.75 + 0*.25)
    assert p[1].tolist() == [1.0, 2.0, 0.0]      # before the first key: clamped to key 0
    assert p[2].tolist() == [1.0, 2.0, 0.0]      # step inside the span: value[k]
    assert p[3].tolist() == [3.0, 1.0, 0.0]      # after the last key: clamped to key 1
    assert p[4].tolist() == [1.5, 0.0, 0.0]      # spec Hermite, u = .5
    assert p[5].tolist() == [10.0, 20.0, 5.0]    # clip without a weights channel: the host's weights
    assert p[6].tolist() == [4.0, 0.0, 0.0]      # linear, past the end
    assert r.key_index[:, 0].tolist() == [0, 0, 0, 1, 0, 0, 1]
    rq = oracle32.evaluate(rig, tau=[1.0], clip_id=[2], cubic_quirk=1, want=("position",))
    assert rq.position[0, 0].tolist() == [2.75, 0.0, 0.0]   # t2 = t^3 as written (animation.rs:57)
    # rest pose (clip id out of range): no channel applies, host weights / defaults
    r0 = oracle32.evaluate(rig, tau=[0.5], clip_id=[0xFFFFFFFF], want=("position",))
    assert r0.position[0, 0].tolist() == [0.25, 0.5, 0.0]


def test_weights_channel_leaves_a_matrix_node_a_matrix(oracle32):
    # a sheared matrix cannot survive to_scale_rotation_translation: if the weights channel counted as "animated"
    # (node.rs:73-87) the node's world matrix would change
    m = np.eye(4); m[0, 1] = 0.5; m[:3, 3] = [1, 2, 3]
    cols = np.ascontiguousarray(m.T).reshape(1, 16)
    t0 = MorphTarget(position=[[1, 0, 0]])
    ch = Channel(0, abi.PATH_WEIGHTS, abi.INTERP_STEP, [0], [[2.0]])
    rig = rig_of(parent=[-1], joint_node=(0,), pos=[(0, 0, 0)], targets=[t0], clips=[[ch]], matrix=cols, has_matrix=np.array([1], np.uint8))
    r = oracle32.evaluate(rig, tau=[0.0], clip_id=[0], want=("world", "position"))
    assert r.world.reshape(4, 4).T.tolist() == m.tolist()
    assert r.position[0, 0].tolist() == [3.0, 2.0, 3.0]    # M * (2, 0, 0, 1): x = 2 + 1, y = 2, z = 3


def test_matrix_node_static_and_animated(oracle32):
    # a Matrix node keeps its matrix unless animated; animated -> decomposed once, channel replaces T
    m = oracle32.trs_to_mat4([1, 2, 3], [0, 0, 1, 0], [2, 2, 2]).astype(np.float32)
    mats = np.zeros((2, 16), np.float32); mats[0] = m; mats[1] = m
    ch = Channel(1, abi.PATH_TRANSLATION, abi.INTERP_STEP, [0], [[7, 7, 7]])
    rig = rig_of(parent=[-1, -1], joint_node=(0, 1), matrix=mats, has_matrix=np.array([1, 1], np.uint8), clips=[[ch]])
    r = oracle32.evaluate(rig, tau=[0.0], clip_id=[0], want=("world",))
    w = r.world.reshape(2, 4, 4)
    assert w[0].reshape(16).tolist() == m.tolist()
    assert w[1, 3].tolist() == [7, 7, 7, 1] and w[1, 0].tolist() == [-2, 0, 0, 0]


def test_root_transform_premultiplies(oracle32):
    root = oracle32.trs_to_mat4([100, 0, 0], [0, 0, 0, 1], [1, 1, 1]).astype(np.float32)[None]
    rig = rig_of(parent=[-1], t=[[1, 0, 0]], joint_node=(0,), pos=[(0, 0, 0)])
    r = oracle32.evaluate(rig, tau=[0.0], root=root, want=("position",))
    assert r.position[0, 0].tolist() == [101, 0, 0]


# ---- the tolerance budget is meaningful: fp32 oracle vs fp64 oracle ----------------------------------------------

def test_fp32_oracle_distance_to_fp64(oracle32, oracle64, capsys):
    from renderer_amd import synth
    spec = synth.tiny_spec(seed=11, verts=64, joints=8, nodes=10, targets=3)
    rig = synth.make_rig(spec)
    crowd = synth.make_crowd(spec, rig, instances=8)
    a = oracle32.evaluate(rig, crowd.tau, crowd.clip_id, crowd.morph_weights)
    b = oracle64.evaluate(rig, crowd.tau, crowd.clip_id, crowd.morph_weights)
    assert np.array_equal(a.key_index, b.key_index)
    err = np.abs(a.position.astype(np.float64) - b.position)
    rel = err / (1e-6 / 1e-5 + np.abs(b.position))
    with capsys.disabled():
        print(f"\n[oracle] fp32 vs fp64 position: max abs {err.max():.3e}, budget use {rel.max() / 1e-5:.3f} of 1e-5 rel / 1e-6 abs")
    assert rel.max() < 1e-5


# ---- calculate_tangent (renderer-asset/src/tangent.rs:15-58) ---------------------------------------
def test_tangent_single_triangle_and_shared_vertex(oracle32):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    # triangle (p0,p1,p2): (p2-p1) x (p0-p1) = (-1,1,0) x (-1,0,0) = (0,0,1), already unit
    t = oracle32.calculate_tangent(4, pos[:3])
    assert t.tolist() == [[0, 0, 1]] * 3
    # second face (0,3,1): (p1-p3) x (p0-p3) = (1,0,-1) x (0,0,-1) = (0,1,0); vertices 0 and 1 get both normals
    t = oracle32.calculate_tangent(4, pos, np.array([0, 1, 2, 0, 3, 1], np.uint32))
    s = np.float32(1.0) * (np.float32(1.0) / np.sqrt(np.float32(2.0)))
    assert t[0].tolist() == [0.0, float(s), float(s)] and t[1].tolist() == [0.0, float(s), float(s)]
    assert t[2].tolist() == [0, 0, 1] and t[3].tolist() == [0, 1, 0]


def test_tangent_order_of_accumulation_and_repeated_corner(oracle32):
    # a vertex named twice in a triangle receives that normal twice (tangent.rs:38 iterates the three indices)
    pos = np.array([[0, 0, 0], [2, 0, 0], [0, 3, 0], [0, 0, 5]], np.float32)
    one = oracle32.calculate_tangent(4, pos, np.array([0, 1, 2], np.uint32))
    assert one[0].tolist() == [0, 0, 1]            # (p2-p1) x (p0-p1) = (-2,3,0) x (-2,0,0) = (0,0,6) -> (0,0,1)
    # faces (0,1,2) -> (0,0,1) and (0,3,1): (p1-p3) x (p0-p3) = (2,0,-5) x (0,0,-5) = (0,10,0) -> (0,1,0);
    # vertex 0 sums (0,0,1) + (0,1,0), then a degenerate face (0,0,0) adds NaN to vertex 0 only
    t = oracle32.calculate_tangent(4, pos, np.array([0, 1, 2, 0, 3, 1, 0, 0, 0], np.uint32))
    assert np.isnan(t[0]).all() and np.isfinite(t[1:]).all()


def test_tangent_modes(oracle32):
    pos = np.eye(3, dtype=np.float32)
    for mode in (0, 1, 2):
        assert np.isnan(oracle32.calculate_tangent(mode, pos)).all()    # zero sums normalise to NaN (tangent.rs:54-56)
    with pytest.raises(NotImplementedError):
        oracle32.calculate_tangent(3, pos)                              # todo!() (tangent.rs:26)
    with pytest.raises(IndexError):
        oracle32.calculate_tangent(4, pos, np.array([0, 1, 3], np.uint32))
    # trailing vertices that do not fill a triangle are ignored (positions.len() / 3, tangent.rs:41)
    t = oracle32.calculate_tangent(4, np.concatenate([pos, pos[:2]]))
    assert np.isfinite(t[:3]).all() and np.isnan(t[3:]).all()


# ---- un-skinned meshes and several skins (skinrt.h ABI 6) ------------------------------------------------------------

def test_instance_uniform_layout_and_normal_matrix(oracle32):
    """InstanceUniform{transform, normal, padding} — renderer/src/renderer/uniform/transform.rs:8-14, :27-42: the matrix
    column-major, then Mat3(inverse(m)^T) as three vec4 columns with w = 0, then 12 bytes of padding: 124 bytes."""
    m = mat([[2, 0, 0, 0], [0, 4, 0, 0], [0, 0, 8, 0], [5, 6, 7, 1]])
    iu = oracle32.instance_uniform(m)
    assert iu.nbytes == 124
    assert iu[:16].tolist() == m.tolist()
    assert iu[16:28].tolist() == [0.5, 0, 0, 0, 0, 0.25, 0, 0, 0, 0, 0.125, 0]
    assert iu[28:].tolist() == [0, 0, 0]


def test_unskinned_vertex_is_instance_transform_times_position(oracle32):
    """color_vs_main / texture_vs_main — shader.wgsl:143-152, :167-176: instance.transform * vec4(position, 1),
    instance.normal * normal, instance.normal * tangent (not normalised; the outline variants normalise, :198-199)."""
    m = mat([[0, 2, 0, 0], [-2, 0, 0, 0], [0, 0, 2, 0], [1, 2, 3, 1]])  # quarter turn about z, scale 2, translation
    iu = oracle32.instance_uniform(m)
    p, n, t = oracle32.rigid_vertex(iu, [1, 0, 0], [1, 0, 0], [0, 1, 0])
    assert p.tolist() == [1, 4, 3] and n.tolist() == [0, 0.5, 0] and t.tolist() == [-0.5, 0, 0]
    p, n, t = oracle32.rigid_vertex(iu, [1, 0, 0], [1, 0, 0], [0, 1, 0], normalize=1)
    assert n.tolist() == [0, 1, 0] and t.tolist() == [-1, 0, 0]


def test_sections_use_their_own_skin_and_props_their_node(oracle32):
    """Two skins over one tree naming the same nodes in different orders, and a prop on a node without a skin: a vertex's
    joint index counts inside ITS section's skin (node.rs:95, loader/gltf/mod.rs:1124-1166)."""
    from renderer_amd.abi import Section, Skin
    rig = rig_of(parent=[-1, 0, 0], t=[[0, 0, 0], [10, 0, 0], [0, 20, 0]],
                 pos=[(1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)],
                 joints=[np.array([[0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0], [9, 9, 9, 9]], np.uint16)],
                 weights=[np.array([[1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], [np.nan] * 4], np.float32)])
    rig.skins = [Skin(np.array([1, 2], np.uint32)), Skin(np.array([2, 1], np.uint32))]
    rig.sections = [Section(0, 0, 0, 2), Section(0, 1, 2, 1), Section(2, -1, 3, 1)]
    assert rig.palette_count == 5
    r = oracle32.evaluate(rig, tau=[0.0], want=("position", "palette"))
    # vertex 0: skin 0 joint 0 = node 1; vertex 1: skin 0 joint 1 = node 2; vertex 2: skin 1 joint 0 = node 2; vertex 3: node 2's world
    assert r.position[0].tolist() == [[11, 1, 1], [1, 21, 1], [1, 21, 1], [1, 21, 1]]
    assert [r.palette[0, k, 12:15].tolist() for k in range(5)] == [[10, 0, 0], [0, 20, 0], [0, 20, 0], [10, 0, 0], [0, 20, 0]]


def test_section_targets_count_rows_from_the_sections_first_vertex(oracle32):
    from renderer_amd.abi import Section
    rig = rig_of(parent=[-1, 0], pos=[(0, 0, 0)] * 5, joint_node=(0,),
                 targets=[MorphTarget(position=np.array([[1, 0, 0], [2, 0, 0]], np.float32)),                       # dense over section 0
                          MorphTarget(position=np.array([[0, 4, 0]], np.float32), indices=np.array([2], np.uint32))])  # row 2 of section 1
    rig.sections = [Section(0, 0, 0, 2, 0, 1), Section(1, 0, 2, 3, 1, 1)]
    r = oracle32.evaluate(rig, tau=[0.0], morph_weights=[[0.5, 0.25]], want=("position",))
    assert r.position[0].tolist() == [[0.5, 0, 0], [1, 0, 0], [0, 0, 0], [0, 0, 0], [0, 1, 0]]


def test_weights_channels_are_matched_to_sections_by_node(oracle32):
    from renderer_amd.abi import Section
    ch = Channel(1, abi.PATH_WEIGHTS, abi.INTERP_LINEAR, np.array([0, 2], np.float32), np.array([[0.0], [1.0]], np.float32))
    rig = rig_of(parent=[-1, 0], pos=[(0, 0, 0)] * 2, joint_node=(0,), clips=[[ch]],
                 targets=[MorphTarget(position=np.array([[8, 0, 0]], np.float32)), MorphTarget(position=np.array([[0, 8, 0]], np.float32))])
    rig.sections = [Section(0, 0, 0, 1, 0, 1), Section(1, 0, 1, 1, 1, 1)]
    r = oracle32.evaluate(rig, tau=[0.5], morph_weights=[[0.5, 0.5]], want=("position", "key_index"))
    # section 0 keeps the host's weight (0.5), section 1 (node 1) takes the channel's sample at tau = 0.5: 0.25
    assert r.position[0].tolist() == [[4, 0, 0], [0, 2, 0]] and r.key_index[0].tolist() == [0]


def test_sampled_evaluation_is_a_subset_of_the_full_one(oracle32):
    """orc_evaluate_sampled (the full-size parity tests' oracle leg): vertex v of instance i is computed when
    v % step == (phase + i) % step and lands in row v // step — same arithmetic, same bits."""
    from renderer_amd import synth
    spec = synth.tiny_spec(seed=3, verts=333, targets=5, target_mode="sparse", target_frac=0.2)
    rig = synth.make_rig(spec)
    cr = synth.make_crowd(spec, rig, instances=9)
    full = oracle32.evaluate(rig, cr.tau, cr.clip_id, cr.morph_weights, want=("position", "normal", "tangent"))
    step, phase = 7, 3
    part = oracle32.evaluate_mt(rig, cr.tau, cr.clip_id, cr.morph_weights, want=("position", "normal", "tangent"), vertex_step=step,
                                vertex_phase=phase, threads=3)
    for i in range(9):
        ph = (phase + i) % step
        n = len(range(ph, 333, step))
        for k in ("position", "normal", "tangent"):
            assert np.array_equal(getattr(part, k)[i, :n].view(np.uint32), getattr(full, k)[i, ph::step].view(np.uint32)), (i, k)
