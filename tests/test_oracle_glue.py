"""Oracle vs the golden vectors produced by the reference's own pure-Python glue (tests/golden/make_glue_golden.py)."""
import numpy as np


def test_vel_constraint(golden, oracle_mod):
    for c in golden["vel_constraint"]:
        np.testing.assert_allclose(oracle_mod.vel_constraint(c["cur"], c["tar"], c["dv"]), c["out"], rtol=0, atol=1e-15)


def test_rotate_vector(golden, oracle_mod):
    for c in golden["rotate_vector"]:
        np.testing.assert_allclose(oracle_mod.rotate_vector(c["vec"], c["quat"]), c["out"], rtol=0, atol=1e-14)


def test_fsm_clock(golden, oracle_mod):
    g = golden["fsm"]
    tr = oracle_mod.fsm_trace(g["trace_len"], g["timestep"])
    assert [tr.count(s) for s in range(10)] == g["iters_per_state"] == [59, 481, 481, 241, 361, 361, 121, 60, 60, 1]
    assert [tr.index(s) for s in range(10)] == g["first_index_of_state"]
    assert tr[-1] == 9 and len(tr) == 2226


def test_create_env_offsets(golden, oracle_mod):
    for c in golden["create_env_offsets"]:
        np.testing.assert_allclose(oracle_mod.env_offsets(c["offset"], c["n"]), np.array(c["out"]).reshape(c["n"], 3), atol=0)


def test_quat_from_euler_matches_reset_calls(golden, oracle_mod):
    # the recording client used pybullet's published formula; the oracle must agree with the quaternions the
    # reference passed to loadURDF (panda yaw -pi/2, table/hole yaw +pi/2)
    calls = golden["reset"][0]["calls"]
    panda = [c for c in calls if c[0] == "loadURDF" and c[1] == "panda.urdf"][0]
    hole = [c for c in calls if c[0] == "loadURDF" and c[1] == "hole.urdf"][0]
    np.testing.assert_allclose(oracle_mod.quat_from_euler([0, 0, -np.pi / 2]), panda[3], atol=1e-15)
    np.testing.assert_allclose(oracle_mod.quat_from_euler([0, 0, np.pi / 2]), hole[3], atol=1e-15)
    assert hole[2] == [0.5, -0.2, 0.2] and hole[6] == 0.016 and hole[4] == 1
    pipe = [c for c in calls if c[0] == "loadURDF" and c[1] == "pipe.urdf"][0]
    assert pipe[6] == 0.01 and pipe[4] == 0 and pipe[2][2] == 0.11


def test_reset_matches_reference_structure(golden, oracle_mod):
    """Same scene structure / distributions as PegInHole.reset (draw order App. E); the reference never seeds, so only
    ranges and the call structure are comparable, not the values."""
    for r in golden["reset"]:
        rj = [c for c in r["calls"] if c[0] == "resetJointState" and c[1] == 3]
        assert 5 <= len(rj) <= 24 and len(set(c[2] for c in rj)) == len(rj)
        assert all(0 <= c[3] <= np.pi / 3 for c in rj)
        assert r["grasp_joint_idx"] in (0, 23) and abs(r["random_vector"][1]) <= 0.03
    o = oracle_mod.Oracle(256)
    s = o.get_state()
    from oracle.oracle import Config  # noqa: F401
    assert np.all((s[:, 18] >= -0.2) & (s[:, 18] <= 0.2)) and np.all((s[:, 19] >= -0.6) & (s[:, 19] <= -0.4)) and np.all(s[:, 20] == 0.11)
    qj = s[:, 31:54]
    assert np.all((qj >= 0) & (qj <= np.pi / 3))
    nz = (qj != 0).sum(1)
    assert nz.min() >= 4 and nz.max() <= 23          # k in [5,24] draws over 24 indices, index 0 is the fixed joint
    assert set(np.unique(s[:, 89])) <= {0.0, 23.0} and len(np.unique(s[:, 89])) == 2
    assert np.all(np.abs(s[:, 90]) <= 0.03)
    np.testing.assert_allclose(s[:, 0:9], np.tile([0, -0.215, -np.pi / 3, -2.57, 0, 2.356, 2.356, 0, 0], (256, 1)))
    # per-env streams differ, and the same seed reproduces
    assert len(np.unique(s[:, 18])) == 256
    np.testing.assert_array_equal(oracle_mod.Oracle(256).get_state(), s)


def test_registry_and_spaces(golden):
    assert [r["id"] for r in golden["registry"]] == ["peg-in-hole-v0", "peg-in-hole-mp-v0"]
    assert golden["spaces"]["action_shape"] == [4] and golden["spaces"]["observation_shape"] == [5]
    assert golden["pegin_attrs"] == {"pandaEndEffectorIndex": 11, "pandaNumDofs": 7}
