"""The quad-per-env IK of the product (peg_in_hole_gym_amd/csrc/pih_ikq.h: one env per quad of lanes, DPP prefix product of rigid
transforms) checked on the CPU: tests/emul runs the SAME source with the four lanes of a quad as four host threads in lockstep (every
cross-lane primitive = publish, barrier, read), against the oracle's ik_solve (calculateInverseKinematics restated, envs/utils.py:67,79).
On the GPU the same comparison runs through pih_ik / pih_ik_ur5 (tests/test_gpu_parity.py, tests/test_ur5_chain.py)."""
import numpy as np
import pytest

from tests.emul import emul as E

REST = np.array([0, -0.215, -np.pi / 3, -2.57, 0, 2.356, 2.356, 0, 0])


@pytest.fixture(scope="module", autouse=True)
def _build():
    E.build()


def test_panda_quad_ik_matches_oracle(oracle_mod):
    rng = np.random.default_rng(0)
    tq = oracle_mod.quat_from_euler([0, -np.pi, 0])
    for k in range(25):
        q0 = REST + np.concatenate([rng.uniform(-0.6, 0.6, 7), [0, 0]])
        p, qq = oracle_mod.fk_arm(q0, 9)
        tgt = p + rng.uniform(-0.05, 0.05, 3)
        ref = oracle_mod.ik(q0, tgt, tq)
        out, ee = E.ikq(q0, tgt, tq, "f64")
        # forward kinematics of the start pose: the prefix product gives every lane of the quad the same end-effector frame as getLinkState
        R = np.array([[1 - 2 * (qq[1] ** 2 + qq[2] ** 2), 2 * (qq[0] * qq[1] - qq[2] * qq[3]), 2 * (qq[0] * qq[2] + qq[1] * qq[3])],
                      [2 * (qq[0] * qq[1] + qq[2] * qq[3]), 1 - 2 * (qq[0] ** 2 + qq[2] ** 2), 2 * (qq[1] * qq[2] - qq[0] * qq[3])],
                      [2 * (qq[0] * qq[2] - qq[1] * qq[3]), 2 * (qq[1] * qq[2] + qq[0] * qq[3]), 1 - 2 * (qq[0] ** 2 + qq[1] ** 2)]])
        for l in range(4):
            np.testing.assert_allclose(ee[l, :3], p, atol=1e-12)
            np.testing.assert_allclose(ee[l, 3:].reshape(3, 3), R, atol=1e-12)
        np.testing.assert_allclose(out[:7], ref[:7], atol=1e-10)
        np.testing.assert_array_equal(out[7:], q0[7:])
        out32, _ = E.ikq(q0, tgt, tq, "f32")
        np.testing.assert_allclose(out32[:7], ref[:7], atol=2e-5)
    # fixed point: target = FK(q) -> the exit test fires in iteration 0, q* = q exactly
    p, qq = oracle_mod.fk_arm(REST, 9)
    out, _ = E.ikq(REST, p, qq, "f64")
    np.testing.assert_array_equal(out, REST)


def test_ur5_quad_ik_matches_oracle(oracle_mod):
    """6 joints + the ee frame = 7 of the quad's 8 slots (the last one is the identity padding); axes y and z: exercises the change of
    frame that turns every joint into a rotation about local z"""
    rng = np.random.default_rng(1)
    for k in range(25):
        q0 = rng.uniform(-2, 2, 6)
        p, _ = oracle_mod.fk_ur5(q0, 6)
        tgt = p + rng.uniform(-0.1, 0.1, 3)
        tq = oracle_mod.quat_from_euler(rng.uniform(-1, 1, 3))
        ref = oracle_mod.ik_ur5(q0, tgt, tq)
        out, ee = E.ikq(q0, tgt, tq, "f64", ur5=True)
        np.testing.assert_allclose(ee[:, :3], np.tile(p, (4, 1)), atol=1e-12)
        np.testing.assert_allclose(out, ref, atol=1e-9)
        out32, _ = E.ikq(q0, tgt, tq, "f32", ur5=True)
        np.testing.assert_allclose(out32, ref, atol=2e-4)
