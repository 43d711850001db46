"""Analytic known-answer tests that pin the physics oracle: PyBullet itself is absent, so these
replace a golden trajectory.  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

REST = np.array([0, -0.215, -np.pi / 3, -2.57, 0, 2.356, 2.356, 0, 0])


def _T(R=np.eye(3), t=(0, 0, 0)):
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t; return T


def _Rx(a):
    c, s = np.cos(a), np.sin(a); return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _Rz(a):
    c, s = np.cos(a), np.sin(a); return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def franka_dh_ee(q):
    """Published Franka modified-DH chain (Craig): a, d, alpha ; flange +0.107, hand yaw -pi/4, grasptarget +0.105;
    robot base yawed -pi/2 (envs/utils.py:33)."""
    a = [0, 0, 0, 0.0825, -0.0825, 0, 0.088]
    d = [0.333, 0, 0.316, 0, 0.384, 0, 0]
    al = [0, -np.pi / 2, np.pi / 2, np.pi / 2, -np.pi / 2, np.pi / 2, np.pi / 2]
    T = _T(_Rz(-np.pi / 2))
    for i in range(7):
        T = T @ _T(_Rx(al[i])) @ _T(t=(a[i], 0, 0)) @ _T(_Rz(q[i])) @ _T(t=(0, 0, d[i]))
    return T @ _T(t=(0, 0, 0.107)) @ _T(_Rz(-np.pi / 4)) @ _T(t=(0, 0, 0.105))


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
                     [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
                     [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y]])


def test_fk_vs_franka_dh(oracle_mod):
    rng = np.random.default_rng(0)
    for q7 in [REST[:7]] + [rng.uniform(-2.5, 2.5, 7) for _ in range(20)]:
        q = np.concatenate([q7, [0.01, 0.03]])
        p, qt = oracle_mod.fk_arm(q, 9)
        T = franka_dh_ee(q7)
        np.testing.assert_allclose(p, T[:3, 3], atol=1e-12)
        np.testing.assert_allclose(quat_to_R(qt), T[:3, :3], atol=1e-12)


def test_jacobian_vs_finite_differences(oracle_mod):
    rng = np.random.default_rng(1)
    for _ in range(5):
        q = np.concatenate([rng.uniform(-2, 2, 7), [0.02, 0.01]])
        Jl, Ja = oracle_mod.jacobian_ee(q)
        h = 1e-6
        for j in range(9):
            qp, qm = q.copy(), q.copy(); qp[j] += h; qm[j] -= h
            pp, qqp = oracle_mod.fk_arm(qp, 9); pm, qqm = oracle_mod.fk_arm(qm, 9)
            np.testing.assert_allclose(Jl[:, j], (pp - pm) / (2 * h), atol=1e-6)
            dR = quat_to_R(qqp) @ quat_to_R(qqm).T
            w = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / (4 * h)
            np.testing.assert_allclose(Ja[:, j], w, atol=1e-6)
        assert np.all(Jl[:, 7:] == 0) and np.all(Ja[:, 7:] == 0)    # finger joints do not move the grasp target


def test_ik_fixed_point_and_residual(oracle_mod):
    p, qt = oracle_mod.fk_arm(REST, 9)
    np.testing.assert_allclose(oracle_mod.ik(REST, p, qt), REST, atol=1e-9)       # target = FK(q)  =>  q* = q
    # one controller step away (dv = 2/240 per axis).  The restated BussIK DLS (J^T J + 0.5 I, 20 iterations) is a
    # contraction, not an exact solver (PyBullet's single-call IK is known to need repeated calls): check that one
    # call removes most of the error and that repeated calls converge below the 1e-4 residual threshold.
    tq = oracle_mod.quat_from_euler([0, -np.pi, 0])
    q = REST.copy()
    for _ in range(200):       # settle the orientation first, as the env does over the first steps
        p, _ = oracle_mod.fk_arm(q, 9); q = oracle_mod.ik(q, p, tq)
    p, _ = oracle_mod.fk_arm(q, 9)
    tgt = p + np.array([1, -1, 1]) * 2 / 240
    qs = oracle_mod.ik(q, tgt, tq)
    p2, q2 = oracle_mod.fk_arm(qs, 9)
    assert np.linalg.norm(p2 - tgt) < 0.25 * np.linalg.norm(p - tgt)
    for _ in range(10):
        qs = oracle_mod.ik(qs, tgt, tq)
    p2, q2 = oracle_mod.fk_arm(qs, 9)
    assert np.linalg.norm(p2 - tgt) < 1e-4
    assert abs(abs(np.dot(q2, tq)) - 1) < 1e-5


def test_mass_matrix_properties(oracle_mod):
    o = oracle_mod.Oracle(4)
    for s in o.get_state():
        M = oracle_mod.mass_matrix(s)
        np.testing.assert_allclose(M, M.T, atol=1e-12)
        assert np.all(np.linalg.eigvalsh(M) > 0)
        assert np.all(M[:9, 9:] == 0)
        np.testing.assert_allclose(M[9:12, 9:12], np.eye(3) * (0.00111 + 24 * 0.0111), atol=1e-12)   # total pipe mass


def test_kinetic_energy_matches_link_sum(oracle_mod):
    """u^T M u / 2 == sum over links of (m |v_c|^2 + w^T I w)/2 with link twists from finite-differenced FK of the
    arm (independent of the RNEA used to build M)."""
    rng = np.random.default_rng(3)
    q = np.concatenate([rng.uniform(-1, 1, 7), [0.01, 0.02]]); qd = rng.uniform(-1, 1, 9)
    s = np.zeros(128); s[0:9] = q; s[24] = 1.0
    M = oracle_mod.mass_matrix(s)[:9, :9]
    import re
    hdr = open(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "include", "pih_model.h")).read()
    def arr(name):
        body = re.search(r"#define %s (.*)" % name, hdr).group(1).split("/*")[0]
        return np.array(eval(body.replace("{", "[").replace("}", "]")), dtype=float)
    mass, com, iner = arr("PIH_LINK_MASS")[:9], arr("PIH_LINK_COM")[:9], arr("PIH_LINK_INERTIA")[:9]
    h, ke = 1e-6, 0.0
    for L in range(9):
        pp, qp = oracle_mod.fk_arm(q + h * qd, L); pm, qm = oracle_mod.fk_arm(q - h * qd, L)
        p0, q0 = oracle_mod.fk_arm(q, L)
        Rp, Rm, R0 = quat_to_R(qp), quat_to_R(qm), quat_to_R(q0)
        vc = ((pp + Rp @ com[L]) - (pm + Rm @ com[L])) / (2 * h)
        dR = Rp @ Rm.T
        w = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / (4 * h)
        I = iner[L]; Il = np.array([[I[0], I[3], I[4]], [I[3], I[1], I[5]], [I[4], I[5], I[2]]])
        ke += 0.5 * mass[L] * vc @ vc + 0.5 * w @ (R0 @ Il @ R0.T) @ w
    np.testing.assert_allclose(0.5 * qd @ M @ qd, ke, rtol=1e-6)


def test_free_fall_closed_form(oracle_mod):
    """Semi-implicit Euler with Bullet's link damping (k = 0.04): v += dt(-g - k v (1+|v|)), z += dt v."""
    o = oracle_mod.Oracle(1, enable_self_collision=0)
    s = o.get_state(); s[0, 20] = 1.0; o.set_state(s)          # lift the pipe clear of the table
    z, v, dt = 1.0, 0.0, 1 / 240
    p, _ = oracle_mod.fk_arm(REST, 9)
    a = np.array([[p[0], p[1], p[2], 0.0]])
    for n in range(40):
        o.step(a)
        v += dt * (-9.8 - 0.04 * v * (1 + abs(v))); z += dt * v
        st = o.get_state()[0]
        assert abs(st[20] - z) < 1e-12 and abs(st[27] - v) < 1e-12
        assert np.allclose(st[28:31], 0, atol=1e-12) and o.ncontacts()[0] == 0
        np.testing.assert_allclose(st[54:77], 0, atol=1e-9)     # joints stay put in uniform gravity


def test_resting_normal_force_is_weight(oracle_mod):
    """Pipe at rest on the table: sum of normal forces = m g = (0.00111 + 24*0.0111)*9.8 = 2.6215 N within 1e-2 N."""
    o = oracle_mod.Oracle(2)
    p, _ = oracle_mod.fk_arm(REST, 9)
    a = np.tile([p[0], p[1], p[2], 0.0], (2, 1))
    for _ in range(1500):
        o.step(a)
    f = np.mean([(o.step(a), o.contact_force())[1] for _ in range(60)], axis=0)
    np.testing.assert_allclose(f, 2.6215, atol=1e-2)
    st = o.get_state()
    # quasi-static: with 50 PGS iterations the 23 velocity-motor rows do not fully converge, so the bent pipe keeps
    # sagging very slowly ("flexible tube"); linear speed of the base is already below 2 cm/s
    assert np.all(np.abs(st[:, 25:28]) < 2e-2) and np.all(np.abs(st[:, 54:77]) < 5e-2)
    assert np.all(st[:, 20] > -0.05) and np.all(st[:, 20] < 0.2)


def _straight_pipe_on_table(oracle_mod):
    o = oracle_mod.Oracle(1)
    s = o.get_state(); s[0, 31:54] = 0; s[0, 20] = -0.04 + 1e-4; s[0, 18:20] = [0.3, -0.6]; o.set_state(s)
    p, _ = oracle_mod.fk_arm(REST, 9); a = np.array([[p[0], p[1], p[2], 0.0]])
    for _ in range(120):
        o.step(a)
    return o, a


def test_friction_rolling_without_slipping(oracle_mod):
    """A straight pipe shoved sideways (perpendicular to its axis) must end up rolling: contact-point velocity
    v_x - w_y r -> 0 (static friction holds, no rolling resistance in the model)."""
    o, a = _straight_pipe_on_table(oracle_mod)
    s = o.get_state(); s[0, 25] = 0.5; o.set_state(s)
    for _ in range(30):
        o.step(a)
    st = o.get_state()[0]
    assert st[25] > 0.15 and st[29] > 10.0
    assert abs(st[25] - st[29] * 0.01) < 0.02 * st[25] + 2e-3
    np.testing.assert_allclose(o.contact_force(), 2.6215, atol=0.1)


def test_coulomb_sliding_decelerates_within_cone(oracle_mod):
    """Shoved ALONG its axis the pipe cannot roll: it slides, never speeds up, decelerates by at most
    mu_max g (mu clamped to 10, Bullet MAX_FRICTION) and by at least mu_min g = 0.5 g, and stops."""
    o, a = _straight_pipe_on_table(oracle_mod)
    s = o.get_state(); s[0, 26] = 0.5; o.set_state(s)
    vprev, t_stop = 0.5, None
    for n in range(240):
        o.step(a)
        vy = o.get_state()[0, 26]
        assert vy <= vprev + 1e-6
        if vprev > 0.05:
            dec = (vprev - vy) * 240
            assert 0.5 * 9.8 * 0.9 <= dec <= 10.0 * 9.8 * 1.1
        vprev = vy
        if t_stop is None and abs(vy) < 1e-3:
            t_stop = n
    assert t_stop is not None and t_stop >= 2


def test_joint_limit_stops_the_joint(oracle_mod):
    """Bullet's velocity-level joint limit: a joint 0.02 rad below its upper limit moving at 50 rad/s may keep exactly
    pen / dt = 0.02 * 240 = 4.8 rad/s in the first step (it arrives AT the limit) and is stopped there afterwards.  Scripted
    mode, state 0: the arm is held by the weak load-time velocity motors only, so the limit row is what acts."""
    O = oracle_mod
    hi3 = 2.9671                                   # Panda joint 3 (index 2) upper limit, include/pih_model.h
    o = O.Oracle(1, enable_self_collision=0, mode=1, dv=0.05)
    s = o.get_state()
    s[0, 2] = hi3 - 0.02; s[0, 9 + 2] = 50.0; s[0, 18] = 5.0     # pipe out of the way
    o.set_state(s)
    o.step(np.zeros((1, 4)))
    st = o.get_state()
    assert abs(st[0, 11] - 0.02 * 240) < 1e-4 and abs(st[0, 2] - hi3) < 1e-6
    for t in range(20):
        o.step(np.zeros((1, 4)))
        st = o.get_state()
        assert abs(st[0, 2] - hi3) < 1e-5 and abs(st[0, 11]) < 1e-3


def test_hole_tube_contact_geometry(oracle_mod):
    """Signed distance to the annular tube: a straight pipe threaded through the bore, its axis 2 mm short of touching the
    inner wall from above... i.e. the sphere surfaces 2 mm from the bore's bottom: every sample inside the tube's length gets
    a contact with the exact radial normal (+z), depth 0.002 and the contact point on the common normal."""
    O = oracle_mod
    hole = np.array([0.5, -0.2, 0.2]); rin, r, hl = 0.01536, 0.01, 0.016
    o = O.Oracle(1, enable_self_collision=0)
    s = o.get_state()
    s[0, 31:54] = 0                                              # straight pipe
    q = np.array([0, 0, np.sin(-np.pi / 4), np.cos(-np.pi / 4)])  # local y -> world x
    rho = rin - r - 0.002                                         # distance of the pipe axis below the bore axis
    s[0, 18:21] = [hole[0] - 0.30, hole[1], hole[2] - rho]; s[0, 21:25] = q
    s[0, 25:31] = 0
    o.set_state(s)
    o.step(np.array([[0.3, 0.0, 0.5, 0.0]]))
    c = o.debug_contacts(0)
    tube = c[(c[:, 10] >= 100) & (c[:, 10] < 300)]
    assert len(tube) >= 3
    inside = tube[np.abs(tube[:, 2] - hole[0]) <= hl]
    assert len(inside) >= 2
    np.testing.assert_allclose(inside[:, 5:8], np.tile([0, 0, 1.0], (len(inside), 1)), atol=1e-9)
    np.testing.assert_allclose(inside[:, 8], 0.002, atol=1e-9)
    # contact point: on the sphere-centre -> wall normal, half the gap beyond the sphere surface (z = centre - r - depth/2)
    np.testing.assert_allclose(inside[:, 4], hole[2] - rho - r - 0.001, atol=1e-9)
    np.testing.assert_allclose(inside[:, 3], hole[1], atol=1e-12)
    # samples beyond the tube's ends see its rim: their normals tilt outwards along the axis and the gap grows
    outside = tube[np.abs(tube[:, 2] - hole[0]) > hl + 1e-6]
    assert (outside[:, 8] > 0.002).all() and (np.abs(outside[:, 5]) > 0).all()


def test_new_seed_needs_a_full_reset_oracle(oracle_mod):
    """the oracle mirrors pih_reset's rule: a new seed (seed != 0 or a pending reseed) together with a mask is refused"""
    o = oracle_mod.Oracle(6, seed=4)
    s0 = o.get_state().copy()
    m = np.array([1, 0, 1, 0, 1, 0], dtype=np.uint8)
    with pytest.raises(ValueError):
        o.reset(m, seed=5)
    o.reseed(5)
    with pytest.raises(ValueError):
        o.reset(m)
    np.testing.assert_array_equal(o.get_state(), s0)
    o.reset()
    np.testing.assert_array_equal(o.get_state()[:, :91], oracle_mod.Oracle(6, seed=5).get_state()[:, :91])
    before = o.get_state()[:, 92].copy()
    o.reset(m)
    after = o.get_state()[:, 92]
    assert (after[m == 1] > before[m == 1]).all() and np.array_equal(after[m == 0], before[m == 0])


def test_structural_variants_default_is_the_product_algorithm(oracle_mod):
    """oracle/pih_oracle.h piho_variant (round 4: the structure sweep of tools/ill_conditioned_causes.py): the DEFAULT variant is bit for bit
    the oracle every parity test uses; each switch changes the rollout; one friction direction leaves the dir-2 multipliers at zero"""
    N = 16
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (120, N, 4))

    def roll(**var):
        o = oracle_mod.Oracle(N, seed=5)
        if var is not None:
            o.set_variant(**var)
        for a in acts:
            o.step(a)
        return o
    base = oracle_mod.Oracle(N, seed=5)
    for a in acts:
        base.step(a)
    sb = base.get_state()
    np.testing.assert_array_equal(roll().get_state(), sb)                      # explicit defaults == never touched
    assert base.ncontacts().max() >= 3
    for var in (dict(row_order=1), dict(friction_dirs=1), dict(mu_clamp=1.0), dict(pipe_motor_impulse=0.0), dict(row_impulse_cap=1e-3), dict(max_coord_vel=5.0)):
        o = roll(**var)
        assert not np.array_equal(o.get_state()[:, :77], sb[:, :77]), var
        assert np.isfinite(o.get_state()).all()
        if var == dict(friction_dirs=1):
            lt, _ = o.debug_friction()
            assert np.abs(lt[:, :, 1]).max() == 0 and np.abs(lt[:, :, 0]).max() > 0
        if var == dict(max_coord_vel=5.0):
            assert np.abs(o.get_state()[:, 9:18]).max() <= 5.0 + 1e-12
