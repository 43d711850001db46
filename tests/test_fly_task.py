"""'random-fly' task (BASELINE configs[4]: UR5 + free-flying object): known-answer tests that pin the CPU oracle, the product's
per-lane algorithm (pih_fly.h, host build in tests/emul) against that oracle, and the task plugin path of the facade.  The GPU
parity tests proper are in test_gpu_fly.py.  PARITY UNPINNED vs PyBullet; the task class is not in the reference snapshot."""
import os

import numpy as np
import pytest

from tests.emul import emul as E

DT = 1.0 / 120.0      # README.md:38 args=['Banana', 1/120.]
REST = np.array([0, -np.pi / 2, np.pi / 2, -np.pi / 2, -np.pi / 2, 0])


@pytest.fixture(scope="module", autouse=True)
def _build():
    E.build()


# ------------------------------------------------------------------------------------------------ oracle known answers
def test_free_flight_closed_form(oracle_mod):
    """Object far from arm and table: semi-implicit Euler with Bullet's damping v' = v + dt (g - (k + k|v|) v), k = 0.04."""
    O = oracle_mod
    o = O.FlyOracle(1, dt=DT, residual_threshold=0.0)
    s = o.get_state()
    s[0, O.F_OPOS:O.F_OPOS + 3] = [3.0, 3.0, 5.0]; s[0, O.F_OVLIN:O.F_OVLIN + 3] = [1.0, -2.0, 3.0]; s[0, O.F_OVANG:O.F_OVANG + 3] = 0
    o.set_state(s)
    p = np.array([3.0, 3.0, 5.0]); v = np.array([1.0, -2.0, 3.0])
    for _ in range(60):
        o.step(np.tile(np.r_[0.3, 0.0, 0.5, 0, 0, 0], (1, 1)))
        v = v + DT * (np.array([0, 0, -9.8]) - 0.04 * (1 + np.linalg.norm(v)) * v); p = p + DT * v
    st = o.get_state()[0]
    np.testing.assert_allclose(st[O.F_OPOS:O.F_OPOS + 3], p, atol=1e-12); np.testing.assert_allclose(st[O.F_OVLIN:O.F_OVLIN + 3], v, atol=1e-12)


def test_torque_free_spin_conserves_angular_momentum_direction(oracle_mod):
    """Spin about a principal axis with damping only: the axis stays fixed and |w| decays as w' = w (1 - dt (k + k|w|))."""
    O = oracle_mod
    o = O.FlyOracle(1, dt=DT)
    s = o.get_state(); s[0, O.F_OPOS:O.F_OPOS + 3] = [3, 3, 5]; s[0, O.F_OVLIN:O.F_OVLIN + 3] = 0; s[0, O.F_OVANG:O.F_OVANG + 3] = [0, 0, 4.0]
    o.set_state(s)
    w = 4.0
    for _ in range(30):
        o.step(np.zeros((1, 6))); w = w * (1 - DT * 0.04 * (1 + w))
    st = o.get_state()[0]
    np.testing.assert_allclose(st[O.F_OVANG:O.F_OVANG + 3], [0, 0, w], atol=1e-12)
    assert abs(st[O.F_OQUAT] ) < 1e-12 and abs(st[O.F_OQUAT + 1]) < 1e-12       # rotation stays about z


def test_arm_mass_matrix_is_spd_and_matches_kinetic_energy(oracle_mod):
    """Two independent routes to the arm's inertia: M from unit accelerations through RNEA vs the sum of link kinetic energies."""
    O = oracle_mod
    rng = np.random.default_rng(0)
    for _ in range(5):
        q = rng.uniform(-2, 2, 6); qd = rng.uniform(-2, 2, 6)
        M = O.fly_mass_matrix(q)
        np.testing.assert_allclose(M, M.T, atol=1e-12)
        assert np.linalg.eigvalsh(M).min() > 0
        assert abs(0.5 * qd @ M @ qd - O.fly_arm_kinetic_energy(q, qd)) < 1e-12 * (1 + abs(qd @ M @ qd))
    # the merged wrist_3 + ee_link (1.1879 kg) swung by the last joint: M[5][5] = izz-like term about the joint axis (y)
    M = O.fly_mass_matrix(REST)
    assert 1e-4 < M[5, 5] < 1e-1 and M[0, 0] > M[5, 5]


def test_arm_holds_and_tracks_with_ur_execute(oracle_mod):
    """ur_execute (envs/utils.py:70-82): commanding the current end-effector pose keeps the arm at rest against gravity to within
    the POSITION_CONTROL droop; commanding a pose 10 cm away moves the end effector towards it."""
    O = oracle_mod
    o = O.FlyOracle(1, dt=DT, max_episode_steps=100000)
    s = o.get_state(); s[0, O.F_OPOS:O.F_OPOS + 3] = [5, 5, 5]; s[0, O.F_OVLIN:O.F_OVLIN + 3] = 0; o.set_state(s)     # object out of the way
    ee0, qe = O.fk_ur5(REST, 6)
    eul = O.euler_from_quat(qe)
    a = np.r_[ee0, eul][None]
    for _ in range(240):
        s = o.get_state(); s[0, O.F_OPOS:O.F_OPOS + 3] = [5, 5, 5]; s[0, O.F_OVLIN:O.F_OVLIN + 3] = 0; o.set_state(s)
        obs, _, _ = o.step(a)
    assert np.abs(obs[0, :3] - ee0).max() < 0.02
    tgt = ee0 + np.array([0.0, 0.1, 0.05]); a2 = np.r_[tgt, eul][None]
    d0 = np.linalg.norm(obs[0, :3] - tgt)
    for _ in range(480):
        s = o.get_state(); s[0, O.F_OPOS:O.F_OPOS + 3] = [5, 5, 5]; s[0, O.F_OVLIN:O.F_OVLIN + 3] = 0; o.set_state(s)
        obs, _, _ = o.step(a2)
    assert np.linalg.norm(obs[0, :3] - tgt) < 0.35 * d0


def test_object_lands_on_table_and_is_stopped(oracle_mod):
    """Frictionless sphere contacts against the table plane: dropped flat, the object stops at z = table + sphere radius (no
    penetration beyond the slop, no bounce: restitution 0), and the episode ends (landed)."""
    O = oracle_mod
    o = O.FlyOracle(1, dt=DT, auto_reset=0)
    s = o.get_state(); s[0, O.F_OPOS:O.F_OPOS + 3] = [0.9, 0.9, 0.1]; s[0, O.F_OVLIN:O.F_OVLIN + 3] = 0; s[0, O.F_OVANG:O.F_OVANG + 3] = 0; o.set_state(s)
    dmin = 1.0; force = 0; vz = []
    for t in range(60):
        _, _, done = o.step(np.zeros((1, 6)))
        st = o.get_state()[0]
        c = o.debug_contacts(0)
        if c[5:, 0].any():
            dmin = min(dmin, c[5:][c[5:, 0] > 0][:, 8].min())
        force = max(force, st[O.F_CFORCE]); vz.append(st[O.F_OVLIN + 2])
        if done[0]:
            break
    assert done[0] and t < 40
    assert dmin > -1e-3                              # no sphere sinks into the table (speculative contact rows stop it at the surface)
    assert force > 9.8                               # the landing impulse exceeds the static weight m g
    assert vz[-1] > -0.3 and min(vz) < -1.0          # it was falling at > 1 m/s and has been stopped (restitution 0: no bounce)


def test_random_pos_on_the_panda_shell_and_launch_reaches_the_aim_point(oracle_mod):
    """random_pos_in_panda_space (envs/utils.py:97-107) with the counter RNG: every spawn lies on the 0.7 m sphere centred
    (0, 0, 0.2), upper half; the build-defined launch law sends the object through its aim point after the drawn flight time
    (checked without damping: closed-form ballistic arc)."""
    O = oracle_mod
    for seed in range(50):
        p = O.fly_random_pos(seed, 0)
        assert abs(np.linalg.norm(p - [0, 0, 0.2]) - 0.7) < 1e-12 and p[2] >= 0.2
    o = O.FlyOracle(64, seed=3, dt=DT)
    s = o.get_state()
    p0, v0 = s[:, O.F_OPOS:O.F_OPOS + 3], s[:, O.F_OVLIN:O.F_OVLIN + 3]
    assert (np.abs(np.linalg.norm(p0 - [0, 0, 0.2], axis=1) - 0.7) < 1e-9).all()
    # aim point c = p0 + v0 T + g T^2 / 2 must fall in the box the launch law draws it from, for SOME T in [0.6, 1.0]
    ok = 0
    for i in range(64):
        for T in np.linspace(0.6, 1.0, 401):
            c = p0[i] + v0[i] * T + 0.5 * np.array([0, 0, -9.8]) * T * T
            if abs(c[0]) <= 0.151 and abs(c[1]) <= 0.151 and 0.349 <= c[2] <= 0.651:
                ok += 1; break
    assert ok == 64


# ------------------------------------------------------------------------------------------------ product algorithm vs oracle (CPU)
@pytest.mark.parametrize("prec,obj", [("f64", 0), ("f32", 0), ("f64", 1), ("f32", 1)])
def test_per_lane_algorithm_matches_oracle(oracle_mod, prec, obj):
    """pih_fly.h (articulated-body algorithm + impulse responses, what every GPU lane runs) against the oracle (RNEA + Cholesky),
    resynchronised every step through episodes with auto-reset: state, observation, reward, done, contact count -- for both objects of
    the generated table (object_id 0 'Banana': 5 spheres, 1 'Amicelli': 2 spheres) and with the UR5-vs-table contacts live (random
    actions drive the arm into the table in about a third of the env-steps)."""
    O = oracle_mod
    N = 24
    kw = dict(seed=1, dt=DT, residual_threshold=0.0, auto_reset=1, max_episode_steps=150, object_id=obj)
    o = O.FlyOracle(N, **kw); e = E.EmulFly(N, prec, debug=1, **kw)
    tol = 0.0 if prec == "f64" else 1e-6
    np.testing.assert_allclose(e.get_state(), o.get_state(), atol=max(tol, 1e-15))
    rng = np.random.default_rng(0)
    perr = []; ncs = 0
    for t in range(400):
        a = rng.uniform(-1, 1, (N, 6))
        e.set_state(o.get_state())
        oo, ro, do = o.step(a); oe, re, de = e.step(a)
        so, se = o.get_state(), e.get_state()
        np.testing.assert_array_equal(do, de); np.testing.assert_array_equal(ro, re)
        np.testing.assert_array_equal(so[:, 44], se[:, 44])
        ncs += int(so[:, 44].sum())
        perr.append(np.abs(so[:, :31] - se[:, :31]).max())
        np.testing.assert_allclose(oo, oe, atol=1e-9 if prec == "f64" else 2e-4)
    print(prec, "max one-step state error %.2e, contact env-steps %d" % (max(perr), ncs))
    assert ncs > 300
    # (fp64: 1e-13 median; up to 1e-8 where an arm link is pressed into the table by its position motor -- motor and contact rows
    #  disagree and 50 sweeps do not converge, so the two derivations' rounding differences are amplified)
    assert max(perr) < (5e-8 if prec == "f64" else 2e-3)
    assert np.median(perr) < (2e-9 if prec == "f64" else 5e-5)      # (per step: the max over the 24 envs)


def test_arm_stops_at_the_table(oracle_mod):
    """UR5 capsules vs the table top (envs/assets/meshes/ur5/collision/<link>.stl -> capsules, slots 2 NS .. 2 NS + 4): an end-effector
    target far below the table drives the arm down; the wrist capsules come to rest ON the table (no capsule end more than the contact
    slop below z = -0.05 + r), the contact carries a positive normal force, and the host build of pih_fly.h sees the same contacts."""
    O = oracle_mod
    o = O.FlyOracle(1, seed=3, dt=DT, auto_reset=0, max_episode_steps=100000); e = E.EmulFly(1, "f64", debug=1, seed=3, dt=DT, auto_reset=0, max_episode_steps=100000)
    s = o.get_state(); s[0, O.F_OPOS:O.F_OPOS + 3] = [5.0, 5.0, 50.0]; s[0, O.F_OVLIN:O.F_OVLIN + 3] = 0; o.set_state(s)      # object out of the way
    ee0, qe = O.fk_ur5(REST, 6)
    a = np.r_[ee0[0], ee0[1], -0.6, O.euler_from_quat(qe)][None]
    seen = 0
    for t in range(400):
        e.set_state(o.get_state())
        o.step(a); e.step(a)
        c = o.debug_contacts(0)
        nslots = c.shape[0]
        assert nslots == 15
        if c[10:, 0].any():
            seen += 1
            d = e.get_debug()[0]
            for k in range(10, 15):
                g = d[16 + 10 * k:16 + 10 * k + 10]
                assert g[0] == c[k, 0]
                if c[k, 0]:
                    assert g[1] == c[k, 1] == k - 9                                  # slot 2 NS + a holds arm link 1 + a
                    np.testing.assert_allclose(g[2:9], c[k, 2:9], atol=1e-9)
        np.testing.assert_allclose(e.get_state()[0, :31], o.get_state()[0, :31], atol=5e-8)
    st = o.get_state()[0]
    assert seen > 100 and st[O.F_CFORCE] > 1.0
    # lowest capsule end of the distal links: resting on the table top within the slop / one step of ERP
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pih_model.h")).read()
    def mac(n):
        return np.array(eval(re.search(r"#define %s (.*)" % n, hdr).group(1).split("/*")[0].replace("{", "[").replace("}", "]")))
    A, B, R = mac("PIH_UR5_CAP_A"), mac("PIH_UR5_CAP_B"), mac("PIH_UR5_CAP_R")
    low = 1e9
    for L in range(1, 6):
        p, q = O.fk_ur5(st[0:6], L)
        qx, qy, qz, qw = q
        Rm = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                       [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                       [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
        for end in (A[L], B[L]):
            low = min(low, (p + Rm @ end)[2] - R[L])
    assert -0.05 - 0.003 < low < -0.05 + 0.01, low


def test_arm_object_contact_transfers_momentum(oracle_mod):
    """Object thrown straight at the forearm capsule: the contact slot of the hit sphere becomes valid with that link, the
    normal impulse is positive and the object's approach velocity along the normal is removed (restitution 0); oracle and product
    algorithm agree on the contact record."""
    O = oracle_mod
    kw = dict(dt=DT, residual_threshold=0.0)
    o = O.FlyOracle(1, **kw); e = E.EmulFly(1, "f64", debug=1, **kw)
    # forearm (link 2) capsule midpoint in the rest pose
    import ctypes  # noqa: F401
    q = REST
    p2, _ = O.fk_ur5(q, 2); p3, _ = O.fk_ur5(q, 3)
    mid = 0.5 * (p2 + p3)
    s = o.get_state()
    s[0, O.F_OPOS:O.F_OPOS + 3] = mid + np.array([0.0, 0.25, 0.0]); s[0, O.F_OVLIN:O.F_OVLIN + 3] = [0, -4.0, 0]; s[0, O.F_OVANG:O.F_OVANG + 3] = 0
    o.set_state(s)
    hit = False
    a = np.r_[O.fk_ur5(q, 6)[0], O.euler_from_quat(O.fk_ur5(q, 6)[1])][None]
    for t in range(20):
        e.set_state(o.get_state())
        o.step(a); e.step(a)
        c = o.debug_contacts(0)
        d = e.get_debug()[0]
        if c[:5, 0].any():
            k = int(np.argmax(c[:5, 0]))
            assert c[k, 1] in (1, 2, 3) and c[k, 9] > 0
            g = d[16 + 10 * k:16 + 10 * k + 10]
            assert g[0] == 1 and g[1] == c[k, 1]
            np.testing.assert_allclose(g[2:9], c[k, 2:9], atol=1e-9)
            np.testing.assert_allclose(d[200 + int(g[9])], c[k, 9], rtol=1e-6)
            st = o.get_state()[0]
            assert st[O.F_OVLIN + 1] > -4.0 + 0.5       # the approach velocity was (partly) removed by the arm
            hit = True
            break
    assert hit


# ------------------------------------------------------------------------------------------------ task plugin path (host logic)
def test_task_registry_and_facade_for_random_fly():
    import peg_in_hole_gym_amd as pih
    from peg_in_hole_gym_amd.envs import TASK_LIST
    from peg_in_hole_gym_amd.envs.peg_in_hole import MetaEnv, PegInHole, RandomFly
    from tests.oracle_backend import factory
    assert TASK_LIST["random-fly"] is RandomFly and TASK_LIST["peg-in-hole"] is PegInHole
    assert (PegInHole.task_id, RandomFly.task_id) == (0, 1) and issubclass(RandomFly, MetaEnv)
    assert RandomFly.cfg_from_args(['Banana', 1 / 120.]) == {"object_id": 0, "dt": 1 / 120.}
    assert RandomFly.cfg_from_args(['Amicelli', 1 / 120.]) == {"object_id": 1, "dt": 1 / 120.}
    with pytest.raises(ValueError):
        RandomFly.cfg_from_args(['Apple', 1 / 120.])
    # the second object of the generated table through the same facade: a different body flies (2 spheres instead of 5, other inertia)
    env2 = pih.make('peg-in-hole-mp-v0', client=None, task='random-fly', mp_num=1, sub_num=2, offset=[2., 3., 0.], args=['Amicelli', 1 / 120.],
                    is_test=True, backend_factory=factory)
    assert env2._backend.o.cfg.object_id == 1
    env2.reset(); o2, r2, d2, _ = env2.step(env2.action_space.sample())
    assert o2[0][1].shape == (6,) and np.isfinite(o2[0][1]).all()
    # README.md:38 usage, unchanged but for the import
    env = pih.make('peg-in-hole-mp-v0', client=None, task='random-fly', mp_num=2, sub_num=2, offset=[2., 3., 0.], args=['Banana', 1 / 120.],
                   is_test=True, backend_factory=factory)
    assert env.action_space.shape == (6,) and env.observation_space.shape == (6,)
    obs = env.reset()
    assert len(obs) == 2 and len(obs[0]) == 2 and obs[0][0].shape == (6,)
    assert abs(env._backend.o.cfg.dt - 1 / 120.) < 1e-12 and env._backend.task_id == 1
    obs, rew, done, info = env.step(env.action_space.sample())
    assert obs[1][1].shape == (6,) and isinstance(rew[0][0], float) and isinstance(done[0][0], bool)
    # world coordinates: the reference's grid rule (envs/base_env.py:35-55) puts agent (i, 1) at offset (0, 3, 0) from agent (i, 0)
    assert abs(obs[0][1][0] - obs[0][0][0]) < 0.2 and abs(obs[0][1][1] - obs[0][0][1] - 3.0) < 0.2
    with pytest.raises(AssertionError):
        pih.make('peg-in-hole-mp-v0', client=None, task='no-such-task', backend_factory=factory)
    # the MetaEnv contract for ONE agent (envs/meta_env.py:8-42): apply_action / get_info / reset
    t = RandomFly(client=None, offset=[0, 0, 0], args=['Banana', 1 / 120.], backend_factory=factory)
    t.reset(hard_reset=True)
    t.apply_action(np.zeros(6, dtype=np.float32))
    ob, r, d, inf = t.get_info()
    assert ob.shape == (6,) and r in (0.0, 1.0) and isinstance(d, bool) and inf == {}


def test_fly_exit_cadence_host_build_matches_same_cadence_oracle(oracle_mod):
    """the sampled exit-test cadence of the random-fly PGS (pih_config.exit_check_stride, default 16) on the fp64 host build: iteration
    counts identical to the oracle run at the same cadence, state equal to rounding; against Bullet's cadence 0 .. 16 more iterations"""
    O = oracle_mod
    n = 32
    kw = dict(seed=5, dt=DT, auto_reset=1, max_episode_steps=120)
    A = O.FlyOracle(n, **kw); B = O.FlyOracle(n, exit_check_stride=16, **kw)
    e = E.EmulFly(n, "f64", debug=1, exit_check_stride=16, **kw)
    rng = np.random.default_rng(2)
    dA = []; early = 0
    for t in range(200):
        a = rng.uniform(-1, 1, (n, 6))
        s = B.get_state(); A.set_state(s); e.set_state(s)
        A.step(a); _, _, dn = B.step(a); e.step(a)
        it = e.get_debug()[:, 13].astype(int)
        live = dn == 0
        np.testing.assert_array_equal(it[live], B.pgs_iters()[live])
        assert np.abs(e.get_state()[live][:, :31] - B.get_state()[live][:, :31]).max() < 5e-8      # (the fp64 bound of the one-step test above)
        dA.append((it - A.pgs_iters())[live]); early += int((B.pgs_iters()[live] < 50).sum())
    dA = np.concatenate(dA)
    assert early > 500 and dA.min() >= 0 and dA.max() <= 46 and (dA > 0).any()


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_fly_quad_layout_host_build(oracle_mod, prec):
    """One env per QUAD of lanes (pih_fly.h Q::QUAD; the GPU layout up to 4096 envs), host build with four lockstep threads per env: the four
    lanes of every quad end each step with bit-identical records; against the one-env-per-lane build of the same source only the association
    of the sums differs (fp64: 1e-9 one-step, the bound of the lane build against the oracle where an arm link is pressed into the table);
    contact counts, iteration counts, rewards and done flags are identical; against the oracle as the lane build."""
    O = oracle_mod
    n = 12
    kw = dict(seed=4, dt=DT, auto_reset=1, max_episode_steps=120, exit_check_stride=16)
    o = O.FlyOracle(n, **kw); a = E.EmulFly(n, prec, debug=1, **kw); b = E.EmulFly(n, prec, debug=1, **kw)
    rng = np.random.default_rng(6)
    ncs = 0; err_l = []; err_o = []; its = 0
    for t in range(150):
        act = rng.uniform(-1, 1, (n, 6))
        s = o.get_state(); a.set_state(s); b.set_state(s)
        oo, ro, do = o.step(act); oa, ra, da = a.step(act); ob, rb, db, bad = b.step_quad(act)
        assert bad == 0
        sa, sb, so = a.get_state(), b.get_state(), o.get_state()
        np.testing.assert_array_equal(da, db); np.testing.assert_array_equal(ra, rb); np.testing.assert_array_equal(sa[:, 44], sb[:, 44])
        live = (do == 0) & (da == 0)
        ia, ib = a.get_debug()[:, 13], b.get_debug()[:, 13]
        its += int((ia[live] != ib[live]).sum())
        ncs += int(sb[:, 44].sum())
        err_l.append(np.abs(sa[live][:, :31] - sb[live][:, :31]).max(initial=0)); err_o.append(np.abs(so[live][:, :31] - sb[live][:, :31]).max(initial=0))
        np.testing.assert_allclose(oa, ob, atol=1e-9 if prec == "f64" else 2e-4)
    print(prec, "quad vs lane layout max %.2e, quad vs oracle max %.2e median %.2e; contact env-steps %d; iteration counts differing %d" % (max(err_l), max(err_o), np.median(err_o), ncs, its))
    assert ncs > 100
    assert max(err_l) < (5e-8 if prec == "f64" else 2e-3) and max(err_o) < (5e-8 if prec == "f64" else 2e-3)
    assert np.median(err_o) < (2e-9 if prec == "f64" else 5e-5) and its <= 2


def test_fly_quad_layout_more_contacts_than_register_records(oracle_mod):
    """the quad layout keeps the first 8 contact records of an env in registers and sweeps a 9th .. 15th from lane memory: states with
    5 .. 14 contacts (arm lying on the table, object under the hand), host build in both layouts and the oracle"""
    from tests import parity_util as P
    O = oracle_mod
    n = 48
    kw = dict(seed=4, dt=DT, auto_reset=0, max_episode_steps=100000, exit_check_stride=16, contact_margin=0.02)
    o = O.FlyOracle(n, **kw); a = E.EmulFly(n, "f64", debug=1, **kw); b = E.EmulFly(n, "f64", debug=1, **kw)
    s = P.fly_many_contact_states(o, n, seed=1)
    o.set_state(s); a.set_state(s); b.set_state(s)
    big = 0
    for t in range(6):
        act = np.zeros((n, 6)); act[:, :3] = [0.3, 0.0, 0.3]
        o.set_state(a.get_state()); b.set_state(a.get_state())
        o.step(act); a.step(act); _, _, _, bad = b.step_quad(act)
        assert bad == 0
        sa, sb, so = a.get_state(), b.get_state(), o.get_state()
        np.testing.assert_array_equal(sa[:, 44], sb[:, 44]); np.testing.assert_array_equal(sa[:, 44], so[:, 44])
        big += int((sb[:, 44] > 8).sum())
        np.testing.assert_array_equal(a.get_debug()[:, 13], b.get_debug()[:, 13])
        assert np.abs(sa[:, :31] - sb[:, :31]).max() < 1e-9 and np.abs(so[:, :31] - sb[:, :31]).max() < 1e-6
        assert np.abs(sa[:, 43] - sb[:, 43]).max() < 1e-6 * (1 + np.abs(sa[:, 43]).max())         # the summed normal force
    assert big >= 30


def test_fly_limit_rows_speculation_is_exact(oracle_mod):
    """The joint-limit rows are skipped for joints farther than 0.25 rad from their limits and the skipped rows' right-hand sides verified;
    a violated one repeats the env's solve with every row (pih_fly.h).  Cases: (a) joints inside the 0.25-rad band (all rows from the
    start, debug word 14 = 1), (b) joints 0.26 .. 0.29 rad away running at 60 rad/s into the limit (the verification fires: 2), (c) far away (0):
    both layouts of the host build against the oracle, which always sweeps all rows."""
    O = oracle_mod
    n = 12
    kw = dict(seed=2, dt=DT, auto_reset=0, max_episode_steps=100000, exit_check_stride=16)
    o = O.FlyOracle(n, **kw); a = E.EmulFly(n, "f64", debug=1, **kw); b = E.EmulFly(n, "f64", debug=1, **kw)
    s = o.get_state()
    lo = np.full(6, -np.pi); hi = -lo                                           # (ur5.urdf joint limits, include/pih_model.h PIH_UR5_LO / HI)
    for e in range(n):
        j = e % 6
        if e < 4:   s[e, j] = hi[j] - 0.2; s[e, 6 + j] = 3.0                     # (a) inside the band, moving into the limit
        elif e < 8: s[e, 2] = lo[2] + 0.26 + 0.01 * (e - 4); s[e, 8] = -60.0           # (b) the elbow outside the band, reaches the limit within the step
        else:       s[e, 6 + j] = 5.0                                            # (c) at the rest pose
    seen = set()
    for t in range(4):
        o.set_state(s); a.set_state(s); b.set_state(s)
        act = np.zeros((n, 6)); act[:, :3] = [0.3, 0.1, 0.4]
        o.step(act); a.step(act); _, _, _, bad = b.step_quad(act)
        assert bad == 0
        da, db = a.get_debug(), b.get_debug()
        np.testing.assert_array_equal(da[:, 14], db[:, 14]); np.testing.assert_array_equal(da[:, 13], db[:, 13])
        if t == 0:
            assert (da[:4, 14] == 1).all() and (da[4:8, 14] == 2).all() and (da[8:, 14] == 0).all()
        seen |= set(da[:, 14].astype(int).tolist())
        sa, sb, so = a.get_state(), b.get_state(), o.get_state()
        assert np.abs(sa[:, :31] - so[:, :31]).max() < 1e-9 * 100 and np.abs(sb[:, :31] - so[:, :31]).max() < 1e-9 * 100     # (velocities up to 100 rad/s)
        np.testing.assert_array_equal(a.get_debug()[:, 13], o.pgs_iters())
        s = so
    assert seen == {0, 1, 2}
    # the limit did hold: no joint beyond its limit by more than the one-step overshoot the ERP removes
    assert (s[:8, :6] <= hi + 0.05).all() and (s[:8, :6] >= lo - 0.05).all()
