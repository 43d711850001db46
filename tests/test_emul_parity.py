"""CPU-side checks of the PRODUCT's device algorithm (pih_device.h compiled for the host by tests/emul, test-only) against
the fp64 oracle: in double the articulated-body / impulse-response / on-the-fly-Jacobian path must agree with the oracle's
RNEA + dense-Cholesky path to rounding (two independent derivations of the same physics); in float it shows the fp32
sensitivity the GPU will have.  The real parity tests (through the C ABI, on the GPU) are in test_gpu_parity.py."""
import numpy as np
import pytest

from tests.emul import emul as E

REST = np.array([0, -0.215, -np.pi / 3, -2.57, 0, 2.356, 2.356, 0, 0])
POS = [*range(0, 9), *range(18, 25), *range(31, 54)]
VEL = [*range(9, 18), *range(25, 31), *range(54, 77)]


@pytest.fixture(scope="module", autouse=True)
def _build():
    E.build()


def test_reset_bit_identical_draws(oracle_mod):
    o = oracle_mod.Oracle(64, seed=7)
    for prec, tol in (("f64", 0.0), ("f32", 1e-6)):
        e = E.Emul(64, prec, seed=7)
        np.testing.assert_allclose(e.get_state()[:, :98], o.get_state()[:, :98], rtol=0, atol=tol)


def test_ik_matches_oracle(oracle_mod):
    rng = np.random.default_rng(0)
    tq = oracle_mod.quat_from_euler([0, -np.pi, 0])
    for _ in range(10):
        q0 = REST + np.concatenate([rng.uniform(-0.3, 0.3, 7), [0, 0]])
        p, _ = oracle_mod.fk_arm(q0, 9)
        tgt = p + rng.uniform(-0.02, 0.02, 3)
        ref = oracle_mod.ik(q0, tgt, tq)
        np.testing.assert_allclose(E.ik(q0, tgt, tq, "f64"), ref, atol=1e-10)
        np.testing.assert_allclose(E.ik(q0, tgt, tq, "f32"), ref, atol=2e-5)


def test_one_step_equivalence_f64(oracle_mod):
    """Resynchronised every step (so chaos cannot amplify): ABA == RNEA+Cholesky, response rows == M^-1 J^T, PGS identical."""
    N = 8
    o = oracle_mod.Oracle(N, residual_threshold=0.0, warmstart=0.0)
    e = E.Emul(N, "f64", residual_threshold=0.0, warmstart=0.0, debug=1)
    rng = np.random.default_rng(0)
    for t in range(250):
        a = rng.uniform(-1, 1, (N, 4))
        so = o.get_state(); se = e.get_state(); se[:, :98] = so[:, :98]; se[:, 128] = 0; e.set_state(se)
        oo, ro, do = o.step(a); oe, re, de = e.step(a)
        so = o.get_state(); se = e.get_state()
        ud = np.array([o.debug_udot(i) for i in range(N)])
        assert np.abs(ud - e.get_debug()[:, :38]).max() <= 1e-6 * (1 + np.abs(ud).max())
        # rounding-level agreement (median far below); the bound leaves room for steps with a loaded mu = 10 end-link contact,
        # where pyramid-friction PGS amplifies rounding differences
        assert np.abs(so[:, POS] - se[:, POS]).max() < 1e-6 and np.median(np.abs(so[:, POS] - se[:, POS])) < 1e-12
        assert np.abs(so[:, VEL] - se[:, VEL]).max() < 2e-4
        np.testing.assert_array_equal(o.ncontacts(), se[:, 106].astype(int))
        np.testing.assert_allclose(o.contact_force(), se[:, 105], atol=1e-5 * (1 + np.abs(o.contact_force()).max()))
        np.testing.assert_allclose(oo, oe, atol=1e-7)
        np.testing.assert_array_equal(do, de)


def _stable_scenario(oracle_mod, prec, bent):
    N = 4
    kw = dict(residual_threshold=0.0)
    o = oracle_mod.Oracle(N, **kw); e = E.Emul(N, prec, **kw)
    p0, _ = oracle_mod.fk_arm(REST, 9)
    if bent:
        a = np.tile([p0[0], p0[1], p0[2], 0.0], (N, 1))
        for _ in range(1000):
            o.step(a)
        s = o.get_state(); s[:, 25:31] = 0; s[:, 54:77] = 0; o.set_state(s)
    else:
        s = o.get_state(); s[:, 31:54] = 0; s[:, 20] = -0.04 + 1e-4; o.set_state(s)
    se = e.get_state(); se[:, :98] = s[:, :98]; se[:, 128] = 0; e.set_state(se)
    maxd = maxo = 0.0
    from tests import parity_util as P
    led = P.ForceLedger(oracle_mod, slots=64, **kw)
    for t in range(1000):
        ph = 2 * np.pi * t / 500.0
        a = np.tile([p0[0] + 0.1 * np.sin(ph), p0[1] + 0.1 * np.cos(ph) - 0.1, p0[2] + 0.05 * np.sin(2 * ph), 0.02], (N, 1))
        led.before(o)
        oo, _, _ = o.step(a); oe, _, _ = e.step(a)
        se = e.get_state()
        maxd = max(maxd, np.abs(o.tip_pose()[:, :3] - se[:, 98:101]).max())
        maxo = max(maxo, np.abs(oo - oe).max())
        led.after(a, o.contact_force(), se[:, 105])
    return maxd, led, maxo


@pytest.mark.parametrize("bent", [False, True])
def test_trajectory_parity_contact_stable(oracle_mod, bent):
    """north_star tolerance: peg-tip pose within 1e-3 m and contact-normal force within 1e-2 N over 1000 steps on
    identical seeds, contact-stable scenarios (pipe resting on the table while the arm tracks a smooth target)."""
    d, led, ob = _stable_scenario(oracle_mod, "f32", bent)
    # force: 1e-2 N on EVERY step where the fp64 oracle's own force is stable under a 1e-6 / 1e-5 perturbation of its input, and on the
    # 16-step mean everywhere; transients are classified by probes and bounded by the probes' deviation (tests/parity_util.py ForceLedger)
    led.finish("host build fp32 trajectory bent=%s" % bent)
    assert d < 1e-3 and ob < 1e-3, (d, ob)
    d, led, ob = _stable_scenario(oracle_mod, "f64", bent)
    r = led.finish("host build fp64 trajectory bent=%s" % bent)
    assert d < 1e-4 and r["avg"] < 5e-3 and r["calm_max"] < 5e-3 and ob < 1e-6, (d, r, ob)


def test_frozen_done_and_auto_reset(oracle_mod):
    N = 4
    for auto in (0, 1):
        # (arm-vs-pipe spheres off: a pipe that respawns through the hand is ejected violently, and an un-resynchronised rollout of
        # two fp64 implementations then diverges chaotically; this test is about the done / reset logic)
        o = oracle_mod.Oracle(N, max_episode_steps=5, auto_reset=auto, residual_threshold=0.0, enable_arm_collision=1)
        e = E.Emul(N, "f64", max_episode_steps=5, auto_reset=auto, residual_threshold=0.0, enable_arm_collision=1)
        rng = np.random.default_rng(2)
        for t in range(12):
            a = rng.uniform(-1, 1, (N, 4))
            oo, ro, do = o.step(a); oe, re, de = e.step(a)
            np.testing.assert_array_equal(do, de)
            np.testing.assert_allclose(oo, oe, atol=1e-6)
            so, se = o.get_state(), e.get_state()
            np.testing.assert_allclose(so[:, POS], se[:, POS], atol=1e-4)   # a self-contact (near-parallel capsules) amplifies rounding
            np.testing.assert_allclose(so[:, VEL], se[:, VEL], atol=1e-3)
            np.testing.assert_array_equal(so[:, 86:98], se[:, 86:98])
        if auto == 0:
            assert do.all() and o.get_state()[0, 93] == 5       # frozen after done (envs/base_env.py:62,66)
        else:
            assert o.get_state()[0, 93] == 12 % 5


def test_arm_table_contact_f64(oracle_mod):
    """Arm collision spheres vs the table (keys 3000+): drive the gripper down into the table.  The device algorithm (arm
    link as linkA, no pipe response) must match the oracle step by step, and the finger tips must stop at the table."""
    N = 3
    kw = dict(residual_threshold=0.0, warmstart=0.0)
    o = oracle_mod.Oracle(N, **kw); e = E.Emul(N, "f64", debug=1, **kw)
    p0, _ = oracle_mod.fk_arm(REST, 9)
    seen = 0
    lowest = 1.0
    for t in range(420):
        a = np.tile([p0[0], p0[1] - 0.25, -1.0, 0.04], (N, 1))     # target far below the table, away from the pipe's spawn area... y in [-0.6,-0.4]
        a[:, 0] += 0.25
        so = o.get_state(); se = e.get_state(); se[:, :98] = so[:, :98]; se[:, 128] = 0; e.set_state(se)
        o.step(a); e.step(a)
        so = o.get_state(); se = e.get_state()
        keys = [int(k) for k in o.debug_contacts(0)[:, 10]]
        seen += any(k >= 3000 for k in keys)
        assert np.abs(so[:, POS] - se[:, POS]).max() < 1e-7, t
        assert np.abs(so[:, VEL] - se[:, VEL]).max() < 5e-5 * (1 + np.abs(so[:, VEL]).max()), t
        np.testing.assert_array_equal(o.ncontacts(), se[:, 106].astype(int))
        ee, _ = oracle_mod.fk_arm(so[0, 0:9], 9)
        lowest = min(lowest, ee[2])
    assert seen > 50                       # the arm reached the table and stayed in contact
    # grasp target is 7 mm above the finger-tip sphere bottoms; the table is at -0.05: the EE cannot sink below it
    assert lowest > -0.05 - 0.004, lowest
    # without arm collision the same command drives the gripper through the table
    o2 = oracle_mod.Oracle(1, enable_arm_collision=0, **kw)
    for t in range(420):
        o2.step(a[:1])
    assert oracle_mod.fk_arm(o2.get_state()[0, 0:9], 9)[0][2] < -0.08


def test_joint_limit_rows_f64(oracle_mod):
    """Limit rows active (joints thrown at their limits at 50 rad/s in scripted state 0, where only the weak load-time motors
    hold the arm): device algorithm == oracle, and the closed form pen / dt of the first step."""
    N = 2
    kw = dict(residual_threshold=0.0, warmstart=0.0, enable_self_collision=0, mode=1, dv=0.05)
    o = oracle_mod.Oracle(N, **kw); e = E.Emul(N, "f64", **kw)
    s = o.get_state()
    s[:, 2] = 2.9671 - 0.02; s[:, 11] = 50.0; s[1, 3] = -0.01; s[1, 12] = 30.0; s[:, 18] = 5.0
    o.set_state(s)
    for t in range(20):
        so = o.get_state(); se = e.get_state(); se[:, :98] = so[:, :98]; se[:, 128] = 0; e.set_state(se)
        o.step(np.zeros((N, 4))); e.step(np.zeros((N, 4)))
        so = o.get_state(); se = e.get_state()
        assert np.abs(so[:, POS] - se[:, POS]).max() < 1e-7 and np.abs(so[:, VEL] - se[:, VEL]).max() < 2e-5
        if t == 0:
            assert abs(se[0, 11] - 4.8) < 1e-4 and abs(se[1, 12] - 2.4) < 1e-4
    assert abs(se[0, 2] - 2.9671) < 1e-5 and abs(se[1, 3]) < 1e-5


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_many_contacts_spill_rows(oracle_mod, prec):
    """> 20 and > 32 simultaneous contacts (rows / records of contacts 21..48 live in the global spill area): pipes coiled flat
    on the table, gripper coming down on the coil (scripted mode).  Contact keys, per-contact normal impulse, pose and force
    against the oracle, resynchronised every checked step; the counts reached are asserted.  Steps with a loaded mu = 10
    finger contact (ill-conditioned for PGS, tests/scenarios.py) are reported separately."""
    from tests.scenarios import coil_pipe_flat, stiff_finger_contact
    N = 4
    kw = dict(mode=1, dv=0.05, residual_threshold=0.0, warmstart=0.0)
    o = oracle_mod.Oracle(N, **kw); e = E.Emul(N, prec, debug=1, **kw)
    s0 = oracle_mod.Oracle(8, **kw).get_state()
    o.set_state(coil_pipe_flat(s0)[4:8])
    a = np.zeros((N, 4))
    seen = np.zeros(49, int); perr, ferr, lerr, stiff = [], [], [], []
    for t in range(1100):
        check = t < 40 or 640 <= t < 700 or 1021 <= t < 1100
        if check:
            so = o.get_state(); se = e.get_state(); se[:, :98] = so[:, :98]; se[:, 128] = 0; e.set_state(se)
        o.step(a)
        if not check:
            continue
        e.step(a)
        so = o.get_state(); se = e.get_state()
        nco = o.ncontacts()
        np.testing.assert_array_equal(nco, se[:, 106].astype(int))
        dbg = e.get_debug()
        for i in range(N):
            oc = o.debug_contacts(i); k = len(oc)
            gc = dbg[i, 40:40 + 12 * k].reshape(k, 12)
            np.testing.assert_array_equal(oc[:, 10], gc[:, 10])
            st = stiff_finger_contact(oc); stiff.append(st)
            if not st:
                seen[k] += 1
            lerr.append(np.abs(oc[:, 11] - gc[:, 11]).max() / (1e-3 + np.abs(oc[:, 11]).max()))
        perr.append(np.abs(so[:, POS] - se[:, POS]).max(1))
        cf = o.contact_force(); ferr.append(np.abs(se[:, 105] - cf) / (1 + np.abs(cf)))
    perr = np.concatenate(perr); ferr = np.concatenate(ferr); lerr = np.array(lerr); stiff = np.array(stiff)
    ok = ~stiff
    print(prec, "well-conditioned env-steps: max contacts", seen.nonzero()[0].max(), ">20:", seen[21:].sum(), ">32:", seen[33:].sum(),
          "pose p50/p99/max %.2e %.2e %.2e force %.2e %.2e %.2e lam %.2e %.2e | stiff env-steps %d: pose p50 %.2e max %.2e" % (
              np.percentile(perr[ok], 50), np.percentile(perr[ok], 99), perr[ok].max(), np.percentile(ferr[ok], 50), np.percentile(ferr[ok], 99),
              ferr[ok].max(), np.percentile(lerr[ok], 50), np.percentile(lerr[ok], 99), stiff.sum(), np.percentile(perr[stiff], 50), perr[stiff].max()))
    assert seen[21:].sum() > 150 and seen[33:].sum() > 50
    if prec == "f64":
        assert np.percentile(perr[ok], 99) < 1e-7 and np.percentile(ferr[ok], 99) < 1e-6 and np.percentile(lerr[ok], 99) < 1e-5
    else:
        assert np.percentile(perr[ok], 50) < 5e-6 and np.percentile(perr[ok], 99) < 1e-4
        assert np.percentile(ferr[ok], 50) < 1e-3 and np.percentile(ferr[ok], 99) < 1e-2


def test_operation_counting_build_counts_and_computes_the_same():
    """tools/count_flops.py's CountedReal build of the algorithm (the source of profiles/flops_latest.json) must compute exactly
    what the plain fp64 build computes, and report a plausible operation count per env-step."""
    import ctypes as C, subprocess, os
    subprocess.check_call(["make", "-C", os.path.dirname(E.__file__), "-s", "count"])
    N = 4
    a = E.Emul(N, "f64", seed=3); c = E.Emul(N, "cnt", seed=3)
    c.L.emul_flops_reset.restype = None
    c.L.emul_flops_get.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    rng = np.random.default_rng(0)
    c.L.emul_flops_reset()
    for t in range(60):
        act = rng.uniform(-1, 1, (N, 4))
        a.step(act); c.step(act)
    np.testing.assert_array_equal(a.get_state(), c.get_state())
    ph = (C.c_ulonglong * 96)(); tot = (C.c_ulonglong * 6)()
    c.L.emul_flops_get(ph, tot)
    per = sum(tot[:5]) / (N * 60)
    assert 1e5 < per < 3e6 and sum(ph[6 * 5:6 * 5 + 5]) > 0.3 * sum(tot[:5])     # PGS is the bulk


def test_defaults_parity_on_host_build(oracle_mod):
    """The check of tests/test_gpu_defaults.py (library defaults: residual_threshold 1e-7, warmstart 0.85, exit_check_stride 16; oracle
    at Bullet's cadence and at the product's cadence; every env-step bounded, tests/parity_util.py) on the fp32 HOST build of the product
    algorithm -- the same assertions the HIP library must pass on the GPU box, exercised in the CPU suite."""
    from tests import parity_util as P
    E.build()
    N = 192
    g = E.Emul(N, "f32", seed=5, exit_check_stride=16)
    assert abs(g.cfg.residual_threshold - 1e-7) < 1e-12 and abs(g.cfg.warmstart - 0.85) < 1e-6
    P.defaults_one_step_check("host build (fp32) at the library defaults", oracle_mod, g, N, 130)


def test_exit_cadence_of_the_oracle(oracle_mod):
    """piho_config.exit_check_stride: stride 16 tests the residual in iterations 1..4, 20, 36 and 50 only -- from identical states it runs
    at least as many iterations as Bullet's cadence, and exits only in a tested iteration; the states differ at the threshold level."""
    N = 64
    A = oracle_mod.Oracle(N, omp=True, seed=9); B = oracle_mod.Oracle(N, omp=True, seed=9, exit_check_stride=16)
    rng = np.random.default_rng(1)
    seen_early = 0
    for t in range(60):
        a = rng.uniform(-1, 1, (N, 4))
        B.set_state(A.get_state()); B.set_warm_cache(A.warm_cache())
        A.step(a); B.step(a)
        ia, ib = A.pgs_iters(), B.pgs_iters()
        assert (ib >= ia).all() and np.isin(ib, [1, 2, 3, 4, 20, 36, 50]).all()
        seen_early += int((ia < 50).sum())
        assert np.abs(A.get_state()[:, :77] - B.get_state()[:, :77]).max() < 1e-2
    assert seen_early > 100
