"""GPU parity tests of the 'random-fly' task (BASELINE configs[4]: UR5 + free-flying object, 4096 envs): the HIP path (one env
per lane, pih_fly.h) through the C ABI against the fp64 oracle on the same seeded inputs.  Tolerances: positions 1e-3 m / forces
1e-2 N relative as in north_star; one-step errors are asserted far tighter.  PARITY UNPINNED vs PyBullet."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = 1.0 / 120.0


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch


def _gpu(n, **kw):
    from peg_in_hole_gym_amd.vec_env import PihVecEnv
    kw.setdefault("max_episode_steps", 480); kw.setdefault("contact_margin", 0.02); kw.setdefault("dt", DT)
    return PihVecEnv(n, task_id=1, **kw)


def test_fly_reset_and_state_roundtrip(torch_mod, oracle_mod):
    torch = torch_mod
    n = 70                                   # two waves, ragged
    g = _gpu(n, seed=3); o = oracle_mod.FlyOracle(n, seed=3, dt=DT)
    sg = g.state().cpu().numpy()
    assert sg.shape == (n, 48)
    np.testing.assert_allclose(sg, o.get_state(), atol=2e-6)
    assert g.obs.shape == (n, 6)
    np.testing.assert_allclose(g.ee_position().cpu().numpy(), sg[:, 40:43], atol=0)
    s2 = torch.tensor(sg) + 0.0; s2[:, 18] += 0.25
    g.set_state(s2)
    assert torch.equal(g.state().cpu(), s2)                        # env-major <-> structure-of-arrays transposition is exact
    mask = torch.zeros(n, dtype=torch.uint8); mask[::4] = 1
    g.reset(mask)
    s3 = g.state().cpu()
    assert torch.equal(s3[mask == 0], s2[mask == 0]) and (s3[mask == 1][:, 33] > s2[mask == 1][:, 33]).all()
    g.reset(hard_reset=True)
    s4 = g.state().cpu()
    assert not torch.equal(s4[:, 18:21], torch.tensor(sg)[:, 18:21]) and (s4[:, 33] > s3[:, 33]).all()   # hard reset: a NEW scene (draw counter advances)
    g.reset(hard_reset=True, seed=int(g.cfg.seed))
    np.testing.assert_array_equal(g.state().cpu().numpy(), sg)   # explicit replay (seed != 0): the seed's first scene bit for bit


@pytest.mark.parametrize("N,steps,obj", [(4096, 200, 0), (70, 400, 0), (1024, 200, 1)])
def test_fly_one_step_parity_resynchronised(torch_mod, oracle_mod, N, steps, obj):
    """Random-action episodes with auto-reset (throws, arm swings, arm-object and object-table contacts, landings, catches), the GPU
    state overwritten with the oracle's before every step.  An env-step whose done flag, reward or contact count differs from the oracle's
    took another DISCRETE branch (the object within float rounding of a contact / catch / landing threshold): those are counted and their
    share is bounded (5e-5).  EVERY other env-step is bounded (tests/parity_util.py ConditionedParity): position and velocity x dt within
    1e-3, or within 10 x what the fp64 oracle itself does under 1e-6 perturbations of that step's input (an object sphere equally deep in
    two neighbouring capsules picks one or the other: a discrete tie that the contact count does not show)."""
    torch = torch_mod
    kw = dict(seed=7, dt=DT, residual_threshold=0.0, auto_reset=1, max_episode_steps=150, object_id=obj)      # obj: 0 'Banana', 1 'Amicelli'
    o = oracle_mod.FlyOracle(N, omp=N > 256, **kw); g = _gpu(N, debug=1 if N <= 256 else 0, **kw)
    from tests import parity_util as P
    pk = dict(kw); pk["auto_reset"] = 0       # (the probes re-run single env-steps: no reset inside them)
    led = P.ConditionedParity(oracle_mod, task="random-fly", slots=256, **pk)
    rng = np.random.default_rng(0)
    perr, verr, ferr = [], [], []; ncs = 0; nrew = 0; mism = 0
    for t in range(steps):
        a = rng.uniform(-1, 1, (N, 6))
        g.set_state(torch.tensor(o.get_state(), dtype=torch.float32)); led.before(o)
        oo, ro, do = o.step(a)
        og, rg, dg = g.step(torch.tensor(a, dtype=torch.float32))
        so = o.get_state(); sg = g.state().cpu().numpy().astype(np.float64)
        # an env whose object passes within float rounding of a contact / catch / landing threshold, or whose sphere is equally deep
        # in two neighbouring capsules, takes a different discrete branch than the oracle: counted, bounded, and left out of the stats
        same = (do == dg.cpu().numpy()) & (so[:, 44] == sg[:, 44]) & (ro == rg.cpu().numpy())
        mism += int((~same).sum())
        ncs += int(so[:, 44].sum()); nrew += int(ro.sum())
        perr.append(np.abs(so[same][:, [*range(0, 6), *range(18, 25)]] - sg[same][:, [*range(0, 6), *range(18, 25)]]).max(1))
        verr.append(np.abs(so[same][:, [*range(6, 12), *range(25, 31)]] - sg[same][:, [*range(6, 12), *range(25, 31)]]).max(1))
        ferr.append(np.abs(so[same][:, 43] - sg[same][:, 43]) / (1 + np.abs(so[same][:, 43])))
        # every env-step that took the oracle's branch and did not finish (a finished env is reset inside the step) goes into the ledger:
        # error = position words, and the velocity error over one dt
        PW = [*range(0, 6), *range(18, 25)]; VW = [*range(6, 12), *range(25, 31)]
        live = same & (do == 0)
        e_all = np.maximum(np.abs(so[:, PW] - sg[:, PW]).max(1), DT * np.abs(so[:, VW] - sg[:, VW]).max(1))
        led.after(o, a, np.where(live, e_all, 0.0), np.where(live, np.abs(so[:, 43] - sg[:, 43]) / (1 + np.abs(so[:, 43])), 0.0))
        oerr = np.abs(og.cpu().numpy() - oo).max(1)
        assert oerr[same & (e_all < 1e-4)].max() < 2e-4            # the observation follows the state
    led.finish("fly one-step resynchronised N=%d object %d" % (N, obj), p50=2e-6, p99=2e-5, exempt_share=0.02)
    perr = np.concatenate(perr); verr = np.concatenate(verr); ferr = np.concatenate(ferr)
    print("fly N=%d: %d env-steps, %d with contacts, %d catches, %d threshold flips; pose err p50/p99/max %.2e / %.2e / %.2e ; velocity err p50/p99/max %.2e / %.2e / %.2e ; force rel err p99 %.2e" % (
        N, N * steps, ncs, nrew, mism, np.percentile(perr, 50), np.percentile(perr, 99), perr.max(), np.percentile(verr, 50), np.percentile(verr, 99), verr.max(), np.percentile(ferr, 99)))
    assert ncs > (20000 if N > 256 else 500) and mism <= 5e-5 * N * steps + 2
    assert np.percentile(perr, 50) < 2e-6 and np.percentile(perr, 99) < 2e-5
    assert np.percentile(verr, 50) < 1e-4 and np.percentile(verr, 99) < 5e-3
    assert np.percentile(ferr, 99) < 1e-2


def test_fly_free_flight_trajectory_and_arm_tracking(torch_mod, oracle_mod):
    """Trajectory parity without resynchronisation over 300 steps where the dynamics are smooth: object in damped free flight far from
    the arm (closed form) while the arm tracks a slow circular end-effector target; peg... object pose and ee within 1e-3 m."""
    torch = torch_mod
    O = oracle_mod
    N = 64
    kw = dict(seed=2, dt=DT, residual_threshold=0.0, auto_reset=0, max_episode_steps=100000)
    o = O.FlyOracle(N, **kw); g = _gpu(N, **kw)
    s = o.get_state()
    s[:, O.F_OPOS:O.F_OPOS + 3] = [3.0, 3.0, 40.0]; s[:, O.F_OVLIN:O.F_OVLIN + 3] = [1.0, -2.0, 3.0]; s[:, O.F_OVANG:O.F_OVANG + 3] = [0.5, 1.0, -2.0]
    o.set_state(s); g.set_state(torch.tensor(s, dtype=torch.float32))
    rest = np.array([0, -np.pi / 2, np.pi / 2, -np.pi / 2, -np.pi / 2, 0])
    ee0, qe = O.fk_ur5(rest, 6); eul = O.euler_from_quat(qe)
    p = np.array([3.0, 3.0, 40.0]); v = np.array([1.0, -2.0, 3.0])
    md = 0.0
    for t in range(300):
        ph = 2 * np.pi * t / 300
        a = np.tile(np.r_[ee0 + 0.08 * np.array([np.sin(ph), np.cos(ph) - 1, 0.5 * np.sin(2 * ph)]), eul], (N, 1))
        oo, _, _ = o.step(a); og, _, _ = g.step(torch.tensor(a, dtype=torch.float32))
        v = v + DT * (np.array([0, 0, -9.8]) - 0.04 * (1 + np.linalg.norm(v)) * v); p = p + DT * v
        md = max(md, np.abs(oo - og.cpu().numpy()).max())
    sg = g.state().cpu().numpy()
    print("fly trajectory parity: obs max diff %.2e over 300 steps" % md)
    assert md < 1e-3
    np.testing.assert_allclose(sg[:, 18:21], np.tile(p, (N, 1)), atol=2e-3)        # closed-form damped flight (fp32 accumulation over 300 steps at z ~ 40)
    np.testing.assert_allclose(sg[:, 25:28], np.tile(v, (N, 1)), atol=2e-4)


def test_fly_full_size_properties(torch_mod):
    """BASELINE configs[4] size (4096 envs, random actions, auto-reset): finite state, unit quaternions, joints inside their limits,
    episodes end (landing / catch) and restart, bitwise run-to-run determinism, results independent of the batch size."""
    torch = torch_mod
    N = 4096
    outs = []
    for rep in range(2):
        g = _gpu(N, auto_reset=1, seed=11)
        gen = torch.Generator(device="cuda").manual_seed(1234)
        ndone = 0
        for t in range(300):
            a = torch.rand(N, 6, device="cuda", generator=gen) * 2 - 1
            obs, rew, done = g.step(a)
            ndone += int(done.sum())
        st = g.state()
        outs.append(st.clone())
        assert torch.isfinite(st).all() and torch.isfinite(obs).all()
        assert torch.allclose(st[:, 21:25].norm(dim=1), torch.ones(N, device="cuda"), atol=1e-4)
        assert (st[:, 0:6].abs() <= 3.14159265359 + 0.05).all()
        assert ndone > 2 * N and st[:, 38].sum().item() == 0          # every env finished > 2 episodes; no non-finite resets
        assert (st[:, 32] < 300).all()
    assert torch.equal(outs[0], outs[1])
    gen = torch.Generator(device="cuda").manual_seed(5)
    acts = torch.rand(40, 130, 6, device="cuda", generator=gen) * 2 - 1
    a = _gpu(130, seed=3, auto_reset=1); b = _gpu(37, seed=3, auto_reset=1)
    for t in range(40):
        oa, _, da = a.step(acts[t]); ob, _, db = b.step(acts[t, :37].contiguous())
        assert torch.equal(oa[:37], ob) and torch.equal(da[:37], db)
    assert torch.equal(a.state()[:37], b.state())


def test_fly_facade_readme_usage_on_gpu(torch_mod):
    """README.md:38 of the reference, unchanged but for the import and client=None."""
    import peg_in_hole_gym_amd as peg_in_hole_gym
    env = peg_in_hole_gym.make('peg-in-hole-mp-v0', client=None, task='random-fly', mp_num=4, sub_num=4, offset=[2., 3., 0.],
                               args=['Banana', 1 / 120.], is_test=True)
    obs = env.reset()
    assert len(obs) == 4 and obs[0][0].shape == (6,)
    for _ in range(5):
        obs, reward, done, info = env.step(env.action_space.sample())
    assert np.isfinite(np.asarray(obs)).all() and len(done[3]) == 4
    assert abs(env._backend.cfg.dt - 1 / 120.) < 1e-9 and env._backend.task_id == 1
    env.close()
    # the other object of the generated table (asset -> task): Amicelli_800_tex.urdf, selected by args[0]
    env = peg_in_hole_gym.make('peg-in-hole-mp-v0', client=None, task='random-fly', mp_num=2, sub_num=4, offset=[2., 3., 0.],
                               args=['Amicelli', 1 / 120.], is_test=True)
    assert env._backend.cfg.object_id == 1
    obs = env.reset()
    for _ in range(5):
        obs, reward, done, info = env.step(env.action_space.sample())
    assert np.isfinite(np.asarray(obs)).all()
    env.close()
    from peg_in_hole_gym_amd.envs.peg_in_hole import RandomFly
    t = RandomFly(client=None, offset=[0, 0, 0], args=['Banana', 1 / 120.])
    t.reset(hard_reset=True); t.apply_action(np.zeros(6, dtype=np.float32))
    ob, r, d, inf = t.get_info()
    assert ob.shape == (6,) and np.isfinite(ob).all()
    t.close()


def test_fused_fly_launch_equals_ik_inside_the_step_wavefront():
    """Round 4: the random-fly step as ONE launch with the IK in controller wavefronts (mailbox + flag, targets read right before the PGS
    loop) against the IK inside the step wavefront (pih_config.schedule + 8: the same per-lane IK code) -- bit-identical states,
    observations, rewards and dones across auto-resets, in both layouts of the step wavefronts: one env per quad of lanes (default; fused
    while controller + step workgroups are resident together, n <= 13 104) and one env per lane (schedule + 32; fused up to 8 192 envs)."""
    import torch
    from peg_in_hole_gym_amd.vec_env import PihVecEnv
    for n, lane in ((4096, 0), (8192, 0), (12000, 0), (13000, 0), (4096, 32), (8192, 32), (10000, 32)):
        kw = dict(task_id=1, seed=3, dt=DT, auto_reset=1, max_episode_steps=60, contact_margin=0.02)
        a = PihVecEnv(n, schedule=1 + lane, **kw); b = PihVecEnv(n, schedule=1 + 8 + lane, **kw)
        gen = torch.Generator(device="cuda").manual_seed(9)
        for t in range(130):
            act = torch.rand(n, 6, device="cuda", generator=gen) * 2 - 1
            oa = [x.clone() for x in a.step(act)]; ob = b.step(act)
            for x, y in zip(oa, ob):
                assert torch.equal(x, y), "n %d layout %d step %d" % (n, lane, t)
        assert torch.equal(a.state(), b.state())
        a.set_timing(1); a.step(act); a.timing2()          # (-5 if a step wavefront ever timed out waiting for its controller wavefront)


@pytest.mark.parametrize("n", [5, 1000, 4096, 20000])
def test_fly_quad_layout_against_lane_layout(n):
    """One env per QUAD of lanes (the default: 16 envs per step wavefront, the PGS sweep split over the quad, pih_fly.h)
    against one env per LANE (schedule + 32), resynchronised every step over auto-resets: done flags, rewards, contact counts and
    iteration counts identical but for threshold ties (counted: <= 2e-5 of the env-steps), the state within the float association noise of
    the sums (p50 < 2e-6, p99.9 < 1e-3 one-step; the same figures the lane layout has against the oracle).  Ragged sizes: a last
    wavefront with 5 of 16 quads in use, a last controller group of 40 envs; 20 000 envs: beyond the fused launch (IK inside the quads)."""
    import torch
    from peg_in_hole_gym_amd.vec_env import PihVecEnv
    kw = dict(task_id=1, seed=3, dt=DT, auto_reset=1, max_episode_steps=90, contact_margin=0.02, debug=1 if n <= 1000 else 0)
    a = PihVecEnv(n, **kw); b = PihVecEnv(n, schedule=1 + 32, **kw)
    assert torch.equal(a.state(), b.state())
    gen = torch.Generator(device="cuda").manual_seed(5)
    errs = []; flips = 0; ncs = 0; its = 0
    PV = [*range(0, 6), *range(18, 25)]
    T = 200 if n <= 4096 else 60
    for t in range(T):
        act = torch.rand(n, 6, device="cuda", generator=gen) * 2 - 1
        b.set_state(a.state())
        oa, ra, da = [x.clone() for x in a.step(act)]; ob, rb, db = b.step(act)
        sa, sb = a.state(), b.state()
        same = (da == db) & (ra == rb) & (sa[:, 44] == sb[:, 44])
        flips += int((~same).sum()); ncs += int(sa[:, 44].sum())
        live = same & (da == 0)
        errs.append((sa[live][:, PV] - sb[live][:, PV]).abs().max(1).values.cpu())
        if kw["debug"]:
            its += int((a.debug()[live][:, 13] != b.debug()[live][:, 13]).sum())
    e = torch.cat(errs).numpy()
    print("   quad vs lane layout n=%d: %d env-steps, %d contacts, %d threshold flips, %d iteration counts differ; one-step pose difference p50 / p99.9 / max %.2e / %.2e / %.2e" % (
        n, len(e), ncs, flips, its, np.percentile(e, 50), np.percentile(e, 99.9), e.max()))
    assert flips <= 2e-5 * n * T + 1 and its <= 2e-3 * n * T + 1
    assert np.percentile(e, 50) < 2e-6 and np.percentile(e, 99.9) < 1e-3
    a.set_timing(1); a.step(act); a.timing2()


def test_fly_quad_layout_more_contacts_than_register_records(torch_mod, oracle_mod):
    """The quad layout keeps an env's first 8 contact records in registers and sweeps a 9th .. 15th from lane memory: 2 048 states with
    5 .. 14 contacts (arm lying on the table, the object under the hand) stepped by both layouts and by the oracle, resynchronised."""
    torch = torch_mod
    from tests import parity_util as P
    from peg_in_hole_gym_amd.vec_env import PihVecEnv
    n = 2048
    kw = dict(seed=4, dt=DT, auto_reset=0, max_episode_steps=100000, contact_margin=0.02)
    o = oracle_mod.FlyOracle(n, omp=True, exit_check_stride=16, **kw)
    a = PihVecEnv(n, task_id=1, **kw); b = PihVecEnv(n, task_id=1, schedule=1 + 32, **kw)
    s = P.fly_many_contact_states(o, n, seed=2)
    big = 0; eo = []; el = []
    act = np.zeros((n, 6)); act[:, :3] = [0.3, 0.0, 0.3]
    for t in range(8):
        o.set_state(s); a.set_state(torch.tensor(s, dtype=torch.float32)); b.set_state(torch.tensor(s, dtype=torch.float32))
        o.step(act); a.step(torch.tensor(act, dtype=torch.float32)); b.step(torch.tensor(act, dtype=torch.float32))
        sa = a.state().cpu().numpy().astype(np.float64); sb = b.state().cpu().numpy().astype(np.float64); so = o.get_state()
        same = (sa[:, 44] == so[:, 44]) & (sb[:, 44] == so[:, 44])
        assert (~same).sum() <= 2                               # (a contact within float rounding of the margin)
        big += int((sa[:, 44] > 8).sum())
        eo.append(np.abs(sa[same][:, :31] - so[same][:, :31]).max(1)); el.append(np.abs(sa[same][:, :31] - sb[same][:, :31]).max(1))
        s = so
    eo, el = np.concatenate(eo), np.concatenate(el)
    print("   many-contact states: %d env-steps with > 8 contacts; quad layout vs oracle p50 / p99 / max %.2e / %.2e / %.2e; vs lane layout %.2e / %.2e / %.2e" % (
        big, np.percentile(eo, 50), np.percentile(eo, 99), eo.max(), np.percentile(el, 50), np.percentile(el, 99), el.max()))
    assert big > 2000
    assert np.percentile(eo, 50) < 2e-5 and np.percentile(eo, 99) < 2e-3 and np.percentile(el, 99) < 2e-3


@pytest.mark.parametrize("sched", [1, 1 + 32])
def test_fly_limit_rows_speculation_is_exact(torch_mod, oracle_mod, sched):
    """The joint-limit rows of joints farther than 0.25 rad from their limits (in every env of the wavefront) are skipped and verified; a
    violated verification repeats the wavefront's solve with every row (pih_fly.h).  96 envs = 6 step wavefronts of the quad layout
    (sched 1; 2 of the lane layout, sched 33): wavefronts 0-1 joints inside the band (debug word 14 = 1), 2-3 the elbow 0.26 .. 0.28 rad
    away at -60 rad/s (reaches the limit within the step: 2 in the envs that do and in their wavefront neighbours), 4-5 at rest (0);
    state against the oracle, which always sweeps every row, and iteration counts equal."""
    torch = torch_mod
    n = 96 if sched == 1 else 192
    per = n // 3
    kw = dict(seed=2, dt=DT, auto_reset=0, max_episode_steps=100000)
    o = oracle_mod.FlyOracle(n, exit_check_stride=16, **kw)
    g = _gpu(n, debug=1, schedule=sched, **kw)
    s = o.get_state()
    for e in range(n):
        j = e % 6
        if e < per:        s[e, j] = np.pi - 0.2; s[e, 6 + j] = 3.0
        elif e < 2 * per and e % 4 == 0: s[e, 2] = -np.pi + 0.26 + 0.01 * (e % 3); s[e, 8] = -60.0
        elif e >= 2 * per: s[e, 6 + j] = 5.0
    act = np.zeros((n, 6)); act[:, :3] = [0.3, 0.1, 0.4]
    for t in range(3):
        o.set_state(s); g.set_state(torch.tensor(s, dtype=torch.float32))
        o.step(act); g.step(torch.tensor(act, dtype=torch.float32))
        d = g.debug().cpu().numpy(); so = o.get_state(); sg = g.state().cpu().numpy().astype(np.float64)
        if t == 0:
            assert (d[:per, 14] == 1).all() and (d[per:2 * per, 14] == 2).all() and (d[2 * per:, 14] == 0).all(), d[:, 14]
        assert (d[:, 13] == o.pgs_iters()).mean() > 0.97
        assert np.abs(sg[:, :6] - so[:, :6]).max() < 2e-4 and np.abs(sg[:, 6:12] - so[:, 6:12]).max() < 2e-2       # (velocities up to 100 rad/s in float)
        s = so
    assert (s[:, :6] <= np.pi + 0.05).all() and (s[:, :6] >= -np.pi - 0.05).all()


def test_fly_defaults_exit_test_and_cadence(torch_mod, oracle_mod):
    """The random-fly step AT THE LIBRARY DEFAULTS (residual_threshold 1e-7, exit_check_stride 16: what bench.py --task random-fly runs): the
    PGS exit test is live, evaluated at the sampled cadence.  One-step resynchronised against the oracle at the SAME cadence: iteration
    counts equal (but for residuals sitting on the threshold), every env-step bounded by the ledger; and against the oracle at Bullet's
    cadence (test after every iteration): the product runs 0 .. 15 further iterations, each below the threshold."""
    torch = torch_mod
    from tests import parity_util as P
    N, steps = 256, 300
    kw = dict(seed=11, dt=DT, auto_reset=1, max_episode_steps=150)
    A = oracle_mod.FlyOracle(N, omp=True, **kw)                               # Bullet's cadence
    B = oracle_mod.FlyOracle(N, omp=True, exit_check_stride=16, **kw)          # the product's
    g = _gpu(N, debug=1, **kw)
    assert g.cfg.exit_check_stride == 16 and abs(g.cfg.residual_threshold - 1e-7) < 1e-12
    pk = dict(kw); pk["auto_reset"] = 0; pk["exit_check_stride"] = 16
    led = P.ConditionedParity(oracle_mod, task="random-fly", slots=256, **pk)
    rng = np.random.default_rng(3)
    PW = [*range(0, 6), *range(18, 25)]; VW = [*range(6, 12), *range(25, 31)]
    dA, dB, itB = [], [], []
    for t in range(steps):
        a = rng.uniform(-1, 1, (N, 6))
        s = B.get_state(); A.set_state(s); g.set_state(torch.tensor(s, dtype=torch.float32)); led.before(B)
        A.step(a); _, rB, dnB = B.step(a)
        _, rg, dg = g.step(torch.tensor(a, dtype=torch.float32))
        so = B.get_state(); sg = g.state().cpu().numpy().astype(np.float64)
        same = (dnB == dg.cpu().numpy()) & (so[:, 44] == sg[:, 44]) & (rB == rg.cpu().numpy())
        live = same & (dnB == 0)
        e_all = np.maximum(np.abs(so[:, PW] - sg[:, PW]).max(1), DT * np.abs(so[:, VW] - sg[:, VW]).max(1))
        led.after(B, a, np.where(live, e_all, 0.0))
        gi = g.debug().cpu().numpy()[:, 13].astype(int)
        dA.append(np.where(live, gi - A.pgs_iters(), 0)); dB.append(np.where(live, gi - B.pgs_iters(), 0)); itB.append(B.pgs_iters().copy())
    dA, dB, itB = map(np.concatenate, (dA, dB, itB))
    led.finish("fly defaults (exit test live, cadence 16) N=%d" % N, p50=2e-6, p99=2e-5, exempt_share=0.02, check_force=False)
    print("   early exit (same-cadence oracle < 50 iterations) in %.1f %% of the env-steps; product - oracle(same cadence) != 0 in %.3f %%; product - oracle(Bullet's cadence) min / max %d / %d" % (
        100 * (itB < 50).mean(), 100 * (dB != 0).mean(), dA.min(), dA.max()))
    assert (itB < 50).mean() > 0.3                         # the exit test fires in a large share of the env-steps (free flight, arm at rest)
    assert (dB != 0).mean() < 1e-2 and dA.min() >= -1 and dA.max() <= 46      # never more than the gap between two tests (4 -> 20 -> 36 -> 50)
