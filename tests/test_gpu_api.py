"""C-ABI surface on the GPU: masks, state round trip, step_n == n x step, error codes, odd sizes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _gpu(n, **kw):
    from peg_in_hole_gym_amd.vec_env import PihVecEnv
    return PihVecEnv(n, **kw)


def test_masked_reset_and_state_roundtrip(torch_mod):
    torch = torch_mod
    n = 37                                        # deliberately not a multiple of anything
    g = _gpu(n, seed=9)
    s0 = g.state().clone()
    a = torch.rand(n, 4, device="cuda") * 2 - 1
    for _ in range(20):
        g.step(a)
    s1 = g.state().clone()
    assert not torch.equal(s0[:, :86], s1[:, :86])
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda"); mask[::3] = 1
    g.reset(mask)
    s2 = g.state()
    keep = mask == 0
    assert torch.equal(s2[keep], s1[keep])                                   # untouched envs keep their state bit for bit
    assert (s2[mask == 1][:, 93] == 0).all() and (s2[mask == 1][:, 20] == 0.11).all()   # reset envs: step counter 0, spawn height
    assert (s2[mask == 1][:, 92] > s1[mask == 1][:, 92]).all()               # RNG counter advanced: a NEW random scene
    g.set_state(s1)
    assert torch.equal(g.state(), s1)


def test_hard_reset_reseed_and_two_handles(torch_mod):
    """Like the reference (resetSimulation + fresh random draws, envs/base_env.py:84-94, envs/peg_in_hole.py:239-267), a hard reset gives a
    NEW scene; replay is explicit (pih_reset(seed != 0) / pih_reseed rewind the draw sequence).  Two live handles in one process do not
    disturb each other (every entry point selects its handle's device and restores the caller's)."""
    torch = torch_mod
    n = 33
    a = _gpu(n, seed=4); b = _gpu(n, seed=4)
    first = a.state().clone()
    act = torch.rand(n, 4, device="cuda") * 2 - 1
    for _ in range(5):
        a.step(act); b.step(act)                     # interleaved launches on two handles
    assert torch.equal(a.state(), b.state())
    a.reset()                                         # soft reset: a NEW scene (counter continues)
    assert not torch.equal(a.state()[:, 18:20], first[:, 18:20])
    second = a.state().clone()
    a.reset(hard_reset=True)                          # hard reset: again a NEW scene (an unmodified training loop sees fresh scenes)
    third = a.state().clone()
    assert not torch.equal(third[:, 18:20], first[:, 18:20]) and not torch.equal(third[:, 18:20], second[:, 18:20])
    assert (third[:, 92] > second[:, 92]).all()       # the draw counter kept advancing
    a.reset(hard_reset=True, seed=4)                  # explicit replay: the seed's first scene again, bit for bit
    assert torch.equal(a.state()[:, :98], first[:, :98])
    a.reset()
    assert torch.equal(a.state()[:, :98], second[:, :98])      # ... and the same sequence after it
    a.reseed(77); a.reset(hard_reset=True)            # pih_reseed: new base seed, the next reset starts its sequence
    c = _gpu(n, seed=77)
    assert torch.equal(a.state()[:, :98], c.state()[:, :98]) and not torch.equal(a.state()[:, 18:20], first[:, 18:20])
    assert torch.equal(b.state()[:, 93], torch.full((n,), 5.0, device="cuda"))     # b was never touched by a's resets


def test_non_finite_env_is_frozen_and_flagged(torch_mod):
    """auto_reset = 0 (the facade's setting): an env whose state turns non-finite is re-initialised, reported done, flagged
    invalid and then stays frozen -- it does not silently start a second episode."""
    torch = torch_mod
    n = 8
    g = _gpu(n, seed=1)
    s = g.state().clone(); s[2, 18] = float("nan"); s[5, 31] = float("nan"); g.set_state(s)
    a = torch.zeros(n, 4, device="cuda")
    _, _, done = g.step(a)
    st = g.state()
    assert done.cpu().tolist() == [0, 0, 1, 0, 0, 1, 0, 0]
    assert torch.isfinite(st).all() and g.invalid().cpu().tolist() == [False, False, True, False, False, True, False, False]
    assert (st[[2, 5], 97] == 1).all() and (st[[2, 5], 88] == 1).all()
    frozen = st[[2, 5]].clone()
    for _ in range(3):
        _, _, done = g.step(a)
    assert torch.equal(g.state()[[2, 5], :98], frozen[:, :98]) and done[[2, 5]].all()
    g.reset()
    assert not g.invalid().any()


def test_step_n_equals_repeated_step(torch_mod):
    torch = torch_mod
    n = 64
    a = torch.rand(n, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 2 - 1
    g1 = _gpu(n, seed=2); g2 = _gpu(n, seed=2)
    for _ in range(17):
        g1.step(a)
    g2.step_n(17, a)
    assert torch.equal(g1.state(), g2.state())
    assert torch.equal(g1.obs, g2.obs) and torch.equal(g1.done, g2.done)


def test_error_paths(torch_mod):
    from peg_in_hole_gym_amd import _lib
    L = _lib.load()
    g = _gpu(4)
    assert L.pih_step(g.h, None, None, None, None, None) != 0               # action mode needs actions
    assert b"actions_dev is NULL" in L.pih_last_error(g.h)
    out = torch_mod.empty(4, 1024, device="cuda")
    assert L.pih_get_state(g.h, _lib.FIELD_DEBUG, out.data_ptr(), None) != 0   # debug buffer not enabled
    assert L.pih_get_state(g.h, 99, out.data_ptr(), None) != 0
    assert L.pih_set_state(g.h, _lib.FIELD_TIP_POSE, out.data_ptr(), None) != 0
    c = _lib.default_config(n_envs=0); h = C.c_void_p()
    assert L.pih_create(C.byref(c), None, C.byref(h)) != 0


def test_timing_api_and_obs_fields(torch_mod):
    torch = torch_mod
    g = _gpu(128)
    a = torch.zeros(128, 4, device="cuda")
    g.set_timing(True)
    for _ in range(5):
        obs, rew, done = g.step(a)
    ms, k = g.timing(reset=False)
    pre, phys, k2 = g.timing2()
    assert k == 5 and k2 == 5 and 0 < ms < 50 and pre > 0 and phys > pre and abs(pre + phys - ms) < 1e-9
    for _ in range(1100):                        # more timed launches than the event pool holds: folded into running sums
        g.step(a)
    assert g.timing()[1] == 1100
    torch.cuda.synchronize()
    st = g.state()
    assert torch.allclose(obs[:, 2:5], g.ee_position()) and torch.allclose(obs[:, 0:2], st[:, 7:9])
    assert torch.equal(g.tip_pose(), st[:, 98:105]) and torch.equal(g.contact_force(), st[:, 105])
    assert ((rew == 0) | (rew == 1)).all() and ((done == 0) | (done == 1)).all()


def test_graft_entry_build_then_smoke_in_a_fresh_process():
    """The driver calls build() and smoke() in one fresh interpreter: the HIP library must not be loaded before torch
    (two HIP runtimes in one process -> pih_create sees no device)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "smoke ok" in out.stdout


def test_integration_md_ctypes_stub_runs_as_written(torch_mod):
    """INTEGRATION.md section 1 is the binding a maintainer of the reference would add: execute that code block verbatim
    (only the library name is made absolute) with the free variables a caller would own, and check that it did step."""
    import os, re
    torch = torch_mod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# peg_in_hole_gym/envs/_pih\.py.*?)```", md, re.S).group(1)
    code = code.replace('"libpih_hip.so"', repr(os.path.join(root, "peg_in_hole_gym_amd", "csrc", "libpih_hip.so")))
    n = 6
    ns = dict(mp_num=2, sub_num=3, offsets_host_ptr=None, hard_reset=False,
              actions=torch.zeros(n, 4, device="cuda"), obs=torch.full((n, 5), -7.0, device="cuda"),
              reward=torch.zeros(n, device="cuda"), done=torch.zeros(n, dtype=torch.uint8, device="cuda"),
              img=torch.zeros(n, 300, 300, 4, device="cuda"))
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    torch.cuda.synchronize()
    assert ns["cfg"].n_envs == n and ns["cfg"].solver_iters == 50
    assert (ns["obs"] != -7.0).all() and torch.isfinite(ns["obs"]).all()          # pih_step wrote the observations
    assert 0.0 < float(ns["img"][..., 0].min()) <= float(ns["img"][..., 0].max()) <= 1.0   # pih_render wrote a depth buffer (near..far)
    assert ns["L"].pih_destroy(ns["h"]) == 0


def test_results_do_not_depend_on_batch_size_or_dispatch_order(torch_mod):
    """Envs never interact and env seeds follow the GLOBAL index, so env i must evolve bit-identically in a batch of 37, of 130
    (three controller groups, ragged last group) and with the longest-job-first dispatch switched off."""
    torch = torch_mod
    gen = torch.Generator(device="cuda").manual_seed(11)
    acts = torch.rand(60, 130, 4, device="cuda", generator=gen) * 2 - 1
    a = _gpu(130, seed=3, auto_reset=1, max_episode_steps=40)
    b = _gpu(37, seed=3, auto_reset=1, max_episode_steps=40)
    c = _gpu(130, seed=3, auto_reset=1, max_episode_steps=40, schedule=0)
    for t in range(60):
        oa, ra, da = a.step(acts[t]); ob, rb, db = b.step(acts[t, :37].contiguous()); oc, rc, dc = c.step(acts[t])
        assert torch.equal(oa[:37], ob) and torch.equal(da[:37], db) and torch.equal(oa, oc)
    assert torch.equal(a.state()[:37], b.state()) and torch.equal(a.state(), c.state())
    assert int(a.state()[:, 93].max()) <= 40                                   # episodes were cut and auto-reset at 40 steps


def test_scripted_facade_on_gpu(torch_mod):
    """BaseEnvMp in scripted mode on the real backend: one step() = one whole grasp-and-insert episode, observation = the
    wrist-camera image at state-2 entry, info = label images + [x, y, angle, width, length] (envs/peg_in_hole.py:33-37,116)."""
    from peg_in_hole_gym_amd.envs import BaseEnvMp
    env = BaseEnvMp(client=None, task="peg-in-hole", mp_num=2, sub_num=2, offset=[2., 3., 0.], mode="scripted", seed=5)
    env.reset()
    obs, rew, done, info = env.step(env.action_space.sample())
    assert done == [[True, True], [True, True]]
    for i in range(2):
        for j in range(2):
            img = obs[i][j]
            assert img.shape == (300, 300, 4) and 0.0 < img[..., 0].min() and img[..., 0].max() <= 1.0
            assert ((img[..., 1] > 153.0 * 0.95 + 1e-3) & (img[..., 1] < 255.0)).any()   # the (shaded) pipe is in view from above the grasp point
            (pos, sn, cs, wid), meta = info[i][j]
            assert pos.shape == (300, 300) and set(np.unique(pos)) == {0.0, 50.0} and abs(meta[3] - 60.0) < 1e-3 and abs(meta[4] - 30.0) < 1e-3
            assert rew[i][j] in (0.0, 1.0)
    env.render()
    assert env.images[1][1].shape == (300, 300, 4)
    env.close()


def test_integration_md_drop_in_usage_on_gpu(torch_mod):
    """INTEGRATION.md section 3 (smaller batch) and the single-process path of section 4 on the real backend."""
    import peg_in_hole_gym_amd as peg_in_hole_gym
    env = peg_in_hole_gym.make('peg-in-hole-mp-v0', client=None, task='peg-in-hole', mp_num=4, sub_num=4,
                               offset=[2., 3., 0.], args=None, is_test=False)
    obs = env.reset()
    assert len(obs) == 4 and len(obs[0]) == 4 and obs[0][0].shape == (5,)
    for _ in range(3):
        obs, reward, done, info = env.step(env.action_space.sample())
    assert len(reward) == 4 and len(done[3]) == 4 and np.isfinite(np.asarray(obs)).all()
    o2, r2, d2 = env.step_tensor(torch_mod.zeros(16, 4, device="cuda"))
    assert o2.shape == (16, 5) and o2.is_cuda
    env.close()
    from peg_in_hole_gym_amd.distributed import ShardedVecEnv
    sv = ShardedVecEnv(total_envs=128, gather_obs=True)
    o, r, d = sv.step(torch_mod.zeros(128, 4, device="cuda"))
    assert o.shape == (128, 5) and sv.n_local == 128 and sv.obs_all is None          # world size 1: nothing to gather


def test_state_dict_round_trip_resumes_bit_for_bit(torch_mod, tmp_path):
    """Checkpoint / resume: state_dict() -> torch.save -> a NEW handle -> load_state_dict(): the resumed handle continues
    bit for bit, including the scenes later auto-resets draw (RNG counters are part of the record); a checkpoint of another config or
    seed is refused."""
    torch = torch_mod
    from peg_in_hole_gym_amd import _lib
    from peg_in_hole_gym_amd.vec_env import PihVecEnv
    n = 70
    kw = dict(seed=6, auto_reset=1, max_episode_steps=30)
    a = _gpu(n, **kw)
    gen = torch.Generator(device="cuda").manual_seed(3)
    acts = torch.rand(80, n, 4, device="cuda", generator=gen) * 2 - 1
    for t in range(40):
        a.step(acts[t])
    path = tmp_path / "ckpt.pt"
    torch.save(a.state_dict(), path)
    sd = torch.load(path, weights_only=True)
    b = _gpu(n, **kw); b.load_state_dict(sd)
    c = PihVecEnv.from_state_dict(sd)
    for t in range(40, 80):                       # crosses an auto-reset of every env (episode length 30)
        oa = [x.clone() for x in a.step(acts[t])]; ob = b.step(acts[t]); oc = c.step(acts[t])
        for x, y, z in zip(oa, ob, oc):
            assert torch.equal(x, y) and torch.equal(x, z)
    assert torch.equal(a.state(), b.state()) and torch.equal(a.state(), c.state())
    with pytest.raises(_lib.PihError):
        _gpu(n, seed=6, auto_reset=1, max_episode_steps=31).load_state_dict(sd)      # another config
    with pytest.raises(_lib.PihError):
        _gpu(n, seed=7, auto_reset=1, max_episode_steps=30).load_state_dict(sd)      # another base seed
    with pytest.raises(_lib.PihError):
        _gpu(n + 1, **kw).load_state_dict(sd, strict=False)                          # another batch size


def test_new_seed_needs_a_full_reset(torch_mod):
    """pih_reset(seed != 0) / a pending pih_reseed with a MASK would leave the unmasked envs on another seed's stream ->
    rejected (-2) and nothing changes; the unmasked form works as before."""
    torch = torch_mod
    from peg_in_hole_gym_amd import _lib
    n = 9
    g = _gpu(n, seed=4)
    s0 = g.state().clone()
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda"); mask[::2] = 1
    with pytest.raises(_lib.PihError, match="reset of all envs"):
        g.reset(mask, seed=5)
    assert torch.equal(g.state(), s0)
    g.reseed(5)
    with pytest.raises(_lib.PihError, match="reset of all envs"):
        g.reset(mask)
    g.reset()                                                  # consumes the pending reseed: seed 5's first scene
    assert torch.equal(g.state()[:, :98], _gpu(n, seed=5).state()[:, :98])
    before = g.state()[:, 92].clone()
    g.reset(mask)                                              # masked resets are fine again afterwards: the masked envs draw on, the others rest
    after = g.state()[:, 92]
    assert (after[mask == 1] > before[mask == 1]).all() and torch.equal(after[mask == 0], before[mask == 0])


def test_roctx_ranges_do_not_disturb_the_step(torch_mod):
    """config.debug != 0 puts roctx ranges around the step / reset launches (tracing): same results as without"""
    torch = torch_mod
    n = 16
    a = _gpu(n, seed=2, debug=1); b = _gpu(n, seed=2)
    act = torch.rand(n, 4, device="cuda") * 2 - 1
    for _ in range(5):
        a.step(act); b.step(act)
    a.reset(); b.reset()
    a.step(act); b.step(act)
    assert torch.equal(a.state()[:, :128], b.state()[:, :128])


@pytest.mark.parametrize("mode,n,steps", [(0, 2500, 150), (1, 2500, 150), (0, 16384, 40)])
def test_fused_launch_equals_the_two_launch_step(torch_mod, mode, n, steps):
    """Round 4: one launch per step (controller wavefronts + env wavefronts in one grid, mailbox + release / acquire flag, dispatch order
    built in-kernel) against the two-launch step of rounds 1-3 with the same one-env-per-lane controller (pih_config.schedule + 8): the
    same arithmetic, so the states are bit-identical -- action mode with auto-reset and scripted mode (where the env waves read the
    state-machine words from the mailbox before collision detection); odd batch size, several rounds of wavefronts; no controller time-out.
    16 384 envs: 16 640 workgroups per launch (the mailbox race of DESIGN section 6.0 needed hundreds of workgroups in flight to show)."""
    torch = torch_mod
    kw = dict(seed=3, auto_reset=1, max_episode_steps=90) if mode == 0 else dict(seed=3, mode=1, dv=0.05)
    a = _gpu(n, **kw); b = _gpu(n, schedule=1 + 8, **kw); c = _gpu(n, schedule=0, **kw)      # fused; two launches; fused without the dispatch order
    gen = torch.Generator(device="cuda").manual_seed(5)
    for t in range(steps):
        act = torch.rand(n, 4, device="cuda", generator=gen) * 2 - 1
        oa = [x.clone() for x in a.step(act)]; ob = [x.clone() for x in b.step(act)]; oc = c.step(act)
        for x, y, z in zip(oa, ob, oc):
            assert torch.equal(x, y) and torch.equal(x, z), "step %d" % t
    assert torch.equal(a.state(), b.state()) and torch.equal(a.state(), c.state())
    a.set_timing(1); a.step(act); a.timing2()                 # timing2 also reports a controller time-out of the fused launch (-5): none
    assert int(a.state()[:, 106].max().item()) > (10 if steps >= 100 else 3)
