"""Task descriptors.  In the reference a task is a MetaEnv subclass whose methods call PyBullet once per agent
(envs/meta_env.py:8-42) and is registered by name in TASK_LIST (envs/base_env.py:9-11).  Here the per-step methods of a
task are COMPILED INTO THE HIP LIBRARY and selected by `task_id` in pih_config (include/pih.h PIH_TASK_*): a descriptor carries
what the host needs -- the spaces (the wire format of actions / observations), the kernel task id, the model tables the kernel
was generated from, and how the reference's constructor `args` map to library settings -- and its MetaEnv methods drive the
library for ONE agent, so code written against the reference's task objects keeps working."""
import numpy as np

from .utils import Box


def _objects_from_model_header():
    """The object list of the 'random-fly' task = PIH_FLY_OBJ_NAMES of the GENERATED include/pih_model.h (tools/gen_model_header.py reads
    every single-link free body under the reference's envs/assets/urdf); index = pih_config.object_id.  The library exports the same list
    (pih_object_name), tests/test_abi_exports.py checks that the two agree."""
    import os
    import re
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "include", "pih_model.h")
    try:
        m = re.search(r"#define PIH_FLY_OBJ_NAMES \{([^}]*)\}", open(hdr).read())
        names = tuple(re.findall(r'"([^"]+)"', m.group(1)))
    except (OSError, AttributeError) as ex:
        raise ImportError("peg_in_hole_gym_amd: cannot read the object table PIH_FLY_OBJ_NAMES from %s (%r); regenerate it with "
                          "tools/gen_model_header.py" % (hdr, ex)) from ex
    if not names:
        raise ImportError("peg_in_hole_gym_amd: PIH_FLY_OBJ_NAMES in %s is empty" % hdr)
    return names


class MetaEnv(object):
    """Task plugin contract of the reference (envs/meta_env.py:8-42): class attrs action_space / observation_space;
    __init__(client, offset, args); apply_action, get_info -> (obs, reward, done, info), reset(hard_reset), render(mode).
    `client` (a BulletClient in the reference) is accepted and ignored: the physics server is the HIP library."""
    action_space = Box(np.array([-1]), np.array([1]))
    observation_space = Box(np.array([-1]), np.array([1]))
    task_id = -1                 # PIH_TASK_* compiled into libpih_hip.so
    model_tables = ()            # macro prefixes of include/pih_model.h this task's kernel instantiates
    default_cfg = {}             # library settings of the task (pih_config fields)

    @classmethod
    def cfg_from_args(cls, args):
        """Map the reference constructor's `args` list to pih_config fields."""
        return {}

    def __init__(self, client=None, offset=(0, 0, 0), args=None, backend_factory=None, **cfg):
        self.p = client
        self.offset = np.array(offset, dtype=float)
        self.args = args
        kw = dict(self.default_cfg); kw.update(self.cfg_from_args(args)); kw.update(cfg)
        if backend_factory is None:
            from ..vec_env import PihVecEnv
            self._backend = PihVecEnv(1, offsets=self.offset.reshape(1, 3), task_id=self.task_id, **kw)
        else:
            self._backend = backend_factory(1, self.offset.reshape(1, 3), task_id=self.task_id, **kw)
        self._action = np.zeros(self.action_space.shape, dtype=np.float32)
        self.done = False
        import inspect      # decided once from the signature (a TypeError raised INSIDE a backend's reset must propagate)
        params = inspect.signature(self._backend.reset).parameters
        self._hard_ok = "hard_reset" in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())

    def _load_models(self):
        """The reference loads URDFs here; the model tables are baked into the kernel (include/pih_model.h)."""

    def _reset_internals(self):
        self._backend.reset(None)

    def apply_action(self, action):
        """Stored; executed together with the physics step by get_info (the reference's BaseEnv.step calls apply_action,
        stepSimulation, get_info in this order, envs/base_env.py:61-71)."""
        self._action = np.asarray(action, dtype=np.float32).reshape(self.action_space.shape)

    def get_info(self):
        a = self._action.reshape(1, -1)
        try:
            import torch
            if hasattr(self._backend, "device"):
                a = torch.as_tensor(a, device=self._backend.device)
        except ImportError:  # pragma: no cover
            pass
        obs, rew, done = self._backend.step(a)
        to_np = lambda x: x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        self.done = bool(to_np(done)[0])
        return to_np(obs)[0].astype(np.float32), float(to_np(rew)[0]), self.done, {}

    def reset(self, hard_reset=False):
        if hard_reset:
            self._load_models()
        if self._hard_ok:
            self._backend.reset(None, hard_reset=bool(hard_reset))
        else:
            self._backend.reset(None)
        self.done = False

    def render(self, mode="rgb_array"):
        raise NotImplementedError("this task has no camera")

    def close(self):
        if hasattr(self._backend, "close"):
            self._backend.close()


class PegInHole(MetaEnv):
    """envs/peg_in_hole.py:11-304: Panda + 25-link pipe + hole tube + table."""
    action_space = Box(np.array([-1] * 4), np.array([1] * 4))       # ee xyz target + finger (envs/peg_in_hole.py:12)
    observation_space = Box(np.array([-1] * 5), np.array([1] * 5))  # finger1 finger2 ee xyz (envs/peg_in_hole.py:13)
    task_id = 0
    model_tables = ("PIH_LINK_", "PIH_ARM_", "PIH_EE_", "PIH_FINGER_", "PIH_PIPE_", "PIH_TABLE_", "PIH_HOLE_")
    pandaEndEffectorIndex = 11
    pandaNumDofs = 7

    def render(self, mode="rgb_array"):
        """PegInHole.render (envs/peg_in_hole.py:276-304): [300,300,4] = depth, r, g, b from the wrist camera"""
        img = self._backend.render(300, 300, shaded=True)
        return (img.detach().cpu().numpy() if hasattr(img, "detach") else np.asarray(img))[0].astype(np.float64)


class RandomFly(MetaEnv):
    """'random-fly' (README.md:38: task='random-fly', args=['Banana', 1/120.]): the UR5 of assets/urdf/ur5.urdf driven by
    ur_execute (envs/utils.py:70-82) next to one free-flying object spawned by random_pos_in_panda_space (envs/utils.py:97-107).
    The task class is not in the reference snapshot; rest pose, launch law, reward / done and observation are build-defined
    (DESIGN.md section 6.4).  args[0] = object name -- one of OBJECTS, the single-link free bodies under envs/assets/urdf that
    tools/gen_model_header.py turned into tables of include/pih_model.h (PIH_FLY_OBJ_NAMES: 'Banana', 'Amicelli'); its index is the
    library's pih_config.object_id -- args[1] = physics time step."""
    action_space = Box(np.array([-1] * 6), np.array([1] * 6))       # ee target xyz + euler rpy (envs/utils.py:71-72)
    observation_space = Box(np.array([-1] * 6), np.array([1] * 6))  # ee xyz + object xyz
    task_id = 1
    model_tables = ("PIH_UR5_", "PIH_FLY_OBJ_", "PIH_TABLE_")
    default_cfg = {"max_episode_steps": 480, "contact_margin": 0.02}   # margin = Bullet's contact breaking threshold (the object moves cm per step)
    urEndEffectorIndex = 7
    urNumDofs = 6
    OBJECTS = _objects_from_model_header()

    @classmethod
    def cfg_from_args(cls, args):
        if not args:
            return {}
        if str(args[0]) not in cls.OBJECTS:
            raise ValueError("random-fly: object %r is not compiled into the library (available: %s)" % (args[0], ", ".join(cls.OBJECTS)))
        kw = {"object_id": cls.OBJECTS.index(str(args[0]))}
        if len(args) > 1:
            kw["dt"] = float(args[1])
        return kw
