"""ctypes binding of the C ABI in include/pih.h (libpih_hip.so).  There is no CPU fallback: if the HIP library is
missing or no GPU is present this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIH_LIB_PATH") or os.path.join(_HERE, "csrc", "libpih_hip.so")   # override: A/B of builds on one box

STATE_WORDS = 256
DEBUG_WORDS = 1024
ACTION_DIM = 4
OBS_DIM = 5
FIELD_STATE, FIELD_TIP_POSE, FIELD_CONTACT_FORCE, FIELD_DEBUG, FIELD_EE_POS = 0, 1, 2, 3, 4
# state record word offsets (include/pih.h)
S_QARM, S_QDARM, S_POS, S_QUAT, S_VLIN, S_VANG, S_QJ, S_QDJ, S_TARGET = 0, 9, 18, 21, 25, 28, 31, 54, 77
S_FSM, S_FSMT, S_DONE, S_GRASP, S_RANDY, S_RNG_HI, S_RNG, S_STEPS, S_OFFSET = 86, 87, 88, 89, 90, 91, 92, 93, 94
S_SPARE, S_TIP, S_CFORCE, S_NCONTACT, S_PGS_ITERS, S_INVALID, S_CACHE_N = 97, 98, 105, 106, 107, 112, 128
TASK_PEG_IN_HOLE, TASK_RANDOM_FLY = 0, 1
ABI_VERSION = 4

EXPORTS = ["pih_default_config", "pih_abi_version", "pih_task_dims", "pih_object_name", "pih_create", "pih_destroy", "pih_reset", "pih_reseed", "pih_step", "pih_step_n",
           "pih_get_state", "pih_set_state", "pih_ik", "pih_ik_ur5", "pih_render", "pih_render_ex", "pih_grasp_labels", "pih_timing", "pih_timing2", "pih_set_timing", "pih_last_error"]


class PihConfig(C.Structure):
    """struct pih_config (include/pih.h)"""
    _fields_ = [("n_envs", C.c_int32), ("env_index0", C.c_int32), ("mode", C.c_int32), ("solver_iters", C.c_int32),
                ("ik_iters", C.c_int32), ("max_episode_steps", C.c_int32), ("auto_reset", C.c_int32),
                ("enable_self_collision", C.c_int32), ("debug", C.c_int32), ("schedule", C.c_int32), ("enable_arm_collision", C.c_int32), ("task_id", C.c_int32), ("solver_path", C.c_int32), ("attach_ball", C.c_int32), ("exit_check_stride", C.c_int32), ("object_id", C.c_int32), ("seed", C.c_uint64),
                ("dt", C.c_float), ("residual_threshold", C.c_float), ("erp", C.c_float), ("warmstart", C.c_float),
                ("contact_margin", C.c_float), ("linear_slop", C.c_float), ("ik_damping", C.c_float), ("ik_residual", C.c_float),
                ("dv", C.c_float), ("reserved_f", C.c_float * 3)]


class PihError(RuntimeError):
    pass


_lib = None


def load():
    """Load libpih_hip.so; raises PihError if it has not been built (python peg_in_hole_gym_amd/csrc/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PihError("HIP extension %s is missing: build it with `python peg_in_hole_gym_amd/csrc/build.py` "
                       "(there is no CPU fallback)" % LIB_PATH)
    # torch must be in the process BEFORE libpih_hip.so: both link libamdhip64, and the process has to end up with ONE HIP
    # runtime (the one torch ships) -- loaded the other way round, pih_create sees no device
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.pih_default_config.argtypes = [C.POINTER(PihConfig)]
    L.pih_default_config.restype = None
    L.pih_abi_version.restype = C.c_int
    L.pih_task_dims.argtypes = [C.c_int, C.POINTER(C.c_int32 * 3)]
    L.pih_object_name.argtypes = [C.c_int, C.c_int]
    L.pih_object_name.restype = C.c_char_p
    L.pih_create.argtypes = [C.POINTER(PihConfig), vp, C.POINTER(vp)]
    L.pih_destroy.argtypes = [vp]
    L.pih_reset.argtypes = [vp, vp, C.c_int, C.c_uint64, vp]
    L.pih_step.argtypes = [vp, vp, vp, vp, vp, vp]
    L.pih_step_n.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.pih_get_state.argtypes = [vp, C.c_int, vp, vp]
    L.pih_set_state.argtypes = [vp, C.c_int, vp, vp]
    L.pih_ik.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.pih_ik_ur5.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.pih_render.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.pih_render_ex.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.pih_grasp_labels.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.pih_reseed.argtypes = [vp, C.c_uint64]
    L.pih_timing.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.pih_timing2.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.pih_set_timing.argtypes = [vp, C.c_int]
    L.pih_last_error.argtypes = [vp]
    L.pih_last_error.restype = C.c_char_p
    if L.pih_abi_version() != ABI_VERSION:
        raise PihError("%s has ABI version %d, this package expects %d: rebuild it" % (LIB_PATH, L.pih_abi_version(), ABI_VERSION))
    _lib = L
    return L


FLY_STATE_WORDS, FLY_ACTION_DIM, FLY_OBS_DIM = 48, 6, 6
# random-fly record word offsets (include/pih.h PIH_F_*)
F_Q, F_QD, F_TARGET, F_OPOS, F_OQUAT, F_OVLIN, F_OVANG, F_DONE, F_STEPS, F_RNG, F_RNG_HI, F_OFFSET, F_SPARE, F_INVALID, F_EE, F_CFORCE, F_NCONTACT = \
    0, 6, 12, 18, 21, 25, 28, 31, 32, 33, 34, 35, 38, 39, 40, 43, 44


def task_dims(task_id):
    """(action dim, obs dim, state words per env) of a task, from the library"""
    out = (C.c_int32 * 3)()
    if load().pih_task_dims(int(task_id), C.byref(out)) != 0:
        raise PihError("unknown task_id %r" % (task_id,))
    return int(out[0]), int(out[1]), int(out[2])


def object_names(task_id):
    """names of the objects compiled into the library for a task (random-fly: generated from the reference's asset files), index = object_id"""
    L = load(); out = []
    while True:
        n = L.pih_object_name(int(task_id), len(out))
        if n is None:
            return out
        out.append(n.decode())


def default_config(**kw):
    c = PihConfig()
    load().pih_default_config(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError("pih_config has no field %r" % k)
        setattr(c, k, v)
    return c
