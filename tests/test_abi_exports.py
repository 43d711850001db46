"""The C-ABI library loads and exports every symbol include/pih.h declares (no compute: no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    import __graft_entry__ as ge
    ge.build()
    hdr = open(os.path.join(ROOT, "include", "pih.h")).read()
    names = sorted(set(re.findall(r"\b(pih_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 12
    L = ctypes.CDLL(os.path.join(ROOT, "peg_in_hole_gym_amd", "csrc", "libpih_hip.so"))
    for n in names:
        assert hasattr(L, n), n
    from peg_in_hole_gym_amd import _lib
    assert sorted(_lib.EXPORTS) == names


def test_config_struct_layout_matches_header():
    from peg_in_hole_gym_amd import _lib
    assert ctypes.sizeof(_lib.PihConfig) == 14 * 4 + 8 + 12 * 4     # 14 int32 (task_id, solver_path, reserved), u64 seed, 12 floats
    assert _lib.PihConfig.task_id.offset == 44 and _lib.PihConfig.solver_path.offset == 48
    assert _lib.PihConfig.seed.offset == 56 and _lib.PihConfig.dt.offset == 64
    c = _lib.default_config()
    assert c.solver_iters == 50 and c.ik_iters == 20 and abs(c.dt - 1 / 240) < 1e-9 and c.max_episode_steps == 2227 and c.enable_arm_collision == 3 and c.task_id == 0


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    from peg_in_hole_gym_amd import _lib
    L = _lib.load()
    c = _lib.default_config(n_envs=4)
    h = ctypes.c_void_p()
    assert L.pih_create(ctypes.byref(c), None, ctypes.byref(h)) != 0
    assert b"no HIP device" in L.pih_last_error(None)
    import pytest
    from peg_in_hole_gym_amd.vec_env import PihVecEnv
    with pytest.raises(_lib.PihError):
        PihVecEnv(4)


def test_missing_extension_fails_loudly():
    """No CPU fallback: with the HIP library absent the product raises PihError naming the build command (fresh interpreter,
    PIH_LIB_PATH pointing nowhere)."""
    import subprocess, sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['PIH_LIB_PATH'] = '/nonexistent/libpih_hip.so'\n"
            "from peg_in_hole_gym_amd import _lib\n"
            "try:\n    _lib.load(); print('LOADED')\n"
            "except _lib.PihError as e:\n    print('PihError:', e)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "PihError:" in out.stdout and "no CPU fallback" in out.stdout and "LOADED" not in out.stdout, out.stdout + out.stderr
