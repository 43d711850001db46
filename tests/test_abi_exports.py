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


def test_config_struct_layout_matches_header(tmp_path):
    """Every field of struct pih_config: offset and size as the C compiler lays out include/pih.h == the ctypes mirror in _lib.py
    (a probe compiled with gcc prints offsetof / sizeof for each field name found in the ctypes structure)."""
    import subprocess
    from peg_in_hole_gym_amd import _lib
    fields = [f[0] for f in _lib.PihConfig._fields_]
    src = tmp_path / "probe.c"
    body = "".join('  printf("%s %%zu %%zu\\n", offsetof(pih_config, %s), sizeof(((pih_config*)0)->%s));\n' % (f, f, f) for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pih.h"\nint main(void) {\n%s  printf("total %%zu %%d\\n", sizeof(pih_config), PIH_ABI_VERSION);\n  return 0;\n}\n' % body)
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    lines = subprocess.check_output([str(exe)], text=True).split("\n")
    seen = {}
    for ln in lines:
        if ln.strip():
            k, a, b = ln.split(); seen[k] = (int(a), int(b))
    for f in fields:
        d = getattr(_lib.PihConfig, f)
        assert seen[f] == (d.offset, d.size), (f, seen[f], d.offset, d.size)
    assert seen["total"] == (ctypes.sizeof(_lib.PihConfig), _lib.ABI_VERSION)
    assert ctypes.sizeof(_lib.PihConfig) == 16 * 4 + 8 + 12 * 4     # 16 int32, u64 seed, 12 floats
    c = _lib.default_config()
    assert c.solver_iters == 50 and c.ik_iters == 20 and abs(c.dt - 1 / 240) < 1e-9 and c.max_episode_steps == 2227 and c.enable_arm_collision == 3 and c.task_id == 0
    assert c.exit_check_stride == 16 and abs(c.residual_threshold - 1e-7) < 1e-12 and abs(c.warmstart - 0.85) < 1e-6


def test_integration_md_stub_matches_the_abi():
    """The ctypes stub printed in INTEGRATION.md section 1 (executed verbatim on the GPU by tests/test_gpu_api.py) must describe the
    CURRENT ABI: same struct fields in the same order as _lib.PihConfig, the current version number, pih_reset with its seed argument."""
    import re
    from peg_in_hole_gym_amd import _lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# peg_in_hole_gym/envs/_pih\.py.*?)```", md, re.S).group(1)
    fields = re.findall(r'\("([a-z_0-9]+)", C\.', code)
    assert fields == [f[0] for f in _lib.PihConfig._fields_]
    assert "pih_abi_version() == %d" % _lib.ABI_VERSION in code
    assert re.search(r"pih_reset\(h, None, int\(hard_reset\), C\.c_uint64\(0\), stream\)", code)


def test_object_list_of_random_fly_comes_from_the_generated_header():
    """RandomFly.OBJECTS (host), pih_object_name (library) and piho_fly_object_name (oracle) all list PIH_FLY_OBJ_NAMES of include/pih_model.h"""
    from peg_in_hole_gym_amd import _lib
    from peg_in_hole_gym_amd.envs.peg_in_hole import RandomFly
    from oracle import oracle as O
    assert list(RandomFly.OBJECTS) == _lib.object_names(_lib.TASK_RANDOM_FLY) == O.fly_object_names() == ["Banana", "Amicelli"]
    assert _lib.object_names(_lib.TASK_PEG_IN_HOLE) == []
    assert RandomFly.cfg_from_args(["Amicelli", 1 / 120.]) == {"object_id": 1, "dt": 1 / 120.}
    import pytest
    with pytest.raises(ValueError):
        RandomFly.cfg_from_args(["Apple", 1 / 120.])


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
