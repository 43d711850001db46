"""Env registry: the two ids of the reference (peg_in_hole_gym/__init__.py:3-11).  Registers with gymnasium / gym when
either is importable; `make()` works without them."""
REGISTRY = {
    'peg-in-hole-v0': 'peg_in_hole_gym_amd.envs.base_env:BaseEnv',
    'peg-in-hole-mp-v0': 'peg_in_hole_gym_amd.envs.base_env_mp:BaseEnvMp',
}


def _load(entry_point):
    import importlib
    mod, cls = entry_point.split(":")
    return getattr(importlib.import_module(mod), cls)


def make(id, **kwargs):
    """gym.make-compatible constructor: make('peg-in-hole-mp-v0', client=..., task=..., mp_num=..., sub_num=..., ...)"""
    if id not in REGISTRY:
        raise KeyError("No registered env with id: %s" % id)
    return _load(REGISTRY[id])(**kwargs)


def register_with_gym():
    done = []
    for modname in ("gymnasium", "gym"):
        try:  # pragma: no cover - neither package is present in the build image
            import importlib
            reg = importlib.import_module(modname + ".envs.registration")
            for k, v in REGISTRY.items():
                try:
                    reg.register(id=k, entry_point=v)
                except Exception:  # noqa: BLE001  (already registered)
                    pass
            done.append(modname)
        except Exception:  # noqa: BLE001
            continue
    return done
