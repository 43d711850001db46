"""Build libpih_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by developers."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "pih_hip.hip")
OUT = os.path.join(HERE, "libpih_hip.so")
DEPS = [SRC, os.path.join(HERE, "pih_device.h"), os.path.join(HERE, "pih_common.h"), os.path.join(HERE, "pih_wave.h"), os.path.join(HERE, "pih_step.h"), os.path.join(HERE, "pih_fly.h"), os.path.join(HERE, "pih_render.h"), os.path.join(HERE, "pih_math.h"),
        os.path.join(HERE, "..", "..", "include", "pih.h"), os.path.join(HERE, "..", "..", "include", "pih_model.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-ffast-math", "-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", OUT)
