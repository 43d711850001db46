"""SURVEY section 5 (race / memory-error detection): the CPU restatement (oracle/, C) and the host build of the PRODUCT's device
algorithm (tests/emul: the same pih_common.h / pih_step.h / pih_fly.h the HIP kernels compile, with the wave primitives emulated)
under AddressSanitizer + UndefinedBehaviorSanitizer, rolled through action / scripted-coil (> 32 contacts) / random-fly episodes.
GPU sanitizers are not available on this pool; this is the memory-safety net for the indexing both builds share."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    for cc in ("gcc", "g++"):
        try:
            p = subprocess.run([cc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
        except OSError:
            continue
        if p and os.path.isabs(p) and os.path.exists(p):
            return os.path.realpath(p)
    return None


def test_oracle_and_host_build_are_clean_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("libasan not installed with this gcc")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emul"), "-s", "asan"])
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize_roll.py"), "200"], capture_output=True, text=True, timeout=1500, env=env)
    tail = out.stdout[-1500:] + out.stderr[-4000:]
    assert out.returncode == 0 and "SANITIZE-ROLL-OK" in out.stdout, tail
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, tail
