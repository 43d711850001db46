#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the sources that determine libpih_hip.so: csrc/*.h, csrc/*.hip, include/*.h and the build flags in
csrc/build.py.  Profile summaries under profiles/ carry the hash of the sources they were measured on; bench.py compares it with the
current one and marks quoted counter figures `stale` when a kernel source changed since the profile was taken."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "peg_in_hole_gym_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "peg_in_hole_gym_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(ROOT, "peg_in_hole_gym_amd", "csrc", "build.py")])
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha16())
