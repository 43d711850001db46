"""The north_star's acceptance run (tests/parity_util.first_exceedance_run) from the command line -- test tooling.  Writes
gpurun_out/r04_first_exceedance_N<N>.json when no output file is given and the product is the GPU (tools/merge_first_exceedance.py -> profiles/).
usage: python tools/first_exceedance.py N [steps] [gpu|emul] [out.json]
`emul` = tests/emul's fp32 HOST build of the product algorithm (CPU rehearsal); `gpu` = the HIP library through the C ABI."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O            # noqa: E402  (test infrastructure)
from tests import parity_util as P        # noqa: E402


def main():
    N = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    kind = sys.argv[3] if len(sys.argv) > 3 else "gpu"
    out = sys.argv[4] if len(sys.argv) > 4 else (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_first_exceedance_N%d.json" % N) if kind == "gpu" else None)
    if kind == "gpu":
        g = P.GpuProduct(N, seed=5)
    else:
        from tests.emul import emul as E
        E.build(); g = E.Emul(N, "f32", seed=5)
    r = P.first_exceedance_run(O, g, N, steps, progress=lambda s: print(s, flush=True))     # all four yardsticks
    first = r.pop("first")
    r["product_kind"] = kind
    print(json.dumps(r, indent=1))
    if out:
        r["first_step_per_env"] = {k: {m: v[m].tolist() for m in v} for k, v in first.items()} if N <= 4096 else None
        json.dump(r, open(out, "w"))


if __name__ == "__main__":
    main()
