#!/usr/bin/env python
"""One long reference chain (tests/golden/G19 .. G22) through the C-ABI under the CURRENT environment, one JSON line out.

  python tools/chain_run.py G21b_ddim250_256 f16x3 [batch]

The library reads its PRG_* switches once per process, so precision / dispatch experiments (tools/gpu_precision_budget.sh) run this
script once per variant.  Same code path as tests/test_gpu_f16x3.py::test_long_chain_f16x3_north_star (it calls its helper)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    name, dtype = sys.argv[1], sys.argv[2]
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    import numpy as np
    import conftest
    import test_gpu_parity as T

    cache = {}

    def golden(n):
        if n not in cache:
            cache[n] = dict(np.load(os.path.join(conftest.GOLDEN, n + ".npz")))
        return cache[n]

    from pointreggpt_amd import _lib, geometry
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import MaskUnet, Unet
    _lib.load()

    class NS:
        pass
    hip = NS()
    hip.G, hip.GaussianDiffusion, hip.MaskUnet, hip.Unet, hip.lib = geometry, GaussianDiffusion, MaskUnet, Unet, _lib
    t0 = time.time()
    g, rep, img = T._run_long_chain(hip, golden, name, dtype, batch=batch)
    rep["seconds"] = round(time.time() - t0, 2)
    rep["fixture"], rep["dtype"], rep["batch"] = name, dtype, batch
    rep["env"] = {k: v for k, v in os.environ.items() if k.startswith("PRG_")}
    print("CHAIN " + json.dumps(rep))


if __name__ == "__main__":
    main()
