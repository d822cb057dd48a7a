#!/usr/bin/env python3
"""Does the x pass time of a grid depend on WHERE its buffers were allocated?  Several engines alive at once in one process.
  python3 tools/debug/mode_probe.py 512 5"""
import math
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from conftest import to4
    import pse_amd
    g, k = int(sys.argv[1]), int(sys.argv[2])
    h = 24.0 / 256
    L = g * h * 4
    n = 100_000
    rng = np.random.default_rng(1)
    pos = (rng.random((n, 3)) - 0.5) * L
    xi = math.pi / (2 * 4 * h * math.sqrt(-math.log(1e-3)))
    dpos, dF = to4(pos, 1.0), to4(rng.standard_normal((n, 3)))
    vel = to4(np.zeros((n, 3)), 1.0)
    engs = []
    hold = []
    for e in range(k):
        engs.append(pse_amd.Engine(n, (L, L, L, 0.0), xi=xi, error=1e-3, seed=1, grid=(g, g, g)))
        hold.append(torch.empty((1 << 20) * (3 + 7 * e), dtype=torch.uint8, device="cuda"))   # shifts what the next engine gets
    for rep in range(2):
        out = []
        for eng in engs:
            eng.set_timing(True)
            ts = []
            for it in range(8):
                eng.mobility(dpos, dF, vel=vel)
                ts.append(eng.info()["t_scale"])
            out.append(min(ts[2:]))
        print("grid", g, "t_scale per engine:", " ".join("%.4f" % t for t in out), flush=True)


if __name__ == "__main__":
    main()
